#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (tools/collect_profiles.sh) -> profiles/<tag>_pmc_encoder.md and
profiles/encoder_traffic.json (what bench.py reports as roofline.traffic).
usage: summarize_pmc_traffic.py gpurun_out/<tag>_profiles <tag>"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

d, tag = sys.argv[1], sys.argv[2]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(sub, counter):
    rows = csv.DictReader(open(max(glob.glob(f"{d}/{sub}/*/*counter_collection.csv"), key=os.path.getmtime)))
    acc = defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter:
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            acc[re.sub(r"\(.*", "", name)].append(float(r["Counter_Value"]))
    return acc


fetch, write = per_kernel("pmc_fetch", "FETCH_SIZE"), per_kernel("pmc_write", "WRITE_SIZE")
enc = [k for k in fetch if k.startswith("encode_fast_kernel")][0]
f, w = fetch[enc], write[enc]
fa, wa = sum(f) / len(f), sum(w) / len(w)
n_clouds, n_pts = 1024, 120000
alg_r, alg_w = n_clouds * n_pts * 16, n_clouds * 3200
rd, wr = fa * 1024 * 2, wa * 1024
tot, alg = rd + wr, alg_r + alg_w
others = "; ".join(f"`{k}` FETCH {sum(v) / len(v):,.0f} / WRITE {sum(write.get(k, [0])) / max(len(write.get(k, [0])), 1):,.0f}"
                   for k, v in fetch.items() if k != enc and ("gemm" in k or "aggregate" in k))
md = f"""# Round {tag[1:].lstrip("0")} - HBM traffic of the encoder kernel (PMC counters)

Command (two separate passes, as MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE and WRITE_SIZE do
not fit one pass; no trace domains besides --kernel-trace; tools/collect_profiles.sh {tag}, table by tools/summarize_pmc_traffic.py):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/{tag}_profiles/pmc_fetch -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/{tag}_profiles/pmc_write -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras

Per dispatch of `{enc}` (grid 1024 x 256 threads, 27.9 KB LDS, 1 024 clouds x 120 000 points), {len(f)} dispatches
(counter mode serialises the two pipeline streams, so the figures are per kernel, undisturbed):

| counter | value (KB) | min | max |
|---|---|---|---|
| FETCH_SIZE | {fa:,.1f} | {min(f):,.1f} | {max(f):,.1f} |
| WRITE_SIZE | {wa:,.1f} | {min(w):,.1f} | {max(w):,.1f} |

Corrections (guide, gfx950): FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
16 B/lane streaming read -> x2; WRITE_SIZE is exact.

    read  = {fa:,.1f} KB x 1024 x 2 = {rd:,.0f} B   (algorithmic: 1024 x 120 000 x 16 B = {alg_r:,} B)
    write = {wa:,.1f} KB x 1024     = {wr:,.0f} B   (algorithmic: 1024 x 3 200 B = {alg_w:,} B)
    HBM traffic per launch = {tot:,.0f} B = {tot / alg:.4f} x the algorithmic {alg:,} B

No wasted re-reads: every point is fetched once (the uncertain-point queue lives in LDS, the re-stream fallback did not
trigger), nothing but the 800-float descriptors is written.

Other kernels of the step (same passes, per dispatch, uncorrected KB): {others}.
"""
open(os.path.join(R, "profiles", f"{tag}_pmc_encoder.md"), "w").write(md)
json.dump({"kernel": enc, "workload": "1024 clouds x 120000 points (bench.py default)",
           "source": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes (profiles/{tag}_pmc_encoder.md)",
           "FETCH_SIZE_KB_per_launch": fa, "WRITE_SIZE_KB_per_launch": wa, "gfx950_fetch_correction": 2.0,
           "hbm_bytes_per_launch": int(round(tot)), "algorithmic_bytes_per_launch": alg},
          open(os.path.join(R, "profiles", "encoder_traffic.json"), "w"), indent=1)
print(md)
