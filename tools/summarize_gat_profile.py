#!/usr/bin/env python3
"""rocprofv3 output of tools/gat_profile.sh (kernel trace + two PMC passes of the config-3 GAT forward) -> markdown.
usage: summarize_gat_profile.py gpurun_out/<tag> > profiles/<round>_gat_n4541_rocprof.md"""
import collections
import os
import csv
import glob
import re
import sys

d = sys.argv[1]
N = 4541
FLOP = {  # f32 MFMA work per launch at N = 4541 (2 M N K)
    "1>": 2.0 * N * 256 * 800, "0>": 2.0 * N * 258 * 256, "2>": 2.0 * N * 800 * 256, "band": 2.0 * N * 256 * 256,
}
NAME = {"1>": "input_proj 800->256 (+BN+ReLU)", "0>": "GATConv lin 256->256 (+2 attention columns)",
        "2>": "output_proj 256->800 (+residual)",
        "band": "GATConv layer in one launch: lin 256->256 + attention chains + softmax + aggregation + BN"}
PEAK = 157.3e12


def key(name):
    if "gemm_nt" in name or "gemm_glds" in name:
        m = re.search(r"gemm_(?:nt|glds)\w*kernel<\s*\d+,\s*(\d+)", name)     # <ACC, EPI, ...>: the role is the EPI argument
        if m and m.group(1) + ">" in FLOP:
            return m.group(1) + ">"
    if "gat_layer_banded" in name:
        return "band"
    if "gat_aggregate" in name:
        return "agg"
    return None


stats = list(csv.DictReader(open(max(glob.glob(d + "/trace/*/*kernel_stats.csv"), key=os.path.getmtime))))
dur = {}
print("## Kernel durations (rocprofv3 --kernel-trace --stats, `tools/gat_workload.py 4541 50`)\n")
print("| kernel | role | calls | avg us | min us | GFLOP | TFLOP/s (avg) | % of 157.3 TF |\n|---|---|---|---|---|---|---|---|")
tot = 0.0
for r in stats:
    k = key(r["Name"])
    if k is None:
        continue
    avg, mn, calls = float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, int(r["Calls"])
    dur[k] = avg
    per_fwd = 3 if k in ("0>", "agg", "band") else 1
    tot += avg * per_fwd
    if k == "agg":
        print(f"| `gat_aggregate_kernel<1,8,true>` | attention softmax + aggregation (x3 per forward) | {calls} | {avg:.2f} | {mn:.2f} | - | - | - |")
    else:
        tf = FLOP[k] / (avg * 1e-6) / 1e12
        base = re.sub(r"<.*", "", r["Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        print(f"| `{base}{'' if k == 'band' else '<*,' + k[0] + '>'}` | {NAME[k]}{' (x3 per forward)' if k in ('0>', 'band') else ''} | {calls} | {avg:.2f} | {mn:.2f} | "
              f"{FLOP[k] / 1e9:.3f} | {tf:.1f} | {tf / 157.3 * 100:.1f} |")
fl = FLOP["1>"] + 3 * FLOP["0>"] + FLOP["2>"]       # SURVEY 8(d)'s count (the two attention columns as MFMA work)
print(f"\nSum of kernel time per forward: {tot:.1f} us for {fl / 1e9:.2f} GFLOP of MFMA work = "
      f"{fl / (tot * 1e-6) / 1e12:.1f} TFLOP/s ({fl / (tot * 1e-6) / PEAK * 100:.1f} % of the 157.3 TF f32-MFMA peak).")
for line in open(d + "/unprofiled.log"):
    if line.startswith("N="):
        print("Un-profiled HIP-event time: " + line.strip())

ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/pmc*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = key(r["Kernel_Name"])
        if k:
            ctr[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("\n## MFMA counters (rocprofv3 --pmc, separate passes; averages per launch)\n")
print("| kernel | SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 = FLOP | SQ_VALU_MFMA_BUSY_CYCLES | cycles per MFMA | GRBM_GUI_ACTIVE / 8 XCDs | "
      "MfmaUtil = BUSY / (GUI_ACTIVE/8 x 1024 SIMDs) | SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES | SQ_WAIT_ANY / SQ_WAVE_CYCLES |\n|---|---|---|---|---|---|---|---|")
for k in ("1>", "0>", "band", "2>"):
    c = {n: sum(v) / len(v) for n, v in ctr[k].items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
        continue
    flop = c["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512
    n_mfma = flop / 2048.0
    gui = c["GRBM_GUI_ACTIVE"] / 8.0
    util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024.0)
    wi = c.get("SQ_WAIT_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)
    wa = c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)
    print(f"| `{'gat_layer_banded_kernel' if k == 'band' else 'gemm_glds_kernel<*,' + k[0] + '>'}` | {flop / 1e9:.3f} G | {c['SQ_VALU_MFMA_BUSY_CYCLES']:.3g} | "
          f"{c['SQ_VALU_MFMA_BUSY_CYCLES'] / n_mfma:.1f} | {gui:.0f} | {util * 100:.1f} % | {wi * 100:.0f} % | {wa * 100:.0f} % |")
print("\n`SQ_INSTS_VALU_MFMA_MOPS_F32 x 512` reproduces the algorithmic FLOP count (padding rows of the last tile included); "
      "`SQ_VALU_MFMA_BUSY_CYCLES` is 32 cycles per `v_mfma_f32_16x16x4_f32`. `MfmaUtil` is rocprofv3's own derived formula "
      "(busy cycles over elapsed cycles x SIMDs, elapsed = GRBM_GUI_ACTIVE per XCD); MI355X_MICROARCH.md notes that GRBM_GUI_ACTIVE reads high "
      "on dispatches shorter than 0.3 ms, so this column is a lower bound; the time-based column of the first table is FLOP / duration "
      "against the 157.3 TF peak (2.4 GHz).")
