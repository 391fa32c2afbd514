#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats output directory -> markdown table + steady-state timeline of the pipeline's
streams.  usage: summarize_profile.py gpurun_out/<tag>/bench_trace [steps warmup] > ...
bench.py (N = 1, --no-extras) launches the encoder: spin-up max(40 - W, 1) + W warmup + K timed + 20 alone + 24 alone on two
overlapping streams; the timed region is located by those counts."""
import os
import csv
import glob
import sys

d = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
W = int(sys.argv[3]) if len(sys.argv) > 3 else 20
SOLO, SOLO2 = 20, 24
stats = list(csv.DictReader(open(max(glob.glob(d + "/*/*kernel_stats.csv"), key=os.path.getmtime))))
print("| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|")
for r in stats[:22]:
    print(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | "
          f"{float(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
rows = list(csv.DictReader(open(max(glob.glob(d + "/*/*kernel_trace.csv"), key=os.path.getmtime))))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
enc_all = [k for k in ks if ('encode_fused' in k[2] or 'encode_fast' in k[2])]
if len(enc_all) >= K + SOLO + SOLO2:
    solo2 = enc_all[-SOLO2:]
    solo = enc_all[-(SOLO2 + SOLO):-SOLO2]
    timed = enc_all[-(SOLO2 + SOLO + K):-(SOLO2 + SOLO)]
    enc = timed[len(timed) // 5:]                      # steady state: the last 80 % of the timed region
    dur = [(e[1] - e[0]) / 1e3 for e in enc]
    by_end = sorted(enc, key=lambda e: e[1])
    period_end = (by_end[-1][1] - by_end[0][1]) / 1e3 / (len(enc) - 1)
    period_start = (enc[-1][0] - enc[0][0]) / 1e3 / (len(enc) - 1)
    overlap = [max(0, min(enc[i][1], enc[i + 1][1]) - enc[i + 1][0]) / 1e3 for i in range(len(enc) - 1)]
    busy = 0.0                                          # time during which at least one encoder launch is running
    cur_s, cur_e = enc[0][0], enc[0][1]
    for s, e, _ in enc[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span_all = (max(e[1] for e in enc) - enc[0][0]) / 1e3
    sd = [(e[1] - e[0]) / 1e3 for e in solo]
    p2 = sorted(solo2, key=lambda e: e[1])
    p2_period = (p2[-1][1] - p2[3][1]) / 1e3 / (len(p2) - 4)
    print(f"\nEncoder alone, one launch at a time (the {SOLO} reference launches after the timed region): avg {sum(sd) / len(sd):.1f} us per launch.")
    print(f"Encoder alone, consecutive launches overlapping on two streams (the {SOLO2} launches after those): "
          f"{p2_period:.1f} us per launch (completion to completion).")
    print(f"Steady state of the timed region (last {len(enc)} of its {K} encoder launches): kernel duration avg {sum(dur) / len(dur):.1f} us "
          f"(min {min(dur):.1f}, max {max(dur):.1f}); consecutive launches overlap by avg {sum(overlap) / len(overlap):.1f} us; "
          f"**launch period {period_end:.1f} us** completion to completion ({period_start:.1f} us start to start); an encoder launch is "
          f"running during {busy / 1e3 / span_all * 100:.1f} % of the region.")
    first = [k for k in ks if 'gemm' in k[2] and ', 1>' in k[2]][-(K * 4 // 5):-1]
    last = [k for k in ks if 'gemm' in k[2] and ', 2>' in k[2]][-(K * 4 // 5):-1]
    span = [(b[1] - a[0]) / 1e3 for a, b in zip(first, last) if b[1] > a[0]]
    if span:
        print(f"GNN forward span (input_proj start -> output_proj end) avg {sum(span) / len(span):.1f} us "
              f"(min {min(span):.1f}, max {max(span):.1f}).")
