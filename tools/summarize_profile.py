#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats output directory -> markdown table + steady-state timeline of the two
pipeline streams.  usage: summarize_profile.py gpurun_out/<tag>_prof > profiles/<tag>_rocprof_summary.md"""
import csv
import glob
import sys

d = sys.argv[1]
stats = list(csv.DictReader(open(glob.glob(d + "/*/*kernel_stats.csv")[0])))
print("| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|")
for r in stats[:22]:
    print(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | "
          f"{float(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
rows = list(csv.DictReader(open(glob.glob(d + "/*/*kernel_trace.csv")[0])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
SOLO = 20   # bench.py ends with 20 encoder-only reference launches outside the timed region
enc = [k for k in ks if ('encode_fused' in k[2] or 'encode_fast' in k[2])][-(SOLO + 60):-SOLO]
if len(enc) > 2:
    dur = [(e[1] - e[0]) / 1e3 for e in enc]
    gap = [(enc[i + 1][0] - enc[i][1]) / 1e3 for i in range(len(enc) - 1)]
    period = (enc[-1][0] - enc[0][0]) / 1e3 / (len(enc) - 1)
    solo = [k for k in ks if ('encode_fused' in k[2] or 'encode_fast' in k[2])][-SOLO:]
    sd = [(e[1] - e[0]) / 1e3 for e in solo]
    print(f"\nEncoder alone (the {SOLO} reference launches after the timed region): avg {sum(sd) / len(sd):.1f} us.")
    print(f"Steady state (last {len(enc)} encoder launches of the timed region): encoder duration avg {sum(dur) / len(dur):.1f} us "
          f"(min {min(dur):.1f}, max {max(dur):.1f}); idle between consecutive encoder launches avg {sum(gap) / len(gap):.1f} us; "
          f"launch period {period:.1f} us.")
    first = [k for k in ks if 'gemm' in k[2] and ', 1>' in k[2]][-61:-1]
    last = [k for k in ks if 'gemm' in k[2] and ', 2>' in k[2]][-61:-1]
    span = [(b[1] - a[0]) / 1e3 for a, b in zip(first, last)]
    if span:
        print(f"GNN forward span (input_proj start -> output_proj end) avg {sum(span) / len(span):.1f} us "
              f"(min {min(span):.1f}, max {max(span):.1f}).")
