"""Per-kernel averages of the GAT traces gpu_session.sh's `gat` step leaves under gpurun_out/TAG/gat_trace_<N>."""
import csv
import glob
import sys

O = sys.argv[1]
for n in (4541, 1024):
    fs = glob.glob(f"{O}/gat_trace_{n}/*/*kernel_stats.csv")
    if not fs:
        continue
    print("N", n)
    for r in csv.DictReader(open(fs[0])):
        if "gemm" in r["Name"] or "aggregate" in r["Name"] or "banded" in r["Name"]:
            print("  %-72s calls %4d avg %.2f us min %.2f" % (r["Name"][28:100], int(r["Calls"]),
                                                              float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
