#!/usr/bin/env python3
"""gpurun_out/<tag>_profiles/gat (tools/gat_profile.sh) -> profiles/<round>_gat_n4541_{rocprof.md,kernel_stats.csv,pmc.csv},
<round>_gat_n1024_kernel_stats.csv.  usage: publish_gat_profiles.py gpurun_out/<tag>_profiles/gat r02"""
import collections
import csv
import glob
import os
import re
import shutil
import subprocess
import sys

d, T = sys.argv[1], sys.argv[2]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(R, "profiles")
shutil.copy(max(glob.glob(f"{d}/trace/*/*kernel_stats.csv"), key=os.path.getmtime), f"{P}/{T}_gat_n4541_kernel_stats.csv")
shutil.copy(max(glob.glob(f"{d}/trace1024/*/*kernel_stats.csv"), key=os.path.getmtime), f"{P}/{T}_gat_n1024_kernel_stats.csv")
acc = collections.defaultdict(list)
for sub in ("pmc1", "pmc2"):
    for r in csv.DictReader(open(max(glob.glob(f"{d}/{sub}/*/*counter_collection.csv"), key=os.path.getmtime))):
        name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        if "gemm" in name or "aggregate" in name or "banded" in name:
            acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(f"{P}/{T}_gat_n4541_pmc.csv", "w") as f:
    f.write("kernel,counter,dispatches,avg,min,max\n")
    for (k, c), v in sorted(acc.items()):
        f.write(f'"{k}",{c},{len(v)},{sum(v) / len(v):.1f},{min(v):.1f},{max(v):.1f}\n')
mid = subprocess.run([sys.executable, os.path.join(R, "tools", "summarize_gat_profile.py"), d], capture_output=True, text=True,
                     check=True).stdout
old = open(f"{P}/{T}_gat_n4541_rocprof.md").read()
head = re.sub(r"gpurun_out/\S*?/gat/trace", d.rstrip("/") + "/trace", old[:old.index("## Kernel durations")])
reading = old[old.index("\n## Reading"):]          # hand-written: update it when the figures move
open(f"{P}/{T}_gat_n4541_rocprof.md", "w").write(head + mid + reading)
print(mid)
