#!/bin/bash
# GAT parity tests + forward timings + per-kernel durations (shipped build)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02_gat1}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gat_gpu.py -x -q -m gpu > $O/pytest_gat.log 2>&1; rc=$?; echo "rc=$rc" >> $O/pytest_gat.log
tail -3 $O/pytest_gat.log
[ $rc -eq 0 ] || exit 1
for a in "4541 200 0" "1024 200 0" "4541 200 1" "1024 200 1"; do python tools/gat_workload.py $a >> $O/time.log 2>&1; done
grep -v amdgpu.ids $O/time.log
cd /tmp && export TMPDIR=/tmp
for n in 4541 1024; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_${n} -- python3 $R/tools/gat_workload.py $n 50 > $O/trace_${n}.log 2>&1
done
python3 - <<PY
import csv,glob
for n in (4541,1024):
    f=glob.glob("$O/trace_%d/*/*kernel_stats.csv"%n)[0]
    print("N",n)
    for r in list(csv.DictReader(open(f)))[:5]:
        if 'gemm' in r['Name'] or 'aggregate' in r['Name']:
            print("  %-50s avg %.2f us min %.2f" % (r['Name'][28:78], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
