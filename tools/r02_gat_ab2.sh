#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02_gat5}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gat_gpu.py -x -q -m gpu > $O/pytest_gat.log 2>&1; rc=$?; echo "rc=$rc" >> $O/pytest_gat.log
tail -3 $O/pytest_gat.log
[ $rc -eq 0 ] || exit 1
for defs in "" "NSC_DEV_NOREMAP"; do
NSC_DEV_BUILD=1 NSC_DEV_DEFINES="$defs" python neural-spectral-codec_amd/build.py > $O/devbuild.log 2>&1
cd /tmp && export TMPDIR=/tmp
for acc in 1 2; do for n in 4541 1024; do
  export NSC_TUNE_GEMM_ACC=$acc
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_${defs}_${acc}_${n} -- python3 $R/tools/gat_workload.py $n 50 > $O/trace.log 2>&1
done; done
cd $R
python3 - <<PY
import csv,glob
for acc in (1,2):
  for n in (4541,1024):
    f=glob.glob("$O/trace_${defs}_%d_%d/*/*kernel_stats.csv"%(acc,n))[0]
    print("defs='$defs' ACC",acc,"N",n)
    for r in list(csv.DictReader(open(f)))[:5]:
        if 'gemm' in r['Name'] or 'aggregate' in r['Name']:
            print("  %-70s avg %.2f us min %.2f" % (r['Name'][28:70], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done
