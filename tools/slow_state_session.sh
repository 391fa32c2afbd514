#!/bin/bash
# gpurun -- bash tools/slow_state_session.sh TAG [prof]   (round 4: the slow state of DESIGN.md section 6, one diagnosis session)
# FIRST process on the fresh box = the two-path calibrating bench that keeps its calibrated path object (NSC_BENCH_KEEP_PATH=1:
# the workaround off); then the same again (cured state on 7 of 7 boxes in round 3); then the workaround on.  With `prof` the
# FIRST process runs under rocprofv3 --kernel-trace instead (program directly after --).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
ARGS="--gpus 1 --steps 60 --warmup 5 --ev-every 1 --calibrate2 --no-cpu-baseline --no-extras"
export NSC_BENCH_CLOCKS=1
if [ "$2" == "prof" ]; then
    NSC_BENCH_KEEP_PATH=1 timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_first -- python3 $R/bench.py $ARGS > $O/first_keep.json 2> $O/first_keep.err || { tail -5 $O/first_keep.err; exit 1; }
else
    NSC_BENCH_KEEP_PATH=1 timeout -k 10 600 python3 $R/bench.py $ARGS > $O/first_keep.json 2> $O/first_keep.err || { tail -5 $O/first_keep.err; exit 1; }
fi
NSC_BENCH_KEEP_PATH=1 timeout -k 10 600 python3 $R/bench.py $ARGS > $O/second_keep.json 2> $O/second_keep.err || exit 1
timeout -k 10 600 python3 $R/bench.py $ARGS > $O/third_fresh.json 2> $O/third_fresh.err || exit 1
timeout -k 10 600 python3 $R/bench.py --gpus 1 --steps 60 --warmup 5 --ev-every 1 --no-cpu-baseline --no-extras > $O/fourth_single.json 2> $O/fourth_single.err || exit 1
python3 $R/tools/slow_state_report.py $O/first_keep.json $O/second_keep.json $O/third_fresh.json $O/fourth_single.json
