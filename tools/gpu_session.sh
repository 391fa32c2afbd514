#!/bin/bash
# One driver for the GPU-box sessions of a round (replaces the round-2 r02_*.sh one-offs).
#   gpurun -- bash tools/gpu_session.sh TAG STEP [args...] [-- STEP [args...]] ...
# Steps (joined with `--`; a failing step stops the session, nothing further touches the GPU):
#   suite [pytest args]      GPU test suite (-m gpu) into $O/pytest.log
#   bench [bench args]       the driver's command (--steps 20 --warmup 5) + the long run; both JSON lines kept
#   one NAME [bench args]    one bench.py invocation -> $O/NAME.json, one summary line
#   ab CFG...                development build, interleaved A/B of encoder variants (tools/ab_enc.py); AB_ORDER honoured
#   gat                      GAT parity tests, forward timings, per-kernel trace at N = 4541 / 1024
#   build [DEFINE ...]       rebuild csrc/libnsc_hip.so with -DDEFINE ... (no argument: the product build)
#   py SCRIPT [args]         python SCRIPT args  -> $O/<script>.log
#   prof NAME CMD...         rocprofv3 --kernel-trace --stats -d $O/NAME -- CMD...
# Every step's status is propagated, and a "Memory access fault" / core dump anywhere in the session's logs makes the
# session exit 1 even when the shell saw the faulting program end with status 0 (round 2: a faulting probe's script
# ended in `cat log` and exited 0).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:?tag}; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
status=0

fault_check() {
    local hits
    hits=$(grep -rIl -e "Memory access fault" -e "GPU core dump" -e "dumped core" -e "HSA_STATUS_ERROR" $O 2>/dev/null | head -5)
    if [ -n "$hits" ]; then
        echo "[gpu_session] GPU FAULT reported in:"; echo "$hits"
        status=1
    fi
}

run_step() {
    local step=$1; shift
    local rc=0
    case $step in
    suite)
        cd $R; timeout -k 10 1500 python -m pytest tests -x -q -m gpu "$@" > $O/pytest.log 2>&1; rc=$?
        echo "pytest rc=$rc" >> $O/pytest.log; tail -8 $O/pytest.log; return $rc ;;
    bench)
        cd /tmp
        timeout -k 10 600 python $R/bench.py --gpus 1 --steps 20 --warmup 5 "$@" > $O/bench_20_5.json 2> $O/bench_20_5.err || { tail -5 $O/bench_20_5.err; return 1; }
        timeout -k 10 600 python $R/bench.py --gpus 1 --steps 200 --warmup 50 --no-cpu-baseline "$@" > $O/bench_200_50.json 2> $O/bench_200_50.err || { tail -5 $O/bench_200_50.err; return 1; }
        tail -c 2500 $O/bench_20_5.json; tail -c 900 $O/bench_200_50.json ;;
    one)
        local name=$1; shift; cd /tmp
        timeout -k 10 600 python $R/bench.py --gpus 1 "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
        python3 $R/tools/bench_line.py $O/$name.json ;;
    ab)
        cd $R; NSC_DEV_BUILD=1 python neural-spectral-codec_amd/build.py > $O/devbuild.log 2>&1 || { tail -20 $O/devbuild.log; return 1; }
        NSC_DEV_BUILD=1 timeout -k 10 900 python tools/ab_enc.py "$@" >> $O/ab.log 2>&1; rc=$?
        cat $O/ab.log; return $rc ;;
    gat)
        cd $R; timeout -k 10 900 python -m pytest tests/test_gat_gpu.py -x -q -m gpu > $O/pytest_gat.log 2>&1; rc=$?
        tail -3 $O/pytest_gat.log; [ $rc -eq 0 ] || return 1
        for a in "4541 200 0" "1024 200 0" "4541 200 2" "1024 200 2" "4541 200 1" "1024 200 1"; do python tools/gat_workload.py $a >> $O/gat_time.log 2>&1 || return 1; done
        grep -v amdgpu.ids $O/gat_time.log
        cd /tmp
        for n in 4541 1024; do
            rocprofv3 --kernel-trace --stats --output-format csv -d $O/gat_trace_$n -- python3 $R/tools/gat_workload.py $n 50 > $O/gat_trace_$n.log 2>&1 || return 1
        done
        python3 $R/tools/summarize_gat_trace.py $O ;;
    build)
        # build [DEFINE ...]: rebuild the library with -DDEFINE ... (A/B builds; `build` alone restores the product build)
        cd $R; NSC_DEV_DEFINES="$*" python neural-spectral-codec_amd/build.py > $O/build_$(echo "$*" | tr -c 'A-Za-z0-9\n' '_').log 2>&1; rc=$?
        echo "build [$*] rc=$rc"; return $rc ;;
    py)
        local script=$1; shift; cd $R
        local log=$O/$(basename $script .py).log
        timeout -k 10 900 python $script "$@" >> $log 2>&1; rc=$?
        tail -40 $log; return $rc ;;
    prof)
        local name=$1; shift; cd /tmp
        rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- "$@" > $O/$name.log 2>&1; rc=$?
        tail -3 $O/$name.log; return $rc ;;
    *) echo "unknown step $step"; return 2 ;;
    esac
}

args=()
flush() {
    [ ${#args[@]} -eq 0 ] && return
    echo "[gpu_session] step: ${args[*]}"
    run_step "${args[@]}"; local rc=$?
    fault_check
    if [ $rc -ne 0 ] || [ $status -ne 0 ]; then
        echo "[gpu_session] step '${args[0]}' failed (rc=$rc); stopping"; exit 1
    fi
    args=()
}
for a in "$@"; do
    if [ "$a" == "--" ]; then flush; else args+=("$a"); fi
done
flush
exit $status
