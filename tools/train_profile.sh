R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04g_train; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_trace -- python3 $R/tools/train_probe.py > $O/train_trace.log 2>&1 || exit 1
python3 $R/tools/train_probe.py > $O/train_unprofiled.log 2>&1 || exit 1
python3 $R/tools/train_probe2.py > $O/train_probe2.log 2>&1 || exit 1
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
grep N= $O/train_unprofiled.log $O/train_probe2.log
