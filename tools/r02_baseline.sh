#!/bin/bash
# Round-2 starting point on one box: counter list, driver-style vs long bench, GAT kernel trace.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_5.json 2> $O/bench_20_5.err
python $R/bench.py --gpus 1 --steps 200 --warmup 50 --no-cpu-baseline > $O/bench_200_50.json 2> $O/bench_200_50.err
python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_5b.json 2> $O/bench_20_5b.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gat_prof -- python $R/tools/gat_probe.py > $O/gat_probe.log 2>&1
tail -c 600 $O/bench_20_5.json; tail -c 600 $O/bench_200_50.json; cat $O/gat_probe.log | tail -5
