#!/bin/bash
# GPU suite + the two bench invocations that must agree (driver's --steps 20 --warmup 5 and the long one).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02b}
mkdir -p $O
cd $R
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err
python $R/bench.py --gpus 1 --steps 200 --warmup 50 --no-cpu-baseline > $O/bench_200_50.json 2> $O/bench_200_50.err
python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_5b.json 2> $O/bench_20_5b.err
tail -c 1500 $O/bench_20_5.json; tail -c 700 $O/bench_200_50.json
