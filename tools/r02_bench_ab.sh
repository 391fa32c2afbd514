#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02k}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for k in auto lds auto lds; do
python $R/bench.py --gpus 1 --steps 100 --warmup 20 --no-cpu-baseline --no-extras --pipelined --gnn-kernels $k > $O/b_$k.json 2> $O/b_$k.err
python3 - <<PY
import json
l=json.loads(open("$O/b_$k.json").read().strip().splitlines()[-1])
print("$k", round(l['value']), round(l['ms_per_step'],4), round(l['roofline']['launch_ms'],4), round(l['roofline']['standalone_launch_ms'],4))
PY
done
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/full.json 2> $O/full.err
tail -c 2500 $O/full.json
