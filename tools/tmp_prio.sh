#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02pr}
mkdir -p $O
cd $R
NSC_DEV_BUILD=1 python neural-spectral-codec_amd/build.py > $O/devbuild.log 2>&1
cd /tmp && export TMPDIR=/tmp
for k in 0 32 0 32 0 32; do
NSC_TUNE_SKIP_FINISH=$k python $R/bench.py --gpus 1 --steps 100 --warmup 20 --no-cpu-baseline --no-extras --pipelined > $O/b_$k.json 2> $O/b_$k.err
python3 - <<PY
import json
l=json.loads(open("$O/b_$k.json").read().strip().splitlines()[-1])
print("prio_bit=$k", round(l['value']), round(l['ms_per_step'],4), round(l['roofline']['launch_ms'],4), round(l['roofline']['standalone_launch_ms'],4))
PY
done
