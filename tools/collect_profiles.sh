#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel stats of the same workload.
# Outputs land in gpurun_out/; copy the summaries you want judged into profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
# profile the step path the un-profiled run chose (the calibration would otherwise decide under profiler overhead)
PATHFLAG=$(python -c "import json,sys; print('--' + json.loads(open('$R/gpurun_out/${TAG}_bench.json').read().strip().splitlines()[-1])['step_path'])")
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline $PATHFLAG > $R/gpurun_out/${TAG}_prof.log 2>&1
tail -c 3000 $R/gpurun_out/${TAG}_bench.json
