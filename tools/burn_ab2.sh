#!/bin/bash
# Second series of tools/burn_ab.sh: where do the co-runner's loads cost -- L1 hits, L2 hits, redundancy within a workgroup?
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-burn2}; mkdir -p $O; cd /tmp
run() { name=$1; shift
  timeout -k 10 300 python3 $R/bench.py --gpus 1 --steps 100 --warmup 20 --no-cpu-baseline --no-extras --gnn-graph 0 "$@" > $O/$name.json 2> $O/$name.err || { tail -3 $O/$name.err; return 1; }
  python3 $R/tools/bench_line.py $O/$name.json; }
for rep in 1 2; do
  run nognn_$rep --no-gnn || exit 1
  run l2_304_$rep --gnn-burn 2:256:304 || exit 1
  run l2_152_$rep --gnn-burn 2:256:152 || exit 1
  run l1_304_$rep --gnn-burn 4:256:304 || exit 1
  run wgshared_304_$rep --gnn-burn 5:256:304 || exit 1
  run l2_304_1024wg_$rep --gnn-burn 2:1024:76 || exit 1
  run full_$rep || exit 1
  run full_graph_$rep --gnn-graph 1 || exit 1
done
