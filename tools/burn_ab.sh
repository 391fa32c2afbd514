#!/bin/bash
# What does ONE resource cost the encoder it runs beside?  The pipelined step with a synthetic co-runner (nsc_debug_burn) in
# place of the GNN: MFMAs only / v_fma only / L2-resident loads only / ds_bpermute only, in the co-resident kernels'
# footprint, sized like the GNN forward at 1 024 keyframes (614 k MFMAs, ~310 MB of L1 traffic, ~1.2 M bpermutes).
# usage (GPU box): bash tools/burn_ab.sh TAG
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-burn}; mkdir -p $O; cd /tmp
run() { name=$1; shift
  timeout -k 10 300 python3 $R/bench.py --gpus 1 --steps 100 --warmup 20 --no-cpu-baseline --no-extras --gnn-graph 0 "$@" > $O/$name.json 2> $O/$name.err || { tail -3 $O/$name.err; return 1; }
  python3 $R/tools/bench_line.py $O/$name.json; }
for rep in 1 2; do
  run nognn_$rep --no-gnn || exit 1
  run full_$rep || exit 1
  run mfma_$rep --gnn-burn 0:256:600 || exit 1
  run mfma8_$rep --gnn-burn 0:256:76:8 || exit 1
  run mfma2x_$rep --gnn-burn 0:256:1200 || exit 1
  run valu_$rep --gnn-burn 1:256:6000 || exit 1
  run load_$rep --gnn-burn 2:256:304 || exit 1
  run load8_$rep --gnn-burn 2:256:38:8 || exit 1
  run bperm_$rep --gnn-burn 3:256:1180 || exit 1
  run empty8_$rep --gnn-burn 1:256:4:8 || exit 1
done
