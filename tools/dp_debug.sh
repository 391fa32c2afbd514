#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
run() { local name=$1 g=$2; shift; shift
  env NSC_DP_DEBUG=1 "$@" python tests/multirank_train_worker.py --rank 0 --world 1 --port 0 --out $O/$name.npz --use-graph $g --batches 9 --batch-size 128 > $O/$name.log 2>&1
  echo "$name:"; grep "LOSSES\|STEP" $O/$name.log | cut -c1-420
}
run eager 0 NSC_TRAINER_FUSED_ADAM=0
run cap_nodirect 1 NSC_TRAINER_FUSED_ADAM=0 NSC_TRAINER_NO_DIRECT=1
run cap_direct 1 NSC_TRAINER_FUSED_ADAM=0
