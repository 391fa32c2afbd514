#!/bin/bash
# gpurun -- bash tools/soak_all.sh TAG : every random-shape soak of tools/ at its long setting, one log each
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
run() { local name=$1; shift; timeout -k 10 900 python "$@" > $O/$name.log 2>&1; local rc=$?; echo "$name rc=$rc: $(tail -1 $O/$name.log | cut -c1-400)"; return $rc; }
run fuzz_train tools/fuzz_train.py 300 || exit 1
run fuzz_gat tools/fuzz_gat.py 500 || exit 1
run fuzz_retrieval tools/fuzz_retrieval.py 300 || exit 1
run fuzz_rows tools/fuzz_rows.py 150 || exit 1
run fuzz_miner tools/fuzz_miner.py 60 || exit 1
