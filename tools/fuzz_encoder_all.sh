#!/bin/bash
# gpurun -- bash tools/fuzz_encoder_all.sh TAG : tools/fuzz_encoder.py over its fixed parameter sets and 24 random ones
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
for c in 1 2 3 4 5; do timeout -k 10 300 python tools/fuzz_encoder.py 3 128 120000 $c > $O/cfg_$c.log 2>&1 || { tail -5 $O/cfg_$c.log; exit 1; }; echo "config $c: $(tail -1 $O/cfg_$c.log)"; done
for s in $(seq 1 24); do timeout -k 10 300 python tools/fuzz_encoder.py 2 48 60000 -$s > $O/rnd_$s.log 2>&1 || { grep config $O/rnd_$s.log; tail -5 $O/rnd_$s.log; exit 1; }; echo "random $s: $(grep config $O/rnd_$s.log | cut -c1-160) $(tail -1 $O/rnd_$s.log | cut -c1-120)"; done
