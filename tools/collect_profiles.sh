#!/bin/bash
# Run on the GPU box (via gpurun): the bench line under the driver's command, rocprofv3 kernel stats of the same
# workload, PMC traffic of the encoder kernel (separate passes), the GAT half (trace + MFMA counters), the training step.
# Outputs land in gpurun_out/<TAG>_profiles/; tools/publish_bench_profiles.sh copies the summaries into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}
O=$R/gpurun_out/${TAG}_profiles
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
python3 $R/bench.py --gpus 1 --steps 200 --warmup 50 --no-cpu-baseline --no-extras > $O/bench_n1_long.json 2> $O/bench_n1_long.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras > $O/bench_trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_write.log 2>&1 || exit 1
bash $R/tools/gat_profile.sh ${TAG}_profiles/gat || exit 1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_trace -- python3 $R/tools/train_probe.py > $O/train_trace.log 2>&1 || exit 1
python3 $R/tools/train_probe.py > $O/train_unprofiled.log 2>&1
if grep -rIl -e "Memory access fault" -e "GPU core dump" $O/*.log $O/*.err 2>/dev/null | grep -q .; then echo "GPU FAULT in the logs"; exit 1; fi
tail -c 400 $O/bench_n1.json; grep "N=" $O/train_unprofiled.log
