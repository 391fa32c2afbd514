#!/bin/bash
# gpurun -- bash tools/rehearse_n.sh TAG : the N > 1 paths of bench.py rehearsed on ONE card (NSC_BENCH_REHEARSAL=1: ranks share cuda:0,
# gloo): launched the driver's way (python -m torch.distributed.run, 2 ranks) and by bench.py itself (3 ranks).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp
export NSC_BENCH_REHEARSAL=1
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 $R/bench.py --gpus 2 --steps 10 --warmup 3 --clouds 256 > $O/torchrun2.json 2> $O/torchrun2.err || { tail -5 $O/torchrun2.err; exit 1; }
timeout -k 10 600 python $R/bench.py --gpus 3 --steps 10 --warmup 3 --clouds 256 > $O/self3.json 2> $O/self3.err || { tail -5 $O/self3.err; exit 1; }
unset NSC_BENCH_REHEARSAL
# ... and bench.py's N > 1 branch with every collective through RCCL (one rank, nccl backend), full size
NSC_BENCH_RCCL_WORLD1=1 timeout -k 10 600 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/rccl1.json 2> $O/rccl1.err || { tail -5 $O/rccl1.err; exit 1; }
python3 - "$O" <<'PY'
import json, sys
for f in ("torchrun2", "self3", "rccl1"):
    l = json.loads(open(f"{sys.argv[1]}/{f}.json").read().strip().splitlines()[-1])
    print(f, "n_gpus", l["n_gpus"], "rccl_ranks", l["rccl_ranks"], l["backend"], "|", l["launched_by"], "|", l["step_path"], l["encoder_streams"],
          [round(v, 3) for v in l["ms_per_step_by_rank"]], {k: (round(v, 3) if isinstance(v, float) else v) for k, v in l["allgather"].items()})
    print("   calibration", {k: v for k, v in l["calibration"].items() if k != "rounds_ms_per_step_rank0"})
PY
