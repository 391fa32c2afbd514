#!/bin/bash
# Copy the bench-side records of tools/collect_profiles.sh into profiles/ and write the summaries.
# usage: tools/publish_bench_profiles.sh gpurun_out/<tag>_profiles r02
S=$1; T=${2:-r02}
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
tail -n 1 $S/bench_n1.json > profiles/${T}_bench_n1.json
tail -n 1 $S/bench_n1_long.json > profiles/${T}_bench_n1_long.json
cp $S/bench_trace/*/*kernel_stats.csv profiles/${T}_rocprof_kernel_stats.csv
python3 tools/summarize_pmc_traffic.py $S $T > /dev/null
python3 - "$S" "$T" <<'PY' > profiles/${T}_rocprof_summary.md
import json, subprocess, sys
S, T = sys.argv[1], sys.argv[2]
a = json.loads(open(f"{S}/bench_n1.json").read().strip().splitlines()[-1])
b = json.loads(open(f"{S}/bench_n1_long.json").read().strip().splitlines()[-1])
table = subprocess.run([sys.executable, "tools/summarize_profile.py", f"{S}/bench_trace"], capture_output=True, text=True, check=True).stdout
ra, rb = a["roofline"], b["roofline"]
print(f"""# Round 2 - rocprofv3 --kernel-trace --stats of the bench workload

Commands (tools/collect_profiles.sh, published by tools/publish_bench_profiles.sh): `python3 bench.py --gpus 1 --steps 20 --warmup 5` (un-profiled, the driver's command; profiles/{T}_bench_n1.json),
`python3 bench.py --gpus 1 --steps 200 --warmup 50` (profiles/{T}_bench_n1_long.json), then on the same box
`cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/.../bench_trace -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --{a['step_path']}`
(`--{a['step_path']}` pins the step path the un-profiled run's calibration chose). Full CSV: profiles/{T}_rocprof_kernel_stats.csv; PMC traffic of the encoder
kernel: profiles/{T}_pmc_encoder.md; table by tools/summarize_profile.py.

The step is the two-stream software pipeline (DESIGN.md section 5): `encode_fast_kernel<2>` of batch k+1 on one stream, the LDS-free GNN
kernels of batch k (`gemm_nt_direct_kernel`, `gat_aggregate_kernel<1,4,false>`) beside it on a second one. The GNN kernel durations are
co-running durations (alone: the minima; the LDS-tiled `gemm_nt_kernel` rows are the serial path's calibration steps).

{table}
HIP events vs kernel trace: the un-profiled runs give {ra['launch_ms']*1e3:.1f} us (--steps 20 --warmup 5) / {rb['launch_ms']*1e3:.1f} us (--steps 200 --warmup 50) per timed
encoder launch in situ, {ra['standalone_launch_ms']*1e3:.1f} / {rb['standalone_launch_ms']*1e3:.1f} us for the kernel alone, {a['ms_per_step']*1e3:.1f} / {b['ms_per_step']*1e3:.1f} us per step
({a['value']/1e6:.2f} / {b['value']/1e6:.2f} M keyframes/s: the two invocations agree within {abs(a['value']-b['value'])/max(a['value'],b['value'])*100:.1f} %). `roofline.frac` {ra['frac']:.3f} / {rb['frac']:.3f} in situ,
{ra['standalone_frac']:.3f} / {rb['standalone_frac']:.3f} alone. An event pair also brackets part of the dispatch gap between two launches on the stream, so
`roofline.launch_ms` is a few microseconds above the trace's kernel duration: pessimistic for `roofline.frac`.""")
PY
tail -8 profiles/${T}_rocprof_summary.md
