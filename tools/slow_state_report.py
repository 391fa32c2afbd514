#!/usr/bin/env python3
"""Round 4, one diagnosis session: what IS the slow state of the first multi-path process on a fresh box?
Reads bench lines produced with NSC_BENCH_CLOCKS=1 --ev-every 1 (every encoder launch of the timed region stamped at start
and end) and prints, per run: step time, launch period, mean launch duration, mean overlap of consecutive launches, the
clocks read mid-run -- longer launches at equal overlap = clocks / device state; equal launches with less overlap = queues."""
import json
import sys

for path in sys.argv[1:]:
    try:
        l = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception as ex:  # noqa: BLE001
        print(path, "unreadable:", ex)
        continue
    w = l["stream_debug"].get("launch_windows_us") or []
    dur = [e - s for s, e in w]
    ov = [w[k][1] - w[k + 1][0] for k in range(len(w) - 1)]
    per = [w[k + 1][1] - w[k][1] for k in range(len(w) - 1)]
    mid = slice(4, -1)
    def mean(v):
        v = v[mid] if len(v) > 8 else v
        return sum(v) / max(len(v), 1)
    print(f"{path.split('/')[-1]:28s} ms/step {l['ms_per_step']:.4f}  period {mean(per):6.1f} us  launch {mean(dur):6.1f} us  "
          f"overlap {mean(ov):6.1f} us  path {l.get('step_path')}/{l.get('encoder_streams')}  queues {l['stream_debug'].get('hw_queue_classes')}")
    cal = l.get("calibration")
    if cal:
        print("    calibration:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in cal.items() if k != "rounds_ms_per_step_rank0"},
              cal.get("rounds_ms_per_step_rank0"))
    ck = l.get("clocks_mid_run")
    if ck:
        print("    clocks mid-run:", {k: v for k, v in ck.items() if any(t in k for t in ("sclk", "mclk", "fclk", "power1_average", "power1_input", "freq"))})
