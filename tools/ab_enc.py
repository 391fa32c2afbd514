"""Interleaved A/B timing of encoder variants (median of rounds) -- robust against box-to-box and
minute-to-minute clock drift.  usage: ab_enc.py "VARIANT=0" "VARIANT=1" "VARIANT=0,SKIP_FINISH=64" ...
(VARIANT 0 = encode_fast_kernel, 1 = the fused kernel forced; SKIP_FINISH = phase masks of csrc/nsc_encoder.hip)"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_spectral_codec_amd import synth
from neural_spectral_codec_amd.encoding import SpectralEncoder
n, npts = int(os.environ.get("AB_CLOUDS", "1024")), 120000
enc = SpectralEncoder(n_elevation=16).to("cuda")
pts, off = synth.make_clouds_device(n, npts, "cuda", order=os.environ.get("AB_ORDER", "uniform"))
out = torch.empty((n, 800), device="cuda")
cfgs = sys.argv[1:] or ["VARIANT=0"]
def setenv(c):
    for k in list(os.environ):
        if k.startswith("NSC_TUNE_"): os.environ.pop(k)
    for kv in c.split(","):
        if kv:
            k, v = kv.split("="); os.environ["NSC_TUNE_" + k] = v
def timeit(reps=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): enc.encode_points_batch((pts, off), out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
res = {c: [] for c in cfgs}
for c in cfgs:
    setenv(c); timeit(3)
for rnd in range(7):
    for c in cfgs:
        setenv(c); res[c].append(timeit())
gb = n * (npts * 16 + 3200) / 1e9
for c in cfgs:
    m = statistics.median(res[c])
    print(f"{c:40s} median {m:7.1f} us  min {min(res[c]):7.1f}  max {max(res[c]):7.1f}   {gb/m*1e6:.0f} GB/s", flush=True)
