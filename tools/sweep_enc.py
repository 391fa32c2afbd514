"""Phase costs of the encoder kernels in a development build (NSC_TUNE_* env knobs): VARIANT 0 = encode_fast_kernel, 1 = the
fused kernel forced; SKIP_FINISH masks (1: no finish, 33: loads only); SPLIT forces the split path.  The load-depth / wave-shape
variants of rounds 1-2 (experiments 1-12) are no longer compiled in."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_spectral_codec_amd import synth
from neural_spectral_codec_amd.encoding import SpectralEncoder
n, npts = 1024, 120000
enc = SpectralEncoder(n_elevation=16).to("cuda")
pts, off = synth.make_clouds_device(n, npts, "cuda")
out = torch.empty((n, 800), device="cuda")

def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

gb = n * (npts * 16 + 3200) / 1e9
variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1"]
for v in variants:
    os.environ["NSC_TUNE_VARIANT"] = v
    us = timeit(lambda: enc.encode_points_batch((pts, off), out=out))
    os.environ["NSC_TUNE_SKIP_FINISH"] = "1"
    us2 = timeit(lambda: enc.encode_points_batch((pts, off), out=out))
    os.environ["NSC_TUNE_SKIP_FINISH"] = "33"
    us3 = timeit(lambda: enc.encode_points_batch((pts, off), out=out))
    os.environ.pop("NSC_TUNE_SKIP_FINISH")
    print(f"variant {v}: {us:.1f} us  {gb/us*1e6:.0f} GB/s   scatter-only {us2:.1f} us {gb/us2*1e6:.0f} GB/s   loads-only {us3:.1f} us {gb/us3*1e6:.0f} GB/s", flush=True)
os.environ["NSC_TUNE_VARIANT"] = "0"
for parts in (2, 4):
    os.environ["NSC_TUNE_SPLIT"] = str(parts)
    us = timeit(lambda: enc.encode_points_batch((pts, off), out=out))
    print(f"split parts={parts}: {us:.1f} us  {gb/us*1e6:.0f} GB/s", flush=True)
os.environ.pop("NSC_TUNE_SPLIT")
imgs = torch.rand((n, 16, 360), device="cuda") * 80
us = timeit(lambda: enc.forward(imgs))
print(f"finish-only (forward on {n} images): {us:.1f} us", flush=True)
# plain copy ceiling
a = torch.empty(n * npts * 4, device="cuda"); b = torch.empty_like(a)
us = timeit(lambda: b.copy_(a))
print(f"torch copy {a.numel()*4/1e9:.2f} GB: {us:.1f} us -> read+write {2*a.numel()*4/us/1e3:.0f} GB/s", flush=True)
us = timeit(lambda: a.sum())
print(f"torch sum (read only): {us:.1f} us -> {a.numel()*4/us/1e3:.0f} GB/s", flush=True)
