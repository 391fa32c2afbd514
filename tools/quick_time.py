"""Quick device timing of the encoder (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_spectral_codec_amd import synth
from neural_spectral_codec_amd.encoding import SpectralEncoder
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 120000
enc = SpectralEncoder(n_elevation=16).to("cuda")
pts, off = synth.make_clouds_device(n, npts, "cuda")
out = torch.empty((n, 800), device="cuda")
for _ in range(3):
    enc.encode_points_batch((pts, off), out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10
e0.record()
for _ in range(reps):
    enc.encode_points_batch((pts, off), out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
gb = n * (npts * 16 + 3200) / 1e9
print(f"n={n} npts={npts}: {ms*1e3:.1f} us/batch  {n/ms*1e3:.0f} kf/s  {gb/ms*1e3:.1f} GB/s")
