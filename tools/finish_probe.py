"""Development probe: time finish_kernel (forward on images) with phases masked out."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_spectral_codec_amd import _lib
from neural_spectral_codec_amd.encoding import SpectralEncoder
n = 1024
enc = SpectralEncoder(n_elevation=16).to("cuda")
imgs = torch.rand((n, 16, 360), device="cuda") * 80
out = torch.empty((n, 800), device="cuda")
L = _lib.lib(); p = enc._params(); lut = enc._lut(imgs.device); st = _lib.stream_ptr(imgs.device)
def run():
    L.nsc_encode_range_images(_lib.ptr(imgs), n, 16, p, _lib.ptr(lut), _lib.ptr(out), st)
for mask in (0, 4, 8, 12, 16):
    os.environ["NSC_TUNE_SKIP_FINISH"] = str(mask)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    print(f"mask {mask:2d}: {e0.elapsed_time(e1)/50*1e3:.1f} us per launch", flush=True)
