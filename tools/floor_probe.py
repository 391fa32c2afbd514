"""Steady-state cost of back-to-back tiny dependent kernels on this box (launch floor)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_spectral_codec_amd import _lib
L = _lib.lib()
h = torch.rand(4, 64, device="cuda"); o = torch.empty_like(h)
st = _lib.stream_ptr(h.device)
def run(reps):
    for _ in range(reps): L.nsc_w1_cdf(_lib.ptr(h), 4, 64, 1e-8, 1, _lib.ptr(o), st)
run(50); torch.cuda.synchronize()
for reps in (200, 1000):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(reps); e1.record(); torch.cuda.synchronize()
    print(f"{reps} tiny kernels: {e0.elapsed_time(e1)/reps*1e3:.2f} us each (GPU timeline)")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    st2 = _lib.stream_ptr(h.device)
    for _ in range(3): L.nsc_w1_cdf(_lib.ptr(h), 4, 64, 1e-8, 1, _lib.ptr(o), st2)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        st3 = _lib.stream_ptr(h.device)
        for _ in range(200): L.nsc_w1_cdf(_lib.ptr(h), 4, 64, 1e-8, 1, _lib.ptr(o), st3)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): g.replay()
e1.record(); torch.cuda.synchronize()
print(f"hipGraph replay of 200 tiny kernels: {e0.elapsed_time(e1)/1000*1e3:.2f} us each")
