"""Development probe: two-stream pipelining of scatter(k+1) with finish(k)."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, torch
from neural_spectral_codec_amd import synth, _lib
from neural_spectral_codec_amd.encoding import SpectralEncoder
n, npts = 1024, 120000
enc = SpectralEncoder(n_elevation=16).to("cuda")
pts, off = synth.make_clouds_device(n, npts, "cuda")
out = torch.empty((n, 800), device="cuda"); out2 = torch.empty_like(out)
L = _lib.lib(); p = enc._params(); lut = enc._lut(pts.device)
sq = [torch.empty((n, 16, 360), dtype=torch.int32, device="cuda") for _ in range(2)]
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
gb = n * (npts * 16 + 3200) / 1e9
def fused(reps):
    for _ in range(reps): enc.encode_points_batch((pts, off), out=out)
def serial(reps):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for k in range(reps):
        L.nsc_scatter_clouds(_lib.ptr(pts), _lib.ptr(off), n, n * npts, 4, p, _lib.ptr(sq[0]), st)
        L.nsc_finish_images(_lib.ptr(sq[0]), n, p, _lib.ptr(lut), _lib.ptr(out2), None, None, st)
def piped(reps):
    a, b = C.c_void_p(sA.cuda_stream), C.c_void_p(sB.cuda_stream)
    evs = [torch.cuda.Event() for _ in range(2)]
    fin = [torch.cuda.Event() for _ in range(2)]
    for k in range(reps):
        i = k & 1
        if k >= 2: sA.wait_event(fin[i])                 # image buffer i is free again
        L.nsc_scatter_clouds(_lib.ptr(pts), _lib.ptr(off), n, n * npts, 4, p, _lib.ptr(sq[i]), a)
        evs[i].record(sA)
        sB.wait_event(evs[i])
        L.nsc_finish_images(_lib.ptr(sq[i]), n, p, _lib.ptr(lut), _lib.ptr(out2), None, None, b)
        fin[i].record(sB)
    torch.cuda.current_stream().wait_stream(sA); torch.cuda.current_stream().wait_stream(sB)
def timeit(fn, reps=20):
    fn(3); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sA.wait_stream(torch.cuda.current_stream()); sB.wait_stream(torch.cuda.current_stream())
    e0.record(); fn(reps); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
res = {"fused": [], "serial": [], "piped": []}
for rnd in range(5):
    for name, fn in (("fused", fused), ("serial", serial), ("piped", piped)):
        res[name].append(timeit(fn))
for k, v in res.items():
    m = statistics.median(v); print(f"{k:8s} median {m:7.1f} us  min {min(v):7.1f}  {gb/m*1e6:.0f} GB/s", flush=True)
enc.encode_points_batch((pts, off), out=out); serial(1); torch.cuda.synchronize()
print("same result:", torch.equal(out, out2))
