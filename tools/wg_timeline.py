"""Per-workgroup timeline of encode_fast_kernel (development build, NSC_TUNE_SKIP_FINISH=512): start, end of the
stream, end of the finish of every workgroup on the 100 MHz wall clock -> spread of the phases over the grid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NSC_TUNE_SKIP_FINISH"] = os.environ.get("NSC_TL_MODE", "512")
import numpy as np, torch
from neural_spectral_codec_amd import synth
from neural_spectral_codec_amd.encoding import SpectralEncoder
n, npts = 1024, 120000
enc = SpectralEncoder(n_elevation=16).to("cuda")
pts, off = synth.make_clouds_device(n, npts, "cuda")
out = torch.empty((n, 800), device="cuda")
for _ in range(5): enc.encode_points_batch((pts, off), out=out)
torch.cuda.synchronize()
for rep in range(3):
    enc.encode_points_batch((pts, off), out=out); torch.cuda.synchronize()
    t = out[:, :3].contiguous().view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
    t0, t1, t2 = t[:, 0], t[:, 1], t[:, 2]
    base = t0.min()
    us = lambda v: (v - base) / 100.0
    print(f"rep {rep}: start {us(t0).min():.1f}..{us(t0).max():.1f} us | stream end {us(t1).min():.1f}..{us(t1).max():.1f} "
          f"(p5 {np.percentile(us(t1),5):.1f}, p50 {np.percentile(us(t1),50):.1f}, p95 {np.percentile(us(t1),95):.1f}) | "
          f"finish end {us(t2).min():.1f}..{us(t2).max():.1f} (p50 {np.percentile(us(t2),50):.1f}) | finish len p50 {np.percentile((t2-t1)/100.0,50):.1f} max {((t2-t1)/100.0).max():.1f}")
    st = out[:, 3:10].contiguous().view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
    late = np.argsort(t2)[-128:]                      # the workgroups that end last: their finish is the exposed one
    names = ["sqrt+interp", "tables+sync", "rowfill", "fft+hist", "sum+sync", "normalise+store"]
    for sel, tag in ((late, "last 128 workgroups"), (np.arange(n), "all")):
        d = np.diff(st[sel], axis=1) / 100.0
        print("   finish stages, %s (median us): " % tag + ", ".join(f"{nm} {np.median(d[:, i]):.2f}" for i, nm in enumerate(names))
              + f" | drain+before finish {np.median((st[sel, 0] - t1[sel]) / 100.0):.2f}")
    x = np.arange(n) % 8
    print("   stream end by c%8 (mean us):", " ".join(f"{us(t1)[x == k].mean():.1f}" for k in range(8)))
    grp = (np.arange(n) // 256)
    print("   stream end by c//256 (mean us):", " ".join(f"{us(t1)[grp == k].mean():.1f}" for k in range(4)))
