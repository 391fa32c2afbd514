#!/usr/bin/env python3
"""Development probe: does an LDS-free, low-VGPR kernel on a second stream run CONCURRENTLY with the fully
resident encoder grid (4 workgroups/CU leave 2.3 KB of LDS, 16 wave slots and 64 VGPRs per SIMD lane)?
Filler = chain_graph_kernel (0 LDS, 62 VGPRs) on a 200k-node chain."""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from neural_spectral_codec_amd import synth                                               # noqa: E402
from neural_spectral_codec_amd.encoding import SpectralEncoder                            # noqa: E402
from neural_spectral_codec_amd.keyframe.graph_manager import chain_graph_device           # noqa: E402

n, npts = 1024, 120000
enc = SpectralEncoder(n_elevation=16).to("cuda")
pts, off = synth.make_clouds_device(n, npts, "cuda")
out = torch.empty((n, 800), device="cuda")
NN = 200000
poses = torch.from_numpy(synth.make_pose_chain(NN, 0)).cuda()
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
NE, NF = 30, 120


def run_enc():
    with torch.cuda.stream(sA):
        for _ in range(NE):
            enc.encode_points_batch((pts, off), out=out)


import ctypes as C                                                                          # noqa: E402
from neural_spectral_codec_amd import _lib                                                # noqa: E402
MODE = sys.argv[1] if len(sys.argv) > 1 else "chain"
WGS = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ITERS = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
sink = torch.zeros(4, device="cuda")
if MODE == "spin":
    spin = _lib.lib().nsc_dev_spin                      # dev builds only (NSC_DEV_BUILD=1)
    spin.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]


def run_fill():
    with torch.cuda.stream(sB):
        for _ in range(NF):
            if MODE == "spin":
                spin(WGS, ITERS, sink.data_ptr(), sB.cuda_stream)
            else:
                chain_graph_device(NN, 5, "cuda", poses)


def timed(*fns):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sA.wait_event(e0)
    sB.wait_event(e0)
    for f in fns:
        f()
    torch.cuda.current_stream().wait_stream(sA)
    torch.cuda.current_stream().wait_stream(sB)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for _ in range(2):
    timed(run_enc, run_fill)
a = min(timed(run_enc) for _ in range(3))
b = min(timed(run_fill) for _ in range(3))
c = min(timed(run_enc, run_fill) for _ in range(3))
print(f"mode={MODE} variant={os.environ.get('NSC_TUNE_VARIANT', '0')}  encoder x{NE}: {a:.2f} ms ({a / NE * 1e3:.0f} us each)   "
      f"filler x{NF}: {b:.2f} ms ({b / NF * 1e3:.0f} us each)   both: {c:.2f} ms   (serial sum {a + b:.2f}, ideal overlap {max(a, b):.2f})")
