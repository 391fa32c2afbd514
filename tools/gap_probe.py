"""Dispatch gap between consecutive encoder launches: (a) N launches back to back on one stream between two events,
(b) every launch bracketed by its own event pair, (c) launches alternating over two streams.  usage: gap_probe.py"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_spectral_codec_amd import synth
from neural_spectral_codec_amd.encoding import SpectralEncoder
n, npts = 1024, 120000
enc = SpectralEncoder(n_elevation=16).to("cuda")
pts, off = synth.make_clouds_device(n, npts, "cuda")
outs = [torch.empty((n, 800), device="cuda") for _ in range(4)]
N = 40
def ev(): return torch.cuda.Event(enable_timing=True)
def back_to_back(streams):
    torch.cuda.synchronize()
    cur = torch.cuda.current_stream()
    e0, e1 = ev(), ev()
    for s in streams: s.wait_stream(cur)
    e0.record(cur)
    for s in streams: s.wait_event(e0)
    for k in range(N):
        with torch.cuda.stream(streams[k % len(streams)]):
            enc.encode_points_batch((pts, off), out=outs[k % 4])
    for s in streams: cur.wait_stream(s)
    e1.record(cur); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1e3
def bracketed():
    torch.cuda.synchronize()
    pairs = [(ev(), ev()) for _ in range(N)]
    t0, t1 = ev(), ev()
    t0.record()
    for a, b in pairs:
        a.record(); enc.encode_points_batch((pts, off), out=outs[0]); b.record()
    t1.record(); torch.cuda.synchronize()
    return statistics.mean(a.elapsed_time(b) for a, b in pairs) * 1e3, t0.elapsed_time(t1) / N * 1e3
s1 = [torch.cuda.Stream()]
s2 = [torch.cuda.Stream(), torch.cuda.Stream()]
for _ in range(2): back_to_back(s1); back_to_back(s2); bracketed()
for rnd in range(4):
    a = back_to_back(s1); c = back_to_back(s2); b, bt = bracketed()
    print(f"round {rnd}: one stream {a:7.1f} us/launch | two streams {c:7.1f} | bracketed: kernel {b:7.1f}, period {bt:7.1f}", flush=True)
