"""Where does the idle between two encoder launches of the two-stream step come from?  40 launches on stream E with,
after each launch: (a) nothing, (b) an event record on E, (c) b + stream G waits for it and records its own event,
(d) c + a small kernel on G, (e) d with a host-side query of G's event 4 steps back (what distributed.py does)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_spectral_codec_amd import synth
from neural_spectral_codec_amd.encoding import SpectralEncoder
n, npts = 1024, 120000
enc = SpectralEncoder(n_elevation=16).to("cuda")
pts, off = synth.make_clouds_device(n, npts, "cuda")
outs = [torch.empty((n, 800), device="cuda") for _ in range(4)]
N = 40
E, G = torch.cuda.Stream(), torch.cuda.Stream()
small = torch.zeros(1024, device="cuda")
def run(mode, flags=0):
    torch.cuda.synchronize()
    cur = torch.cuda.current_stream()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    evE = [torch.cuda.Event() if not flags else torch.cuda.Event(blocking=False, interprocess=False) for _ in range(N)]
    evG = [torch.cuda.Event() for _ in range(N)]
    E.wait_stream(cur); G.wait_stream(cur)
    t0.record(cur); E.wait_event(t0); G.wait_event(t0)
    for k in range(N):
        with torch.cuda.stream(E):
            if mode >= 4 and k >= 4: evG[k - 4].query()
            enc.encode_points_batch((pts, off), out=outs[k % 4])
            if mode >= 1: evE[k].record(E)
        if mode >= 2:
            with torch.cuda.stream(G):
                G.wait_event(evE[k])
                if mode >= 3: small.add_(1.0)
                evG[k].record(G)
    cur.wait_stream(E); cur.wait_stream(G)
    t1.record(cur); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / N * 1e3
for m in range(5): run(m)
for rnd in range(3):
    print("round", rnd, " ".join(f"{'abcde'[m]}={run(m):6.1f}" for m in range(5)), flush=True)
