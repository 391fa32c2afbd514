"""Development probe: CU-masked streams -- encoder(k+1) on most CUs concurrently with GAT(k) on a few."""
import sys, os, statistics, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import torch
import gat_oracle as go
from neural_spectral_codec_amd import synth
from neural_spectral_codec_amd.encoding import SpectralEncoder
from neural_spectral_codec_amd.gnn.model import create_spectral_gnn
from neural_spectral_codec_amd.keyframe import graph_manager as gm

hip = C.CDLL("libamdhip64.so")
def masked_stream(mask_bits):
    words = (C.c_uint32 * 8)()
    for i in range(256):
        if mask_bits(i): words[i // 32] |= (1 << (i % 32))
    s = C.c_void_p()
    r = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert r == 0, r
    return torch.cuda.ExternalStream(s.value)

n, npts = int(os.environ.get('NSC_PROBE_CLOUDS', '1024')), 120000
enc = SpectralEncoder(n_elevation=16).to("cuda")
pts, off = synth.make_clouds_device(n, npts, "cuda")
outs = [torch.empty((n, 800), device="cuda") for _ in range(2)]
m = create_spectral_gnn(edge_dim=2); go.randomize_bn_stats(m); m = m.to("cuda").eval()
graphs = [gm.synthetic_chain_graph(n, device="cuda", seed=1) for _ in range(2)]
for g, o in zip(graphs, outs): g.x = o

def run_serial(reps):
    with torch.no_grad():
        for k in range(reps):
            enc.encode_points_batch((pts, off), out=outs[k & 1]); m(graphs[k & 1])

def make_piped(sE, sG):
    def run(reps):
        evE = [torch.cuda.Event() for _ in range(2)]; evG = [torch.cuda.Event() for _ in range(2)]
        with torch.no_grad():
            for k in range(reps):
                i = k & 1
                with torch.cuda.stream(sE):
                    if k >= 2: sE.wait_event(evG[i])        # GAT(k-2) done reading outs[i]
                    enc.encode_points_batch((pts, off), out=outs[i]); evE[i].record(sE)
                with torch.cuda.stream(sG):
                    sG.wait_event(evE[i]); m(graphs[i]); evG[i].record(sG)
        torch.cuda.current_stream().wait_stream(sE); torch.cuda.current_stream().wait_stream(sG)
    return run

def timeit(fn, reps=40):
    fn(6); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(reps); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

mode = sys.argv[1] if len(sys.argv) > 1 else "8"
def enc_only(stream):
    def run(reps):
        with torch.cuda.stream(stream):
            for k in range(reps): enc.encode_points_batch((pts, off), out=outs[0])
        torch.cuda.current_stream().wait_stream(stream)
    return run
def gat_only(stream):
    def run(reps):
        with torch.no_grad(), torch.cuda.stream(stream):
            for k in range(reps): m(graphs[0])
        torch.cuda.current_stream().wait_stream(stream)
    return run
cfgs = {"serial": run_serial}
if mode.startswith("stride"):
    ge = int(mode[6:])
    inG = lambda i: i % ge == 0
else:
    kk = int(mode)                      # first kk CUs of every 32-bit word group -> kk*8 CUs
    inG = lambda i: (i % 32) < kk
sE = masked_stream(lambda i: not inG(i)); sG = masked_stream(inG)
ncu = sum(1 for i in range(256) if inG(i))
cfgs[f"piped gat on {ncu} CUs ({mode})"] = make_piped(sE, sG)
cfgs[f"encoder only on {256-ncu} CUs"] = enc_only(sE)
cfgs[f"gat only on {ncu} CUs"] = gat_only(sG)
res = {k: [] for k in cfgs}
for rnd in range(5):
    for k, fn in cfgs.items(): res[k].append(timeit(fn))
for k, v in res.items():
    med = statistics.median(v); print(f"{k:34s} {med:7.1f} us/step  {n/med*1e6/1e6:.3f} M kf/s", flush=True)
