"""Stage-1 retrieval timing: W1 of Q queries against a 100 k x 800 database + top-10."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_spectral_codec_amd.retrieval import WassersteinRetriever
n = 100000
r = WassersteinRetriever(device="cuda")
db = torch.rand((n, 800), device="cuda") ** 3
r.add_to_database(db)
for q in (1, 4, 16, 32, 128):
    qs = db[:q].clone()
    for _ in range(3): r.query_batch(qs, 10)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): r.query_batch(qs, 10)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"Q={q}: {us:.1f} us per batch ({us/q:.1f} us per query), database stream {n*3200/us/1e3:.0f} GB/s", flush=True)
