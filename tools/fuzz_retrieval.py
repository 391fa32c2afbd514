#!/usr/bin/env python3
"""One-off soak (not part of the test suite): WassersteinRetriever (CDF cache, streaming and register-tiled W1 kernels, device
top-k, spatial filter) on random shapes -- histogram widths 1-1 024 (the limit of the W1 kernels), databases of 1-6 000 rows grown in random chunks, 1-40
queries, k from 1 to the database size, empty histograms -- against oracle/retrieval_oracle.py (numpy).
usage: fuzz_retrieval.py [n_cases]"""
import os
import sys
import time

import numpy as np

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
sys.path.insert(0, os.path.join(R_, "oracle"))
import retrieval_oracle as ro                                                   # noqa: E402
from neural_spectral_codec_amd.retrieval import WassersteinRetriever           # noqa: E402

RTOL = 1e-4
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(5)
t0 = time.time()
worst = 0.0
for ci in range(n_cases):
    dim = int(rng.choice([16, 50, 64, 100, 181, 256, 800, 1024])) if ci % 3 else int(rng.integers(1, 1025))
    n = int(rng.integers(1, 300)) if ci % 4 else int(rng.integers(1000, 6000))
    nq = int(rng.integers(1, 41))
    db = (rng.random((n, dim)) ** 3).astype(np.float32)
    if n > 3:
        db[int(rng.integers(0, n))] = 0.0                         # an empty histogram stays unnormalised
    pos = np.cumsum(rng.normal(0, 1.0, (n, 3)), 0).astype(np.float32)
    q = (rng.random((nq, dim)) ** 3).astype(np.float32)
    qpos = pos[rng.integers(0, n, nq)]
    r = WassersteinRetriever(device="cuda")
    cuts = sorted(set(int(c) for c in rng.integers(0, n + 1, 3)) | {0, n})
    for a, b in zip(cuts[:-1], cuts[1:]):                         # the database grows in chunks (buffers and CDF cache follow)
        if b > a:
            r.add_to_database(db[a:b], positions=pos[a:b])
    assert r.database_size == n
    k = int(rng.integers(1, n + 1)) if ci % 2 else min(10, n)
    for filt in (False, True):
        idx, val = r.query_batch(q, top_k=k, query_positions=qpos if filt else None, min_distance=4.0)
        idx, val = idx.cpu().numpy(), val.cpu().numpy()
        assert idx.shape == (nq, k), (idx.shape, nq, k)
        for j in range(nq):
            d = ro.batch(q[j], db)
            if filt:
                d[np.linalg.norm(pos - qpos[j], axis=1) < 4.0] = np.inf
            o, dv = ro.topk(d, k)
            fin = np.isfinite(dv)
            assert np.allclose(val[j][fin], dv[fin], rtol=RTOL, atol=1e-5), f"case {ci} (dim {dim}, n {n}, nq {nq}, k {k}, filter {filt}) query {j}"
            assert np.isinf(val[j][~fin]).all(), (ci, j)
            close = np.abs(d[idx[j][fin]] - dv[fin]) <= RTOL * np.abs(dv[fin]) + 1e-5     # same set up to float32 near-ties
            assert close.all(), f"case {ci} query {j}: indices"
            if fin.any():
                worst = max(worst, float(np.max(np.abs(val[j][fin] - dv[fin]) / (np.abs(dv[fin]) + 1e-6))))
    if ci % 10 == 9:
        print(f"{ci + 1} cases, worst relative distance error {worst:.1e} ({time.time() - t0:.0f} s)", flush=True)
print(f"TOTAL {n_cases} retrieval cases (widths 1-1 024, 1-6 000 rows, 1-40 queries, k up to the database size, with and without the "
      f"spatial filter): distances within {RTOL} of the numpy oracle, top-k sets identical up to float32 near-ties; worst {worst:.1e}")
