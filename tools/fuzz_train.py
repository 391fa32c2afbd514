#!/usr/bin/env python3
"""One-off soak (not part of the test suite): the training step -- forward_train, triplet loss, backward with its end-of-backward
batched reductions -- on random graphs against torch autograd through the CPU restatement (oracle/gat_oracle.py): loss, every
parameter gradient and the input gradient held to the FLOAT32 restatement's own distance from the float64 one (3 x, floor 1e-3).

Graph kinds: the temporal chain (banded), chain + loop closures, hubs with duplicate edges, sparse random; 20-3 000 nodes (both sides
of the split-K threshold of 512), edge_dim 2 / None, hidden 64 / 256, 1-4 layers, residual_proj now and then; every 5th case runs
the backward twice into the same buffers (exactly twice the gradient).

Ties.  The network has kinks (ReLU after BatchNorm, LeakyReLU on the attention logits): where a pre-activation is ~1e-8, two float32
evaluations that round differently (the kernels and torch) put it on different sides, and the gradients differ by 1e-3 in one
channel and 1e-4 downstream -- an error of neither.  (The kernels' own forward and backward evaluate the SAME expression, bit for
bit.)  A case that misses the bar is therefore repeated with re-drawn input features (same graph, weights, triplets): a tie does not
survive that, an indexing or accumulation error does.  A case counts as failed when it misses the bar on all three draws.
usage: fuzz_train.py [n_cases]"""
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
sys.path.insert(0, os.path.join(R_, "oracle"))
sys.path.insert(0, os.path.join(R_, "tests"))
import gat_oracle as go                                                          # noqa: E402
from neural_spectral_codec_amd.gnn.model import SpectralGNN                     # noqa: E402
from neural_spectral_codec_amd.gnn.trainer import TripletLoss                   # noqa: E402


def rel(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def run_case(ci, draw):
    """One case (draw > 0: other input features).  Returns (failure text or None, worst figures, description)."""
    rng = np.random.default_rng(1000 + ci)
    kind = ci % 4
    n = int(rng.integers(20, 500)) if ci % 3 else int(rng.integers(520, 3000))
    edge_dim = None if ci % 5 == 3 else 2
    hidden = 64 if ci % 2 else 256
    L = int(rng.integers(1, 5))
    in_dim, out_dim = (800, 800) if ci % 6 else (64, 96)
    torch.manual_seed(ci)
    m = SpectralGNN(input_dim=in_dim, hidden_dim=hidden, output_dim=out_dim, n_layers=L, dropout=0.0, residual=True, edge_dim=edge_dim)
    go.randomize_bn_stats(m, ci + 1)
    with torch.no_grad():
        for c in m.convs:
            c.bias.normal_(0, 0.1)
    m = m.to("cuda")
    i = np.arange(n - 1)
    chain = [np.concatenate([i, i + 1, i[:-1], i[:-1] + 2]), np.concatenate([i + 1, i, i[:-1] + 2, i[:-1]])]
    if kind == 0:                                                # the temporal chain (takes the banded CSR)
        src, dst = chain
    elif kind == 1:                                              # chain + loop closures
        lc = rng.integers(0, n, (2, max(n // 12, 1)))
        src, dst = np.concatenate([chain[0], lc[0], lc[1]]), np.concatenate([chain[1], lc[1], lc[0]])
    elif kind == 2:                                              # hubs + duplicate edges + explicit self loops
        hubs = rng.integers(0, n, 2)
        e = int(rng.integers(n, 3 * n + 80))
        src = np.concatenate([chain[0], rng.integers(0, n, e), np.arange(0, n, 7)])
        dst = np.concatenate([chain[1], np.where(rng.random(e) < 0.5, rng.choice(hubs, e), rng.integers(0, n, e)), np.arange(0, n, 7)])
        src, dst = np.concatenate([src, src[:30]]), np.concatenate([dst, dst[:30]])
    else:                                                        # sparse random (isolated nodes: only their self loop)
        e = int(rng.integers(0, 3 * n))
        src, dst = rng.integers(0, n, e), rng.integers(0, n, e)
    ei = torch.from_numpy(np.stack([src, dst]).astype(np.int64))
    x = torch.rand((n, in_dim), generator=torch.Generator().manual_seed(5000 + 17 * ci + draw)) ** 4
    x = x / x.sum(1, keepdim=True)
    ea = torch.rand((ei.shape[1], 2), generator=torch.Generator().manual_seed(ci)) if edge_dim else None
    g = SimpleNamespace(x=x.cuda(), edge_index=ei.cuda(), edge_attr=None if ea is None else ea.cuda(), num_nodes=n)
    T = int(rng.integers(8, 600))
    trip = np.stack([rng.integers(0, n, T) for _ in range(3)], 1)
    tt = torch.from_numpy(trip)
    Rm = torch.randn(n, out_dim, generator=torch.Generator().manual_seed(ci)) * 1e-3
    what = f"case {ci} (kind {kind}, n {n}, L {L}, hidden {hidden}, edge {edge_dim}, T {T}, in/out {in_dim}/{out_dim}, draw {draw})"

    def loss_fn(e_):
        return go.triplet_loss_reference(e_, tt[:, 0], tt[:, 1], tt[:, 2], 0.1) + (e_ * Rm.to(e_.dtype)).sum()
    _, grads32, gx32, _ = go.reference_gradients(m, g, loss_fn)
    _, grads64, gx64, loss64 = go.reference_gradients(m, g, loss_fn, dtype=torch.float64)   # the yardstick: float32's own distance
    m.train()
    g.x.requires_grad_(True)
    reps = 2 if ci % 5 == 4 else 1
    for _ in range(reps):
        emb = m(g)
        loss = TripletLoss(margin=0.1).forward_indexed(emb, trip[:, 0], trip[:, 1], trip[:, 2]) + (emb * Rm.cuda()).sum()
        loss.backward()
    fig = {"loss": abs(loss.item() - loss64.item()) / (abs(loss64.item()) + 1e-6), "grad": 0.0, "gx": 0.0}
    if fig["loss"] > 2e-4:
        return f"{what}: loss {loss.item()} vs {loss64.item()}", fig, what
    params = dict(m.named_parameters())
    gscale = max(v.abs().max().item() for v in grads64.values())
    floor = 1e-3 * max(1, L // 2)
    table, bad = [], None
    for k, ref in grads64.items():
        if k not in params:
            continue
        got = params[k].grad.detach().cpu().reshape(ref.shape).double() / reps
        if not torch.isfinite(got).all():
            return f"{what}: {k} is not finite", fig, what
        if ref.abs().max().item() < 1e-3 * gscale:               # (zero in exact arithmetic: the biases in front of a BatchNorm)
            if got.abs().max().item() >= 2e-3 * gscale * max(1, L // 2):
                bad = bad or f"{k}: {got.abs().max().item():.2e} where the exact gradient is zero (scale {gscale:.2e})"
            continue
        r, r32 = rel(got, ref), rel(grads32[k].double(), ref)
        table.append(f"   {k:34s} kernels {r:.2e}   float32 restatement {r32:.2e}   max |ref| {ref.abs().max().item():.2e}")
        fig["grad"] = max(fig["grad"], r)
        if not r < max(floor, 3 * r32):
            bad = bad or f"{k} rel {r:.2e} (float32 restatement {r32:.2e})"
    r, r32 = rel(g.x.grad.cpu().double() / reps, gx64), rel(gx32.double(), gx64)
    fig["gx"] = r
    if not r < max(floor, 3 * r32):
        bad = bad or f"input gradient rel {r:.2e} (float32 restatement {r32:.2e})"
    if bad:
        return f"{what}: {bad}\n" + "\n".join(table), fig, what
    return None, fig, what


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
worst = {"loss": 0.0, "grad": 0.0, "gx": 0.0}
ties = []
t0 = time.time()
for ci in range(n_cases):
    fails = []
    for draw in range(3):
        err, fig, what = run_case(ci, draw)
        if err is None:
            break
        fails.append(err)
    if len(fails) == 3:
        print("\n\n".join(fails))
        raise AssertionError(f"case {ci} misses the bar on three draws of its input features")
    if fails:
        ties.append((ci, len(fails)))
    for k in worst:
        worst[k] = max(worst[k], fig[k])
    if ci % 10 == 9:
        print(f"{ci + 1} cases, worst: loss {worst['loss']:.1e}, parameter gradient {worst['grad']:.1e}, input gradient {worst['gx']:.1e}; "
              f"ties so far {ties} ({time.time() - t0:.0f} s)", flush=True)
print(f"ties (case, draws that met one): {ties}")
print(f"TOTAL {n_cases} training cases: loss, every parameter gradient and the input gradient as close to the float64 restatement as the "
      f"float32 restatement is (3 x, floor 1e-3 per two layers); worst relative on the passing draws: loss {worst['loss']:.1e}, parameter "
      f"gradient {worst['grad']:.1e}, input gradient {worst['gx']:.1e}; {len(ties)} cases met a ReLU / LeakyReLU tie on some draw")
