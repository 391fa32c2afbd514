"""N > 1 layout on CPU: world_size-2 gloo.  The sharding / all-gather / halo logic is the product's
(neural-spectral-codec_amd/distributed.py); the compute inside each rank is stood in by the oracle so
the test runs without a GPU.  Checks: gathered descriptors == single-process result, and the
halo-sharded GNN rows == the full-graph forward rows (exactness of the 6-node halo)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import gat_oracle as go
import nsc_oracle as orc
from neural_spectral_codec_amd import distributed as nd
from neural_spectral_codec_amd import synth
from neural_spectral_codec_amd.gnn.model import create_spectral_gnn
from neural_spectral_codec_amd.keyframe import graph_manager as gm


class OracleEncoder:
    alpha = torch.tensor(2.0)            # .alpha.device is how the path finds the encoder's device
    output_dim = 800

    def encode_points_batch(self, clouds, out=None):
        d = torch.from_numpy(np.stack([orc.encode_points(c) for c in clouds]))
        if out is None:
            return d
        out.copy_(d)
        return out


class OracleGnn:
    def __init__(self, model):
        self.model = model

    def __call__(self, data):
        return go.forward_reference(self.model, data)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, pipeline, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        model = create_spectral_gnn(edge_dim=2).eval()
        go.randomize_bn_stats(model)
        poses = synth.make_pose_chain(n_total, 3)
        path = nd.ShardedDescriptorPath(OracleEncoder(), OracleGnn(model), n_total, poses, pipeline=pipeline)
        # serial: 24 / 36 keyframes use the boundary exchange + async all-gather, 21 the padded all-gather;
        # pipelined: one all-gather per step into the rotating slots
        assert path.overlap == (n_total % world == 0 and not pipeline)
        lo, hi = path.lo, path.hi
        clouds = [synth.make_cloud(1000 + i, 1500, "uniform") for i in range(lo, hi)]
        other = [synth.make_cloud(5000 + i, 700, "ring") for i in range(lo, hi)]
        # more steps than rotating buffers; the batches alternate, so a result that a LATER step overwrote
        # (a slot reused too early) would no longer equal the single-process result
        n_steps = nd.ShardedDescriptorPath._PIPE_BUFFERS + 2 if pipeline else 3
        kept = None
        for k in range(n_steps):
            res = path.step(clouds if k % 2 == 0 else other)
            if k == n_steps - 2:
                kept = res                                   # still valid one step later
        path.synchronize()
        last_is_clouds = (n_steps - 1) % 2 == 0
        desc_all, emb = res if last_is_clouds else kept      # the newest result computed from `clouds`
        if pipeline:                                         # (serial mode reuses ONE gathered buffer per step)
            desc_other = (kept if last_is_clouds else res)[0]
            assert not np.array_equal(desc_other.numpy(), desc_all.numpy())
        q.put((rank, lo, hi, desc_all.numpy().copy(), emb.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total,world,pipeline", [(24, 2, False), (21, 2, False), (36, 3, False),
                                                    (24, 2, True), (21, 2, True), (36, 3, True)])
def test_multi_rank_gloo_matches_single_process(n_total, world, pipeline):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, pipeline, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    clouds = [synth.make_cloud(1000 + i, 1500, "uniform") for i in range(n_total)]
    desc = np.stack([orc.encode_points(c) for c in clouds])
    torch.manual_seed(0)
    model = create_spectral_gnn(edge_dim=2).eval()
    go.randomize_bn_stats(model)
    full = gm.build_chain_graph(torch.from_numpy(desc), 5, "cpu", synth.make_pose_chain(n_total, 3))
    ref = go.forward_reference(model, full).numpy()
    covered = 0
    for rank, lo, hi, desc_all, emb in res:
        assert (lo, hi) == nd.shard_range(n_total, rank, world)
        assert np.array_equal(desc_all, desc)                # all-gather reproduces the full matrix
        assert emb.shape == (hi - lo, 800)
        assert np.allclose(emb, ref[lo:hi], rtol=1e-5, atol=1e-6)   # halo of 6 is exact
        covered += hi - lo
    assert covered == n_total


@pytest.mark.parametrize("mode", ["serial", "pipelined"])
def test_world8_kitti00_layout(tmp_path, mode):
    """BASELINE configs[3] in its stated shape: 8 contiguous shards of the 4 541-keyframe set (568 x 5 + 567 x 3 rows:
    the padded all-gather branch, halo windows at 7 interior boundaries), all-gather, halo-sharded GNN, row-sharded
    stage-1 retrieval -- the product's distributed.py / ShardedTwoStageRetrieval on every rank, the oracle standing in
    for the kernels.  The ranks are 8 threads of a child process (torch's in-process group; tests/threaded_cpu_worker.py)."""
    import subprocess
    import sys
    n_total, world = 4541, 8
    out = str(tmp_path / "w8.npz")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "threaded_cpu_worker.py"), str(world), str(n_total), mode, out],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    z = np.load(out)
    clouds = [synth.make_cloud(1000 + i, 400, "uniform") for i in range(n_total)]
    desc = np.stack([orc.encode_points(c) for c in clouds])
    torch.manual_seed(0)
    model = create_spectral_gnn(edge_dim=2).eval()
    go.randomize_bn_stats(model)
    poses = synth.make_pose_chain(n_total, 3)
    full = gm.build_chain_graph(torch.from_numpy(desc), 5, "cpu", poses)
    ref = go.forward_reference(model, full).numpy()
    single = OracleLocalRetriever()
    pos = poses[:, :3, 3].astype(np.float32)
    single.add_to_database(desc, pos)
    qsel = [0, n_total // 3, n_total // 2, n_total - 1]
    want_idx, want_val = single.query_batch(desc[qsel], 10, pos[qsel], 8.0)
    want_idx = torch.where(torch.isinf(want_val), torch.full_like(want_idx, -1), want_idx).numpy()
    covered, sizes = 0, []
    for rank in range(world):
        lo, hi = int(z[f"r{rank}_lo"]), int(z[f"r{rank}_hi"])
        assert (lo, hi) == nd.shard_range(n_total, rank, world)
        assert int(z[f"r{rank}_overlap"]) == 0                      # ragged shards: no two-phase exchange
        assert np.array_equal(z[f"r{rank}_desc_all"], desc), rank   # padded all-gather reproduces the full matrix
        emb = z[f"r{rank}_emb"]
        assert emb.shape == (hi - lo, 800)
        assert np.allclose(emb, ref[lo:hi], rtol=1e-5, atol=1e-6), rank      # halo of 6 is exact at every boundary
        assert np.array_equal(z[f"r{rank}_retr_idx"], want_idx), rank
        assert np.array_equal(z[f"r{rank}_retr_val"], want_val.numpy()), rank
        covered += hi - lo
        sizes.append(hi - lo)
    assert covered == n_total and sizes == [568] * 5 + [567] * 3


def test_shard_ranges_and_halo():
    assert [nd.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [nd.shard_range(4541, r, 8)[1] - nd.shard_range(4541, r, 8)[0] for r in range(8)] == [568] * 5 + [567] * 3
    assert nd.halo_window(100, 40, 60) == (34, 66)
    assert nd.halo_window(100, 0, 13) == (0, 19)
    assert nd.halo_window(100, 90, 100, n_layers=2, temporal_neighbors=7) == (84, 100)
    # single process: all_gather is the identity
    t = torch.rand(5, 800)
    assert nd.all_gather_descriptors(t) is t


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        ps = [torch.nn.Parameter(torch.zeros(3, 5)), torch.nn.Parameter(torch.zeros(7)),
              torch.nn.Parameter(torch.zeros(2, 2))]
        ps[0].grad = torch.full((3, 5), float(rank + 1))
        ps[1].grad = torch.arange(7, dtype=torch.float32) * (rank + 1)
        # ps[2] has no grad: skipped consistently on every rank
        nd.all_reduce_gradients(ps)
        trip = np.arange(30).reshape(10, 3)
        part, wgt = nd.split_triplets(trip, rank, world)
        q.put((rank, ps[0].grad.clone(), ps[1].grad.clone(), part, wgt))
    finally:
        dist.destroy_process_group()


def test_gradient_all_reduce_and_triplet_split():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows, wsum = [], 0.0
    for rank, g0, g1, part, wgt in res:
        assert torch.equal(g0, torch.full((3, 5), 3.0))                 # 1 + 2
        assert torch.equal(g1, torch.arange(7, dtype=torch.float32) * 3)
        rows.append(part)
        wsum += wgt
    assert np.array_equal(np.concatenate(rows), np.arange(30).reshape(10, 3))
    assert abs(wsum - 1.0) < 1e-12


# ---------------------------------------------------------------------------------------------
# stage-1 retrieval over database rows sharded across the ranks (BASELINE configs[3])
# ---------------------------------------------------------------------------------------------
class OracleLocalRetriever:
    """query_batch() of WassersteinRetriever over this rank's rows, computed by the numpy oracle."""

    def __init__(self):
        self.rows, self.pos = None, None

    def add_to_database(self, histograms, positions=None):
        self.rows = np.asarray(histograms, np.float32)
        self.pos = None if positions is None else np.asarray(positions, np.float32)

    def query_batch(self, query_hists, top_k=10, query_positions=None, min_distance=0.0):
        import retrieval_oracle as ro
        q = np.asarray(query_hists, np.float32)
        idx = np.zeros((len(q), top_k), np.int64)
        val = np.zeros((len(q), top_k), np.float32)
        for i in range(len(q)):
            d = ro.batch(q[i], self.rows).astype(np.float32)
            if query_positions is not None and self.pos is not None:
                near = np.linalg.norm(self.pos - np.asarray(query_positions, np.float32)[i], axis=1) < min_distance
                d = np.where(near, np.float32(np.inf), d)
            order = np.lexsort((np.arange(len(d)), d))[:top_k]
            idx[i], val[i] = order, d[order]
        return torch.from_numpy(idx), torch.from_numpy(val)


def _retrieval_case(n_total):
    rng = np.random.default_rng(11)
    base = (rng.random((40, 800)) ** 4).astype(np.float32)
    desc = base[np.arange(n_total) % 40] * (1 + 0.05 * rng.random((n_total, 800)).astype(np.float32))
    desc = (desc / desc.sum(1, keepdims=True)).astype(np.float32)
    desc[7] = desc[3]                                   # exact duplicates: ties must resolve to the smaller index
    desc[n_total - 2] = desc[3]
    pos = (rng.random((n_total, 3)) * 200).astype(np.float32)
    queries = [3, n_total // 2, n_total - 1]
    return desc, pos, queries


def _retrieval_worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from neural_spectral_codec_amd.retrieval import ShardedTwoStageRetrieval
        desc, pos, queries = _retrieval_case(n_total)
        sh = ShardedTwoStageRetrieval(OracleLocalRetriever(), n_total, top_k=10, spatial_filter_distance=50.0)
        sh.add_local_rows(desc[sh.lo:sh.hi], pos[sh.lo:sh.hi])
        idx, val = sh.query_batch(desc[queries], pos[queries])
        q.put((rank, sh.lo, sh.hi, idx.numpy(), val.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total,world", [(64, 2), (37, 3), (9, 2)])
def test_row_sharded_retrieval_matches_single_process(n_total, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_retrieval_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    desc, pos, queries = _retrieval_case(n_total)
    single = OracleLocalRetriever()
    single.add_to_database(desc, pos)
    k = min(10, n_total)
    want_idx, want_val = single.query_batch(desc[queries], k, pos[queries], 50.0)
    want_idx = torch.where(torch.isinf(want_val), torch.full_like(want_idx, -1), want_idx).numpy()
    for rank, lo, hi, idx, val in res:
        assert (lo, hi) == nd.shard_range(n_total, rank, world)
        assert np.array_equal(idx, want_idx), (rank, idx, want_idx)          # identical on every rank, ties included
        assert np.array_equal(val, want_val.numpy())
