"""N > 1 path THROUGH THE HIP KERNELS (BASELINE configs[3] on one card): world-size 2 and 3, the ranks are fresh child
processes that share cuda:0 and exchange over gloo.  Both implementations of the step that ``bench.py --gpus N`` can
select -- one stream with the overlapped two-phase exchange ("serial"), and the two-stream software pipeline with
the co-resident GNN kernels ("pipelined") -- must reproduce the single-process result BIT FOR BIT: the gathered
descriptor matrix, and the enhanced rows each rank owns (the 6-node halo is exact, the GEMM / aggregation order of a
row does not depend on the shard).

This file sorts first among the GPU tests and its parent process never touches the GPU: the children are plain
``python tests/multirank_worker.py`` processes started before anything in this process initialises HIP.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "multirank_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_world(tmp, tag, world, mode, n_total, timeout=420, points=6000):
    port = _free_port()
    outs = [os.path.join(tmp, f"{tag}_r{r}.npz") for r in range(world)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, WORKER, "--rank", str(r), "--world", str(world), "--port", str(port),
                               "--mode", mode, "--n-total", str(n_total), "--points", str(points), "--out", outs[r]],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=timeout)
            logs.append(out.decode(errors="replace")[-2000:])
    finally:
        for p in procs:                       # exact PIDs only
            if p.poll() is None:
                p.kill()
                p.wait()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} of {tag} failed:\n{logs[r] if r < len(logs) else ''}"
    return [np.load(o) for o in outs]


@pytest.fixture(scope="module")
def single(tmp_path_factory):
    cache = {}

    def get(n_total):
        if n_total not in cache:
            tmp = str(tmp_path_factory.mktemp(f"single{n_total}"))
            cache[n_total] = (_run_world(tmp, "single_serial", 1, "serial", n_total)[0],
                              _run_world(tmp, "single_pipe", 1, "pipelined", n_total)[0])
        return cache[n_total]
    return get


@pytest.mark.gpu
@pytest.mark.parametrize("world,n_total,mode", [(2, 192, "serial"), (2, 192, "pipelined"),
                                                (3, 193, "serial"), (3, 193, "pipelined")])
def test_sharded_path_through_hip_kernels_matches_single_process(tmp_path, single, world, n_total, mode):
    ref, ref_pipe = single(n_total)
    # the two step implementations (LDS-tiled kernels on one stream / LDS-free kernels on two) agree bit for bit
    assert int(ref["coresident"]) == 0 and int(ref_pipe["coresident"]) == 1
    assert np.array_equal(ref["desc_all"].view(np.uint32), ref_pipe["desc_all"].view(np.uint32))
    assert np.array_equal(ref["emb"].view(np.uint32), ref_pipe["emb"].view(np.uint32))
    assert np.isfinite(ref["emb"]).all() and np.abs(ref["desc_all"].sum(1) - 1).max() < 1e-5
    res = _run_world(str(tmp_path), f"w{world}_{mode}", world, mode, n_total)
    covered = 0
    for r, z in enumerate(res):
        lo, hi = int(z["lo"]), int(z["hi"])
        assert z["desc_all"].shape == (n_total, 800) and z["emb"].shape == (hi - lo, 800)
        assert int(z["coresident"]) == (1 if mode == "pipelined" else 0)
        assert np.array_equal(z["desc_all"].view(np.uint32), ref["desc_all"].view(np.uint32)), f"rank {r} gathered matrix"
        assert np.array_equal(z["emb"].view(np.uint32), ref["emb"][lo:hi].view(np.uint32)), f"rank {r} owned rows"
        if mode == "pipelined":      # a result two steps old is still intact (4 buffers in rotation)
            assert np.array_equal(z["desc_all_kept"].view(np.uint32), ref["desc_all"].view(np.uint32))
            assert np.array_equal(z["emb_kept"].view(np.uint32), ref["emb"][lo:hi].view(np.uint32))
        # row-sharded stage-1 retrieval == retrieval over the whole database on one GPU, ties included
        assert np.array_equal(z["retr_idx"], ref["retr_idx"]), f"rank {r} retrieval indices"
        assert np.array_equal(z["retr_val"].view(np.uint32), ref["retr_val"].view(np.uint32)), f"rank {r} distances"
        covered += hi - lo
    assert covered == n_total
    assert (ref["retr_idx"][:, 0] >= 0).all() and np.isfinite(ref["retr_val"][:, 0]).all()


def _run_threads(tmp, tag, world, mode, n_total, points, timeout=900):
    """All ranks as threads of ONE child process (tests/multirank_worker.py --threads): the GPU boxes allow 6 processes
    on the card, BASELINE configs[3] has 8 ranks."""
    pattern = os.path.join(tmp, f"{tag}_r{{rank}}.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, WORKER, "--threads", "--world", str(world), "--mode", mode, "--n-total",
                        str(n_total), "--points", str(points), "--out", pattern], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=timeout)
    assert p.returncode == 0, f"{tag} failed:\n{p.stdout.decode(errors='replace')[-4000:]}"
    return [np.load(pattern.replace("{rank}", str(r))) for r in range(world)]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["serial", "pipelined"])
def test_config3_eight_way_shard_of_4541_keyframes_through_hip_kernels(tmp_path, mode):
    """BASELINE configs[3] in its stated shape on one card: 8 contiguous shards of the 4 541-keyframe set (568 x 5 +
    567 x 3 rows -- ragged: the padded all-gather, halo windows at 7 interior boundaries), the HIP encoder on every
    shard, all-gather, halo-sharded GNN forward through the HIP kernels, row-sharded stage-1 retrieval over the 8
    shards (reference src/retrieval/two_stage_retrieval.py:145-202).  Every rank's gathered matrix, owned embedding
    rows and merged retrieval result equal the single-process run BIT FOR BIT; and that single-process result is
    itself checked against the oracles on sampled rows, so the test is not only a self-comparison."""
    import torch
    import gat_oracle as go
    import nsc_oracle as orc
    from neural_spectral_codec_amd import distributed as nd
    from neural_spectral_codec_amd import synth
    from neural_spectral_codec_amd.gnn.model import create_spectral_gnn
    from neural_spectral_codec_amd.keyframe import graph_manager as gm
    n_total, world, points = 4541, 8, 2000
    port = _free_port()
    ref_out = os.path.join(str(tmp_path), "single.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, WORKER, "--rank", "0", "--world", "1", "--port", str(port), "--mode", mode,
                        "--n-total", str(n_total), "--points", str(points), "--out", ref_out], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
    ref = np.load(ref_out)
    # (1) the single-process result against the oracles: descriptors of sampled keyframes (incl. both sides of shard
    #     boundaries) vs the C restatement, embedding rows vs the float64 GAT restatement on the full 4 541-node graph
    rows = [0, 1, 567, 568, 569, 1135, 1136, 2271, 2272, 2839, 2840, 3406, 3407, 3973, 3974, 4540]
    for i in rows:
        od = orc.encode_points(synth.make_cloud(1000 + i, points, "uniform"))
        assert np.all(np.abs(ref["desc_all"][i] - od) <= 1e-6 * np.abs(od) + 1e-9), i
    torch.manual_seed(0)
    model = create_spectral_gnn(edge_dim=2)
    synth.randomize_bn_stats(model)
    model = model.eval()
    full = gm.build_chain_graph(torch.from_numpy(ref["desc_all"]), 5, "cpu", synth.make_pose_chain(n_total, 3))
    want = go.forward_reference(model, full, dtype=torch.float64)
    go.assert_within_bar(torch.from_numpy(ref["emb"]), want, what="single-process embedding of the 4541-keyframe chain")
    assert (ref["retr_idx"][:, 0] >= 0).all() and np.isfinite(ref["retr_val"][:, 0]).all()
    # (2) the 8-way run against it, bit for bit
    res = _run_threads(str(tmp_path), f"w8_{mode}", world, mode, n_total, points)
    covered, sizes = 0, []
    for r, z in enumerate(res):
        lo, hi = int(z["lo"]), int(z["hi"])
        assert (lo, hi) == nd.shard_range(n_total, r, world)
        assert int(z["overlap"]) == 0                                    # ragged: the padded all-gather branch
        assert int(z["coresident"]) == (1 if mode == "pipelined" else 0)
        assert z["desc_all"].shape == (n_total, 800) and z["emb"].shape == (hi - lo, 800)
        assert np.array_equal(z["desc_all"].view(np.uint32), ref["desc_all"].view(np.uint32)), f"rank {r} gathered matrix"
        assert np.array_equal(z["emb"].view(np.uint32), ref["emb"][lo:hi].view(np.uint32)), f"rank {r} owned rows"
        if mode == "pipelined":
            assert np.array_equal(z["desc_all_kept"].view(np.uint32), ref["desc_all"].view(np.uint32))
            assert np.array_equal(z["emb_kept"].view(np.uint32), ref["emb"][lo:hi].view(np.uint32))
        assert np.array_equal(z["retr_idx"], ref["retr_idx"]), f"rank {r} retrieval indices"
        assert np.array_equal(z["retr_val"].view(np.uint32), ref["retr_val"].view(np.uint32)), f"rank {r} distances"
        covered += hi - lo
        sizes.append(hi - lo)
    assert covered == n_total and sizes == [568] * 5 + [567] * 3


TRAIN_WORKER = os.path.join(HERE, "multirank_train_worker.py")


@pytest.mark.gpu
def test_data_parallel_train_step_through_hip_kernels(tmp_path):
    """The data-parallel half of BASELINE configs[4] (reference src/gnn/trainer.py:186-221 with the triplet batch split
    over the ranks): world 2, ranks as child processes sharing cuda:0 over gloo, each running GNNTrainer.train_batches on
    the full-size dataset of test_config5_full_size_train_step -- replicated 4 541-node graph forward, ITS HALF of the
    1 024-triplet batch through nsc_triplet_loss + nsc_gat_backward, all_reduce_gradients, one Adam step.  The reduced
    gradient and the updated parameters must equal the single-process ones (the only difference is the order in which
    float32 triplet contributions are summed: per rank, then across ranks), and be identical on both ranks."""
    tmp = str(tmp_path)

    def run(world, tag):
        port = _free_port()
        outs = [os.path.join(tmp, f"{tag}_r{r}.npz") for r in range(world)]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, TRAIN_WORKER, "--rank", str(r), "--world", str(world), "--port",
                                   str(port), "--out", outs[r]], env=env, stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT) for r in range(world)]
        logs = []
        try:
            for p in procs:
                o, _ = p.communicate(timeout=600)
                logs.append(o.decode(errors="replace")[-3000:])
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
                    p.wait()
        for r, p in enumerate(procs):
            assert p.returncode == 0, f"rank {r} of {tag} failed:\n{logs[r] if r < len(logs) else ''}"
        return [np.load(o) for o in outs]

    one = run(1, "single")[0]
    two = run(2, "dp2")
    assert np.array_equal(one["trip"], two[0]["trip"]) and np.array_equal(one["trip"], two[1]["trip"])
    keys = [k[2:] for k in one.files if k.startswith("g:")]
    assert len(keys) >= 20
    gscale = max(np.abs(one["g:" + k]).max() for k in keys)
    worst = 0.0
    # exactly-zero gradients (float32 noise on both sides): a bias in front of a batch-statistics BatchNorm; and, because
    # the triplet gradient sums to zero over the rows, output_proj.bias and the last BatchNorm's bias
    zero = {"input_proj.bias", "output_proj.bias", "batch_norms.2.bias"} | {f"convs.{l}.bias" for l in range(3)}
    for k in keys:
        a, b0, b1 = one["g:" + k], two[0]["g:" + k], two[1]["g:" + k]
        assert np.array_equal(b0, b1), f"{k}: the ranks disagree after the all-reduce"
        if k in zero:
            assert np.abs(a).max() < 1e-3 * gscale and np.abs(b0).max() < 1e-3 * gscale, k
            continue
        # float32 sums of the same terms in a different grouping: relative to the layer's own gradient scale
        d = np.abs(a.astype(np.float64) - b0).max() / max(np.abs(a).max(), 1e-6 * gscale)
        worst = max(worst, d)
        assert d <= 2e-4, (k, d)
        pa, pb0, pb1 = one["p:" + k], two[0]["p:" + k], two[1]["p:" + k]
        assert np.array_equal(pb0, pb1), f"{k}: parameters differ between the ranks after the step"
        # first Adam step = lr * sign(g) where |g| >> eps: identical updates wherever the gradient sign is well determined
        big = np.abs(a) > 1e-3 * np.abs(a).max()
        assert np.allclose(pa[big], pb0[big], rtol=0, atol=2e-5), k
    print(f"data-parallel vs single-process reduced gradient: worst relative max-norm distance {worst:.2e}")
    # each rank reports its weighted share (slice size / batch size) of the mean loss: the shares add up to it
    assert abs(float(one["loss"]) - (float(two[0]["loss"]) + float(two[1]["loss"]))) <= 1e-4 * abs(float(one["loss"])) + 1e-7
    assert np.allclose(one["bn_mean"], two[0]["bn_mean"], rtol=1e-6, atol=1e-7)       # replicated forward: same statistics


@pytest.mark.gpu
def test_data_parallel_captured_step_replays_under_a_process_group(tmp_path):
    """The captured training step under an initialised 2-rank group (advisor, round 3): 9 full batches of 128 triplets + a
    ragged tail = three optimizer steps, so the capture (second full batch) is REPLAYED seven times between all-reduces;
    forced on (`use_graph=True`: captures open in thread-local error mode) against off (the default under a > 1-rank group).
    Both runs: identical parameters on the two ranks; graph vs eager: the same parameters up to Adam's +-lr on noise-only
    elements; only full batches are captured (one capture, the tail runs eagerly)."""
    tmp = str(tmp_path)

    def run(tag, use_graph):
        port = _free_port()
        outs = [os.path.join(tmp, f"{tag}_r{r}.npz") for r in range(2)]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, TRAIN_WORKER, "--rank", str(r), "--world", "2", "--port", str(port), "--out", outs[r],
                                   "--use-graph", str(use_graph), "--batches", "9", "--batch-size", "128"], env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
        logs = []
        try:
            for p in procs:
                o, _ = p.communicate(timeout=600)
                logs.append(o.decode(errors="replace")[-3000:])
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
                    p.wait()
        for r, p in enumerate(procs):
            assert p.returncode == 0, f"rank {r} of {tag} failed:\n{logs[r] if r < len(logs) else ''}"
        return [np.load(o) for o in outs]

    g = run("graph", 1)
    e = run("eager", -1)
    assert int(g[0]["captured"]) == 1 and int(g[1]["captured"]) == 1 and not int(g[0]["capture_failed"])
    assert int(e[0]["captured"]) == 0                                  # the default under a 2-rank group: no capture
    keys = [k[2:] for k in g[0].files if k.startswith("p:")]
    lr = 5e-4
    for k in keys:
        assert np.array_equal(g[0]["p:" + k], g[1]["p:" + k]), f"{k}: ranks differ (captured)"
        assert np.array_equal(e[0]["p:" + k], e[1]["p:" + k]), f"{k}: ranks differ (eager)"
        assert np.isfinite(g[0]["p:" + k]).all()
        # three Adam steps: an element whose gradient is rounding noise steps by +-lr per step on either side
        assert np.abs(g[0]["p:" + k] - e[0]["p:" + k]).max() <= 6.3 * lr, k
        # the gradients of the LAST optimizer step (after the all-reduce): finite, of the eager run's magnitude (round 4: a memset
        # node of the captured graph left them at 1e25-1e32, which Adam's normalisation hid from the parameter check above)
        ga, gb = e[0]["g:" + k], g[0]["g:" + k]
        assert np.isfinite(gb).all() and np.abs(gb).max() <= 3.0 * np.abs(ga).max() + 1e-6, k
    # the mean loss over the ten batches: after the first optimizer step the two runs' weights differ by +-lr on noise-only
    # elements, the later batches' losses follow (measured: 3.7e-4 relative)
    assert abs(float(g[0]["loss"]) - float(e[0]["loss"])) <= 2e-3 * abs(float(e[0]["loss"]))


@pytest.mark.gpu
def test_bench_launches_its_own_ranks(tmp_path):
    """``python bench.py --gpus 2`` with no WORLD_SIZE in the environment -- the shape of the driver's N = 1 command -- must
    run by itself: the parent starts two fresh ranks (it never initialises HIP), relays ONE well-formed line and exits 0.
    NSC_BENCH_REHEARSAL=1: both ranks share cuda:0 and exchange over gloo (one-GPU box)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(NSC_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    bench = os.path.join(os.path.dirname(HERE), "bench.py")
    p = subprocess.Popen([sys.executable, bench, "--gpus", "2", "--steps", "6", "--warmup", "2", "--clouds", "96",
                          "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=str(tmp_path))
    try:
        out, err = p.communicate(timeout=900)
    finally:
        if p.poll() is None:
            p.kill()
            p.wait()
    assert p.returncode == 0, err.decode(errors="replace")[-3000:]
    lines = [ln for ln in out.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["backend"] == "gloo"
    assert line["steps"] == 6 and line["warmup"] == 2 and line["scaling"] == "weak"
    assert len(line["ms_per_step_by_rank"]) == 2 and all(v > 0 for v in line["ms_per_step_by_rank"])
    assert line["launched_by"].startswith("bench.py itself")
    assert line["calibration"] is not None and line["allgather"]["alone_ms"] > 0 and line["allgather"]["events"] == 6
    assert abs(line["value"] - 2 * 96 * 6 / (line["ms_per_step"] * 6e-3)) <= 1e-6 * line["value"]
    assert "REHEARSAL" in line["data"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["serial", "pipelined"])
def test_equal_shards_at_the_bench_shape_through_hip_kernels(tmp_path, mode):
    """The branch ``bench.py --gpus N`` takes: EQUAL shards of 1 024 keyframes per rank (world 4 here, child processes over
    gloo, 2 000-point clouds).  One stream ("serial"): the overlapped two-phase exchange -- boundary rows first, the big
    all-gather asynchronous under the GNN forward (``overlap`` must be on); pipelined: ONE all-gather per step into the
    rotating gathered-matrix slots.  Every rank's gathered matrix, owned rows (also those kept from two steps before) and
    merged retrieval result equal the single process, bit for bit."""
    world, n_total, pts = 4, 4096, 2000
    ref = _run_world(str(tmp_path), "single", 1, mode, n_total, points=pts, timeout=600)[0]
    res = _run_world(str(tmp_path), f"w{world}_{mode}", world, mode, n_total, points=pts, timeout=600)
    assert np.isfinite(ref["emb"]).all() and np.abs(ref["desc_all"].sum(1) - 1).max() < 1e-5
    for r, z in enumerate(res):
        lo, hi = int(z["lo"]), int(z["hi"])
        assert hi - lo == 1024
        assert int(z["overlap"]) == (1 if mode == "serial" else 0)        # (pipeline mode hides the whole exchange instead)
        assert np.array_equal(z["desc_all"].view(np.uint32), ref["desc_all"].view(np.uint32)), f"rank {r} gathered matrix"
        assert np.array_equal(z["emb"].view(np.uint32), ref["emb"][lo:hi].view(np.uint32)), f"rank {r} owned rows"
        if mode == "pipelined":
            assert np.array_equal(z["desc_all_kept"].view(np.uint32), ref["desc_all"].view(np.uint32))
            assert np.array_equal(z["emb_kept"].view(np.uint32), ref["emb"][lo:hi].view(np.uint32))
        assert np.array_equal(z["retr_idx"], ref["retr_idx"]) and np.array_equal(z["retr_val"].view(np.uint32),
                                                                                  ref["retr_val"].view(np.uint32))


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_multi_rank_step_over_a_one_rank_rccl_group(tmp_path):
    """The N > 1 step with the backend the real run uses: a ONE-rank RCCL process group on the one card of the box
    (`rehearse_collectives=True` issues every collective of the multi-rank step).  What gloo cannot show: nccl orders a
    collective on the stream it is issued on and `wait()` does not block the host.  Every form of the step -- two-phase
    exchange, one gather, pipelined on two / one encoder streams -- equals the exchange-free path bit for bit
    (tests/rccl_world1_worker.py)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(HERE, "rccl_world1_worker.py"), str(_free_port())], env=env,
                       capture_output=True, timeout=900, cwd=str(tmp_path))
    assert p.returncode == 0 and b"RCCL_WORLD1_OK" in p.stdout, (p.stdout.decode(errors="replace")[-2000:],
                                                                  p.stderr.decode(errors="replace")[-3000:])


@pytest.mark.gpu
def test_bench_in_the_multi_rank_shape_over_a_one_rank_rccl_group(tmp_path):
    """bench.py's N > 1 code path -- calibration over three step forms with its all-reduce, barriers, HIP events around every
    step's all-gather, the all-gather alone, the per-rank table -- with every collective through RCCL (NSC_BENCH_RCCL_WORLD1=1:
    one rank, nccl backend), and the descriptors that went through the gather checked against the oracle."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(NSC_BENCH_RCCL_WORLD1="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(_free_port()))
    bench = os.path.join(os.path.dirname(HERE), "bench.py")
    p = subprocess.run([sys.executable, bench, "--gpus", "1", "--steps", "12", "--warmup", "2", "--clouds", "256"], env=env,
                       capture_output=True, timeout=900, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                 # ONE line on stdout: RCCL's version banner (printed on stdout) went to stderr
    assert b"RCCL version" in p.stderr
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["backend"] == "nccl"
    assert line["calibration"] is not None and set(k for k in line["calibration"] if k.endswith("_ms_per_step")) == {
        "pipelined2_ms_per_step", "pipelined1_ms_per_step", "serial_ms_per_step"}
    assert line["allgather"]["events"] == 12 and line["allgather"]["in_situ_ms"] > 0 and line["allgather"]["alone_ms"] > 0
    assert len(line["ms_per_step_by_rank"]) == 1
    assert line["parity"]["ok"] is True
    assert "RCCL" in line["data"] and "rehearsal" in line["launched_by"]
