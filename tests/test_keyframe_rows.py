"""descriptor wire format, chain-graph builder and voxel-IoU.

CPU part: the oracle restatements against the fixtures generated from the reference itself
(tests/golden/keyframe.npz, oracle/gen_golden_keyframe.py).  GPU part: the HIP kernels through the C ABI
against the fixtures and the oracle -- bit-exact for the uint16 / byte / index / count outputs, 2 ulp
for the two float32 edge features (device log1p / acos vs numpy's, see test_chain_graph_gpu)."""
import os

import numpy as np
import pytest
import torch

import keyframe_oracle as ko

GOLD = os.path.join(os.path.dirname(__file__), "golden", "keyframe.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _overlap_keys(g):
    return sorted(k[:-4] for k in g.files if k.endswith("_iou"))


# ------------------------------------------------------------------------------------------- CPU
def test_pairwise_sum_is_numpy_order():
    rng = np.random.default_rng(3)
    for n in list(range(1, 140)) + [255, 256, 257, 799, 800, 801, 1024, 2049, 4096]:
        a = (rng.random(n) ** 3).astype(np.float32)
        assert ko.pairwise_sum_f32(a) == a.sum(), n


@pytest.mark.parametrize("nb", [50, 800])
def test_oracle_quantizer_matches_reference(gold, nb):
    for i, h in enumerate(gold[f"q{nb}_hist"]):
        q = ko.quantize(h)
        assert (q == gold[f"q{nb}_quant"][i]).all(), i
        assert (_bits(ko.dequantize(q)) == _bits(gold[f"q{nb}_deq"][i])).all(), i
    for i, a in enumerate(gold[f"q{nb}_arb"]):
        assert (_bits(ko.dequantize(a)) == _bits(gold[f"q{nb}_arb_deq"][i])).all(), i


def test_oracle_records_match_reference(gold):
    for i in range(len(gold["rec_bytes"])):
        b = ko.pack_record(gold["q50_quant"][i], gold["rec_pose7"][i], float(gold["rec_ts"][i]),
                           int(gold["rec_id"][i]), gold["rec_hash"][i].tobytes())
        assert np.frombuffer(b, np.uint8).tolist() == gold["rec_bytes"][i].tolist()
        q, pose, ts, kid, hsh = ko.unpack_record(b, 50)
        assert (q == gold["q50_quant"][i]).all() and ts == gold["rec_ts"][i] and kid == gold["rec_id"][i]
        assert hsh == gold["rec_hash"][i].tobytes()
        assert (pose == gold["rec_pose7"][i].astype(np.float32)).all()


def test_oracle_overlap_matches_reference(gold):
    keys = _overlap_keys(gold)
    assert len(keys) >= 12
    for k in keys:
        iou, counts = ko.voxel_overlap(gold[k + "_p1"], gold[k + "_p2"], gold[k + "_T"], float(gold[k + "_voxel"]))
        assert iou == float(gold[k + "_iou"]), (k, iou, counts)


def _poses(n, seed=0):
    from neural_spectral_codec_amd import synth
    return synth.make_pose_chain(n, seed)


def test_host_graph_builder_matches_literal_loop():
    from neural_spectral_codec_amd.keyframe.graph_manager import build_chain_graph
    for n, m, loops in [(1, 5, None), (2, 5, None), (3, 5, None), (40, 5, [(3, 30), (50, 2), (7, 7)]), (25, 7, None),
                        (9, 1, None)]:
        poses = _poses(n, n)
        ei, ea = ko.chain_graph_loop(n, m, poses, loops)
        g = build_chain_graph(torch.zeros((n, 4)), m, "cpu", poses, loops)
        assert g.edge_index.shape == (2, ei.shape[1])
        assert (g.edge_index.numpy() == ei).all()
        if ea is None:
            assert g.edge_attr is None
        else:
            assert np.abs(g.edge_attr.numpy() - ea).max() <= 2e-7


def test_build_graph_from_keyframes_host():
    from types import SimpleNamespace
    from neural_spectral_codec_amd.keyframe.graph_manager import build_graph_from_keyframes
    rng = np.random.default_rng(0)
    kfs = [SimpleNamespace(descriptor=rng.random(800).astype(np.float32)) for _ in range(7)]
    g = build_graph_from_keyframes(kfs, temporal_neighbors=5)
    assert tuple(g.x.shape) == (7, 800) and g.edge_attr is None
    assert (g.edge_index.numpy() == ko.chain_graph_loop(7, 5)[0]).all()
    assert build_graph_from_keyframes([], 5) is None


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("nb", [50, 800])
def test_quantizer_gpu_bit_exact(gold, nb):
    from neural_spectral_codec_amd.encoding import quantization as qz
    h = torch.from_numpy(gold[f"q{nb}_hist"]).cuda()
    q = qz.quantize_batch(h)
    assert q.dtype == torch.uint16
    assert (q.cpu().numpy() == gold[f"q{nb}_quant"]).all()
    d = qz.dequantize_batch(q)
    assert (_bits(d.cpu().numpy()) == _bits(gold[f"q{nb}_deq"])).all()
    arb = torch.from_numpy(gold[f"q{nb}_arb"]).cuda()
    assert (_bits(qz.dequantize_batch(arb).cpu().numpy()) == _bits(gold[f"q{nb}_arb_deq"])).all()
    # the reference's per-histogram numpy API
    hq = qz.HistogramQuantizer(n_bins=nb)
    one = hq.quantize(gold[f"q{nb}_hist"][0])
    assert isinstance(one, np.ndarray) and one.dtype == np.uint16 and (one == gold[f"q{nb}_quant"][0]).all()
    assert (_bits(hq.dequantize(one)) == _bits(gold[f"q{nb}_deq"][0])).all()


@pytest.mark.gpu
@pytest.mark.parametrize("nb", [1, 7, 8, 129, 360, 800, 1000, 4096])
def test_quantizer_gpu_vs_oracle_shapes(nb):
    from neural_spectral_codec_amd.encoding import quantization as qz
    rng = np.random.default_rng(nb)
    n = 37 if nb > 1000 else 203                                     # not a multiple of the 4 rows per workgroup
    h = (rng.random((n, nb)) ** 4).astype(np.float32)
    h[: n // 2] /= h[: n // 2].sum(1, keepdims=True)
    h[-1] = 0.0
    q = qz.quantize_batch(torch.from_numpy(h).cuda()).cpu().numpy()
    want = np.stack([ko.quantize(r) for r in h])
    assert (q == want).all()
    normal = want.astype(np.int64).sum(1) > 0
    assert (q.astype(np.int64).sum(1)[normal] == 65535).all()         # the invariant the format promises
    arb = rng.integers(0, 65536, (n, nb)).astype(np.uint16)
    d = qz.dequantize_batch(torch.from_numpy(arb).cuda()).cpu().numpy()
    assert (_bits(d) == _bits(np.stack([ko.dequantize(r) for r in arb]))).all()


@pytest.mark.gpu
def test_quantizer_full_batch_round_trip():
    """1 024 x 800 (the batch of BASELINE configs[1]): sum preserved, error below one quantum."""
    from neural_spectral_codec_amd.encoding import quantization as qz
    g = torch.Generator().manual_seed(5)
    h = torch.rand((1024, 800), generator=g) ** 4
    h = (h / h.sum(1, keepdim=True)).cuda()
    q = qz.quantize_batch(h)
    assert (q.cpu().numpy().astype(np.int64).sum(1) == 65535).all()
    d = qz.dequantize_batch(q)
    err = (d - h).abs()
    # one bin per row (the first largest quantised one) absorbs the summed rounding error of the other 799
    assert float(err.max()) <= 400.5 / 65535
    assert int((err > 0.5 / 65535 + 1e-7).sum(1).max()) <= 1
    assert torch.allclose(d.sum(1), torch.ones(1024, device="cuda"), atol=1e-5)
    with pytest.raises(Exception):
        qz.quantize_batch(h.cpu())
    with pytest.raises(Exception):
        qz.quantize_batch(torch.zeros((2, 5000), device="cuda"))      # NSC_EUNSUPPORTED: dim > 4096


@pytest.mark.gpu
def test_records_gpu_match_reference_bytes(gold):
    from neural_spectral_codec_amd.encoding import quantization as qz
    n = len(gold["rec_bytes"])
    rec = qz.pack_records(torch.from_numpy(gold["q50_quant"][:n]).cuda(), torch.from_numpy(gold["rec_pose7"]),
                          torch.from_numpy(gold["rec_ts"]), torch.from_numpy(gold["rec_id"].astype(np.int64)),
                          torch.from_numpy(gold["rec_hash"]))
    assert rec.shape == (n, 220)
    assert (rec.cpu().numpy() == gold["rec_bytes"]).all()
    q, p7, ts, ids, hs = qz.unpack_records(rec, 50)
    assert (q.cpu().numpy() == gold["q50_quant"][:n]).all()
    assert (p7.cpu().numpy() == gold["rec_pose7"].astype(np.float32)).all()
    assert (ts.cpu().numpy() == gold["rec_ts"]).all()
    assert (ids.cpu().numpy() == gold["rec_id"]).all()
    assert (hs.cpu().numpy() == gold["rec_hash"]).all()
    # dataclass API of the reference, and the 800-bin generalisation against the oracle
    d = qz.CompressedDescriptor(histogram=gold["q50_quant"][2], pose=gold["rec_pose7"][2],
                                timestamp=float(gold["rec_ts"][2]), keyframe_id=int(gold["rec_id"][2]),
                                point_cloud_hash=gold["rec_hash"][2].tobytes())
    b = d.to_bytes()
    assert np.frombuffer(b, np.uint8).tolist() == gold["rec_bytes"][2].tolist()
    back = qz.CompressedDescriptor.from_bytes(b)
    assert back.keyframe_id == int(gold["rec_id"][2]) and back.timestamp == float(gold["rec_ts"][2])
    assert back.point_cloud_hash == gold["rec_hash"][2].tobytes() and (back.histogram == gold["q50_quant"][2]).all()
    d8 = qz.CompressedDescriptor(histogram=gold["q800_quant"][1], pose=gold["rec_pose7"][1], timestamp=12.5,
                                 keyframe_id=4000000000, point_cloud_hash=bytes(range(20)))
    b8 = d8.to_bytes()
    assert len(b8) == 1720
    assert b8 == ko.pack_record(gold["q800_quant"][1], gold["rec_pose7"][1], 12.5, 4000000000, bytes(range(20)))


@pytest.mark.gpu
def test_compress_decompress_api():
    from neural_spectral_codec_amd.encoding import quantization as qz
    rng = np.random.default_rng(0)
    h = (rng.random(800) ** 4).astype(np.float32)
    h /= h.sum()
    pose = _poses(4, 1)[3]
    pts = rng.normal(size=(100, 4)).astype(np.float32)
    d = qz.compress_descriptor(h, pose, 3.25, 17, pts)
    assert (d.histogram == ko.quantize(h)).all() and len(d.point_cloud_hash) == 20
    h2, pose2, ts, kid = qz.decompress_descriptor(qz.CompressedDescriptor.from_bytes(d.to_bytes()))
    assert ts == 3.25 and kid == 17
    assert np.abs(h2 - h).max() <= 400.5 / 65535
    assert np.abs(pose2 - pose).max() < 1e-6


def _ulp_diff(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b).max() if a.size else 0


def _ulp_rows(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b)


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,loops", [(1, 5, None), (2, 5, None), (3, 5, [(0, 2)]), (4, 5, None), (5, 5, None),
                                       (300, 5, [(3, 250), (400, 2), (10, 10), (299, 0)]), (64, 9, None),
                                       (33, 1, [(1, 30)]), (4541, 5, None)])
def test_chain_graph_gpu(n, m, loops):
    from neural_spectral_codec_amd.keyframe.graph_manager import build_chain_graph
    poses = _poses(n, 11 + n)
    ei, ea = ko.chain_graph_loop(n, m, poses, loops)
    g = build_chain_graph(torch.zeros((n, 8)), m, "cuda", poses, loops)
    assert g.edge_index.dtype == torch.int64 and tuple(g.edge_index.shape) == (2, ei.shape[1])
    assert (g.edge_index.cpu().numpy() == ei).all()
    if ea is None:
        assert g.edge_attr is None or g.edge_attr.numel() == 0
    else:
        got = g.edge_attr.cpu().numpy()
        assert got.shape == ea.shape
        # numpy's own float32 log1p / float64 arccos (SVML on AVX-512 hosts) are not correctly rounded: its
        # array and scalar loops already differ by 1 ulp from each other.  The device rounds the float64 result
        # once, so it sits within 1 ulp of numpy before the /5 and /pi, 2 ulp after.  Angles of (numerically)
        # identical rotations are 0 or ~1e-8 rad depending on the last bit of the trace: absolute bound there.
        close = np.abs(got - ea) <= 1e-7
        assert (close | (_ulp_rows(got, ea) <= 2)).all(), _ulp_diff(got, ea)
    # no poses -> no edge_attr (online path, graph_manager.py:124-128)
    g2 = build_chain_graph(torch.zeros((n, 8)), m, "cuda", None, loops)
    assert g2.edge_attr is None and (g2.edge_index.cpu().numpy() == ei).all()


@pytest.mark.gpu
def test_chain_graph_feeds_the_gnn():
    """The device-built graph drives SpectralGNN.forward to the same embeddings as the host-built one."""
    from neural_spectral_codec_amd.gnn.model import create_spectral_gnn
    from neural_spectral_codec_amd.keyframe.graph_manager import build_chain_graph
    torch.manual_seed(0)
    n = 257
    x = torch.rand((n, 800)) ** 4
    x = x / x.sum(1, keepdim=True)
    poses = _poses(n, 2)
    model = create_spectral_gnn(edge_dim=2).cuda().eval()
    with torch.no_grad():
        a = model(build_chain_graph(x, 5, "cuda", poses))
        b = model(build_chain_graph(x, 5, "cpu", poses).to("cuda"))
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_voxel_overlap_gpu_matches_reference(gold):
    from neural_spectral_codec_amd.data import pose_utils as pu
    keys = _overlap_keys(gold)
    for k in keys:                                                     # one pair per call: the reference's signature
        iou = pu.compute_overlap(gold[k + "_p1"], gold[k + "_p2"], gold[k + "_T"], voxel_size=float(gold[k + "_voxel"]))
        assert isinstance(iou, float) and iou == float(gold[k + "_iou"]), k
    # all pairs of one column count and voxel size in one launch, counts against the oracle
    for cols, vox in ((3, 0.2), (4, 0.2), (3, 0.5)):
        ks = [k for k in keys if gold[k + "_p1"].shape[1] == cols and float(gold[k + "_voxel"]) == vox]
        iou, counts = pu.compute_overlap_batch([gold[k + "_p1"] for k in ks], [gold[k + "_p2"] for k in ks],
                                               np.stack([gold[k + "_T"] for k in ks]), vox, return_counts=True)
        for i, k in enumerate(ks):
            want_iou, want_counts = ko.voxel_overlap(gold[k + "_p1"], gold[k + "_p2"], gold[k + "_T"], vox)
            assert float(iou[i]) == want_iou == float(gold[k + "_iou"]), k
            assert counts[i].cpu().numpy().tolist() == want_counts.tolist(), k


@pytest.mark.gpu
def test_voxel_overlap_gpu_properties():
    from neural_spectral_codec_amd.data import pose_utils as pu
    rng = np.random.default_rng(9)
    p = (rng.uniform(-6, 6, (5000, 3))).astype(np.float32)
    p[:, 2] = -1.7
    eye = np.eye(4)
    # a cloud against itself in float32-exact coordinates -> IoU 1 only if both voxelisations agree;
    # the reference divides cloud 1 in float64 and cloud 2 in float32, so check against the oracle instead
    iou, counts = pu.compute_overlap_batch([p], [p], eye[None], return_counts=True)
    want, wc = ko.voxel_overlap(p, p, eye)
    assert float(iou[0]) == want and counts[0].cpu().numpy().tolist() == wc.tolist()
    # duplicates do not change the sets
    iou2 = pu.compute_overlap_batch([np.concatenate([p[:3000], p[:3000]])], [p[:3000]], eye[None], max_points=6000)
    assert float(iou2[0]) == ko.voxel_overlap(p[:3000], p[:3000], eye)[0]
    # symmetric counts: swapping the clouds (identity transform, voxel 0.25 exactly representable) swaps n1/n2
    a = (np.floor(rng.uniform(-40, 40, (4000, 3))) * 0.25 + 0.125).astype(np.float32)
    b = (np.floor(rng.uniform(-40, 40, (3500, 3))) * 0.25 + 0.125).astype(np.float32)
    _, c_ab = pu.compute_overlap_batch([a], [b], eye[None], voxel_size=0.25, return_counts=True)
    _, c_ba = pu.compute_overlap_batch([b], [a], eye[None], voxel_size=0.25, return_counts=True)
    c_ab, c_ba = c_ab[0].cpu().numpy(), c_ba[0].cpu().numpy()
    assert c_ab[0] == c_ba[1] and c_ab[1] == c_ba[0] and c_ab[2] == c_ba[2] and c_ab[2] > 0
    # device tensors in, down-sampling above max_points (statistical only: the reference's RNG is unseeded)
    big = torch.from_numpy((rng.uniform(-6, 6, (20000, 3)) * [1, 1, 0.02]).astype(np.float32)).cuda()
    v = pu.compute_overlap(big, big, eye, voxel_size=0.5, max_points=5000)
    assert 0.5 < v <= 1.0
    with pytest.raises(Exception):
        pu.compute_overlap(big, big, eye, max_points=7000)            # 14 000 points > hash-set capacity
    with pytest.raises(Exception):
        pu.compute_overlap(p, p, eye, voxel_size=0.0)


@pytest.mark.gpu
def test_empty_inputs():
    """n = 0 everywhere: nothing is launched, shapes are kept."""
    from neural_spectral_codec_amd.data import pose_utils as pu
    from neural_spectral_codec_amd.encoding import quantization as qz
    from neural_spectral_codec_amd.keyframe.graph_manager import build_chain_graph
    q = qz.quantize_batch(torch.empty((0, 800), device="cuda"))
    assert tuple(q.shape) == (0, 800) and q.dtype == torch.uint16
    assert tuple(qz.dequantize_batch(q).shape) == (0, 800)
    rec = qz.pack_records(q, torch.empty((0, 7)), torch.empty((0,), dtype=torch.float64),
                          torch.empty((0,), dtype=torch.int64), torch.empty((0, 20), dtype=torch.uint8))
    assert tuple(rec.shape) == (0, 1720)
    g = build_chain_graph(torch.zeros((0, 800)), 5, "cuda", np.zeros((0, 4, 4)))
    assert tuple(g.edge_index.shape) == (2, 0)
    iou = pu.compute_overlap_batch([], [], np.zeros((0, 4, 4)))
    assert tuple(iou.shape) == (0,)


def _literal_window(descs, ids, max_active, m):
    """What graph_manager.py:75-202 leaves after feeding the keyframes one by one: (active ids, x, edge_index)."""
    act = []
    for i in ids:
        act.append(i)
        if len(act) > max_active:
            act.pop(0)
    x = np.stack([descs[i] for i in act])
    return act, x, ko.chain_graph_loop(len(act), m)[0]


@pytest.mark.parametrize("device", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_temporal_graph_manager(device):
    from types import SimpleNamespace
    from neural_spectral_codec_amd.keyframe.graph_manager import TemporalGraphManager
    rng = np.random.default_rng(1)
    descs = {i: rng.random(800).astype(np.float32) for i in range(40)}
    mgr = TemporalGraphManager(temporal_neighbors=5, max_active_nodes=6, device=device)
    assert mgr.get_graph() is None and mgr.get_statistics()["num_edges"] == 0
    kfs = {}
    for step, i in enumerate(range(40)):
        kfs[i] = SimpleNamespace(keyframe_id=1000 + i, descriptor=descs[i], embedding=None)
        idx = mgr.add_keyframe(kfs[i])
        act, x, ei = _literal_window(descs, list(range(step + 1)), 6, 5)
        g = mgr.get_graph()
        assert idx == min(step, 6)                                     # index at insertion, before the window slides
        assert g.num_nodes == len(act) and (g.x.cpu().numpy() == x).all()
        assert (g.edge_index.cpu().numpy() == ei).all() and getattr(g, "edge_attr", None) is None
        assert [mgr.get_node_index(1000 + a) for a in act] == list(range(len(act)))
        if step == 20:                                                 # embeddings of the active window get cached on freeze
            mgr.update_embeddings(torch.arange(len(act) * 4, dtype=torch.float32).reshape(len(act), 4))
    st = mgr.get_statistics()
    assert st == {"num_active_nodes": 6, "num_frozen_nodes": 34, "total_nodes": 40, "num_edges": 18, "avg_degree": 3.0}
    assert mgr.frozen_embeddings is not None and tuple(mgr.frozen_embeddings.shape) == (6, 4)
    assert mgr.get_all_descriptors().shape == (40, 800) and mgr.get_node_index(1000) is None
    # loop closure between active keyframes: bidirectional, gone again after the next rebuild (reference behaviour)
    assert mgr.add_loop_closure_edge(1034, 1039) and not mgr.add_loop_closure_edge(1000, 1039)
    g = mgr.get_graph()
    assert g.edge_index.shape[1] == 20 and g.edge_index[:, -2:].cpu().tolist() == [[0, 5], [5, 0]]
    assert mgr.get_k_hop_neighbors(0, 1) == {0, 1, 2, 5} and mgr.get_k_hop_neighbors(3, 0) == {3}
    sub, mapping = mgr.get_local_subgraph(0, 1)
    assert mapping == {0: 0, 1: 1, 2: 2, 5: 3} and sub.num_nodes == 4
    assert (sub.x.cpu().numpy() == g.x.cpu().numpy()[[0, 1, 2, 5]]).all()
    assert sorted(map(tuple, sub.edge_index.t().cpu().tolist())) == sorted(
        [(0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1), (0, 3), (3, 0)])
    mgr.add_keyframe(SimpleNamespace(keyframe_id=5000, descriptor=descs[0], embedding=None))
    assert mgr.get_graph().edge_index.shape[1] == 18
    with pytest.raises(ValueError):
        mgr.add_keyframe(SimpleNamespace(keyframe_id=1, descriptor=None, embedding=None))
    with pytest.raises(ValueError):
        mgr.update_embeddings(torch.zeros((3, 4)))
    if device == "cuda":                                               # the online loop of pipeline.py:250-256
        from neural_spectral_codec_amd.gnn.model import create_spectral_gnn
        gnn = create_spectral_gnn(edge_dim=None).cuda().eval()
        with torch.no_grad():
            emb = gnn(mgr.get_graph())
        mgr.update_embeddings(emb)
        assert mgr.get_all_keyframes()[-1].embedding.shape == (800,)
    mgr.reset()
    assert mgr.get_graph() is None and mgr.get_statistics()["total_nodes"] == 0
