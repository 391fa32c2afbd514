"""Parity of the HIP GNN forward (nsc_gat_forward through the C ABI) against the CPU restatement.
Bar (north_star): within 1e-4 relative for the GAT output."""
import numpy as np
import pytest
import torch

import gat_oracle as go
from neural_spectral_codec_amd.gnn.model import create_spectral_gnn, SpectralGNN
from neural_spectral_codec_amd.keyframe import graph_manager as gm

pytestmark = pytest.mark.gpu


def _check(out, model, g):
    """north_star: within 1e-4 relative for the GAT output -- ELEMENT-WISE, |gpu - ref| <= 1e-4 |ref| + 1e-6, against
    the restatement in float32 (the reference's own arithmetic, different summation order) AND in float64."""
    go.assert_within_bar(out, go.forward_reference(model, g), what="vs float32 restatement")
    return go.assert_within_bar(out, go.forward_reference(model, g, dtype=torch.float64), what="vs float64 restatement")


def _model(edge_dim=2, seed=0, **kw):
    torch.manual_seed(seed)
    m = create_spectral_gnn(edge_dim=edge_dim, **kw)
    go.randomize_bn_stats(m, seed + 1)
    with torch.no_grad():
        for c in m.gnn.convs:
            c.bias.normal_(0, 0.1)
    return m.to("cuda").eval()


@pytest.mark.parametrize("n", [1, 2, 3, 33, 64, 1000])
def test_chain_graph_forward(n):
    m = _model()
    g = gm.synthetic_chain_graph(n, device="cuda", seed=n)
    with torch.no_grad():
        out = m(g)
    assert out.shape == (n, 800)
    _check(out, m, g)


def test_kitti00_shape():
    """config 3: 4 541 keyframes, 5 temporal neighbours -> 18 158 edges."""
    m = _model()
    g = gm.synthetic_chain_graph(4541, device="cuda", seed=11)
    assert g.edge_index.shape == (2, 18158)
    with torch.no_grad():
        out = m(g)
    _check(out, m, g)


def test_no_edge_attr_paths():
    """online path: graph without edge_attr and/or model without edge_dim (pipeline.py:158-166)."""
    g = gm.synthetic_chain_graph(50, device="cuda", seed=2)
    m_plain = _model(edge_dim=None)
    with torch.no_grad():
        out = m_plain(g)                                   # edge_attr present but model has no edge_dim
    _check(out, m_plain, g)
    g2 = gm.Data(x=g.x, edge_index=g.edge_index, num_nodes=50)
    m_edge = _model(edge_dim=2)
    with torch.no_grad():
        out2 = m_edge(g2)                                  # model has edge_dim but data has none
    _check(out2, m_edge, g2)


def test_irregular_graph_self_loops_and_hubs():
    """random multigraph-free graph with explicit self loops, isolated nodes and a hub of degree 200."""
    rng = np.random.default_rng(0)
    n = 300
    pairs = set()
    while len(pairs) < 900:
        a, b = rng.integers(0, n - 10, 2)
        if a != b:
            pairs.add((int(a), int(b)))
    pairs |= {(int(s), 7) for s in rng.choice(n - 10, 200, replace=False) if s != 7}   # hub target 7
    edges = list(pairs) + [(5, 5), (9, 9)]                 # self loops get removed and re-added
    rng.shuffle(edges)
    ei = torch.tensor(edges, dtype=torch.long).t().contiguous()
    ea = torch.rand(ei.shape[1], 2)
    x = torch.rand(n, 800)
    g = gm.Data(x=x.cuda(), edge_index=ei.cuda(), edge_attr=ea.cuda(), num_nodes=n)
    m = _model()
    with torch.no_grad():
        out = m(g)
    _check(out, m, g)


def test_forward_with_attention():
    m = _model(edge_dim=None)
    g = gm.synthetic_chain_graph(40, device="cuda", seed=5)
    with torch.no_grad():
        out, att = m.gnn.forward_with_attention(g)
        plain = m(gm.Data(x=g.x, edge_index=g.edge_index, num_nodes=40))
    assert torch.allclose(out, plain, rtol=1e-6, atol=1e-6)
    assert len(att) == 3
    ei, alpha = att[0]
    assert ei.shape[1] == g.edge_index.shape[1] + 40 and alpha.shape == (ei.shape[1], 1)
    s = torch.zeros(40, device="cuda").index_add_(0, ei[1], alpha[:, 0])
    assert torch.allclose(s, torch.ones(40, device="cuda"), atol=1e-5)
    # first layer's attention against the oracle, in PyG order
    gnn = m.gnn
    with torch.no_grad():
        h = torch.relu(gnn.input_norm(gnn.input_proj(g.x))).cpu()
    c = gnn.convs[0]
    _, ei_ref, a_ref = go.gatconv_reference(h, g.edge_index.cpu(), None, c.lin_src.weight.cpu(),
                                            c.att_src.cpu(), c.att_dst.cpu(), None, None, c.bias.cpu(),
                                            return_alpha=True)
    assert torch.equal(ei.cpu(), ei_ref)
    assert torch.allclose(alpha[:, 0].cpu(), a_ref, rtol=1e-4, atol=1e-6)


def test_other_dims_and_residual_proj():
    torch.manual_seed(3)
    m = SpectralGNN(input_dim=64, hidden_dim=128, output_dim=32, n_layers=2, edge_dim=2)
    go.randomize_bn_stats(m)
    m = m.to("cuda").eval()
    g = gm.synthetic_chain_graph(70, device="cuda", seed=6, features=torch.rand(70, 64))
    with torch.no_grad():
        out = m(g)
    assert out.shape == (70, 32)
    _check(out, m, g)


@pytest.mark.parametrize("n,edge_dim", [(1, 2), (33, 2), (1024, 2), (4541, 2), (200, None)])
def test_coresident_variant_is_bit_identical(n, edge_dim):
    """NSC_GAT_CORESIDENT (LDS-free GEMMs, 4-row aggregate) keeps the k order and operand assignment of the
    default kernels: same bits, and still within the oracle bar."""
    m = _model(edge_dim=edge_dim)
    g = gm.synthetic_chain_graph(n, device="cuda", seed=n + 3)
    with torch.no_grad():
        a = m(g)
        m.gnn.coresident = True
        b = m(g)
        m.gnn.coresident = "shared_b"                 # NSC_GAT_SHARED_B: 64 x 64 tiles, weight block shared through 10 KB of LDS
        c = m(g)
        m.gnn.coresident = "lds_tiled"                # NSC_GAT_LDS_TILED: the round-2 register-staged GEMMs
        d = m(g)
        m.gnn.coresident = False
    assert torch.equal(a, b) and torch.equal(a, c) and torch.equal(a, d)
    _check(b, m, g)


@pytest.mark.parametrize("n", [15, 16, 17, 255, 256, 257, 1279, 1280, 1281, 2047, 2049, 4095, 4097, 6000, 9000])
def test_glds_gemm_tile_boundaries(n):
    """The default GEMM (gemm_glds_kernel) picks its tile height 16 ACC from the node count (one round of workgroups where
    that is possible): sizes either side of the switches between ACC values, ragged last tiles, several rounds -- every
    kernel set bit-identical, within the oracle bar."""
    m = _model()
    g = gm.synthetic_chain_graph(n, device="cuda", seed=n)
    with torch.no_grad():
        a = m(g)
        m.gnn.coresident = True
        b = m(g)
        m.gnn.coresident = "lds_tiled"
        c = m(g)
        m.gnn.coresident = False
    assert torch.equal(a, b) and torch.equal(a, c)
    _check(a, m, g)


@pytest.mark.parametrize("n", [2496, 2497, 2561, 3071])
def test_large_tile_variant_boundaries(n):
    """output_proj switches to 64 x 64 workgroup tiles (32 x 32 wave tiles) once ceil(n / 64) * 13 >= 512, i.e. from
    n = 2 497 on: sizes either side of the switch and with ragged last tiles -- bit-identical to the LDS-free set (whose
    tiling never changes) and within the oracle bar."""
    m = _model()
    g = gm.synthetic_chain_graph(n, device="cuda", seed=n)
    with torch.no_grad():
        a = m(g)
        m.gnn.coresident = True
        b = m(g)
        m.gnn.coresident = False
    assert torch.equal(a, b)
    _check(a, m, g)


def test_coresident_other_dims():
    m = _model(edge_dim=2, hidden_dim=64, input_dim=48, output_dim=80)
    g = gm.synthetic_chain_graph(77, device="cuda", seed=4)
    g.x = torch.rand((77, 48), device="cuda")
    with torch.no_grad():
        a = m(g)
        m.gnn.coresident = True
        b = m(g)
        m.gnn.coresident = "shared_b"
        c = m(g)
    assert torch.equal(a, b) and torch.equal(a, c)


@pytest.mark.parametrize("dims", [(16, 16, 1), (48, 32, 5), (64, 128, 67), (800, 256, 800), (32, 1024, 33)])
def test_odd_dimensions_all_kernel_sets(dims):
    """Feature sizes off the tiles' grain (K of 16 / 48 / 64, output widths 1, 5, 33, 67 -- scalar epilogue, clamped column
    loads, the residual_proj second GEMM; hidden 1024 = 16 k-chunks and 4 float4 chunks per aggregate lane): the four kernel
    sets agree bit for bit and sit within the oracle bar."""
    i, h, o = dims
    m = _model(edge_dim=2, hidden_dim=h, input_dim=i, output_dim=o)
    n = 131
    g = gm.synthetic_chain_graph(n, device="cuda", seed=i + o)
    g.x = torch.rand((n, i), device="cuda")
    outs = []
    with torch.no_grad():
        for ks in (False, True, "shared_b", "lds_tiled"):
            m.gnn.coresident = ks
            outs.append(m(g))
        m.gnn.coresident = False
    for t in outs[1:]:
        assert torch.equal(outs[0], t)
    _check(outs[0], m, g)


def test_pipelined_path_matches_serial():
    """ShardedDescriptorPath(pipeline=True): encoder of batch k+1 on one stream over the GNN of batch k on a
    second one, double-buffered descriptors -- every step's results equal the one-stream path's."""
    from neural_spectral_codec_amd import distributed as nd, synth
    from neural_spectral_codec_amd.encoding import SpectralEncoder
    n = 96
    enc = SpectralEncoder(n_elevation=16).to("cuda")
    m = _model()
    poses = synth.make_pose_chain(n, 0)
    batches = [synth.make_clouds_device(n, 3000, "cuda", seed=s) for s in (1, 2, 3, 4, 5)]
    serial = nd.ShardedDescriptorPath(enc, m, n, poses)
    piped = nd.ShardedDescriptorPath(enc, m, n, poses, pipeline=True)
    with torch.no_grad():
        want = [tuple(t.clone() for t in serial.step(b)) for b in batches]
        m.gnn.coresident = False
        got = []
        for b in batches:
            d, e = piped.step(b)
            torch.cuda.current_stream().wait_event(piped.last_event)
            got.append((d.clone(), e.clone()))               # the descriptor buffer is reused two steps later
        piped.synchronize()
        torch.cuda.synchronize()
    for (wd, we), (gd, ge) in zip(want, got):
        assert torch.equal(wd, gd) and torch.equal(we, ge)
    # fire-and-forget issue order (what bench.py does): only the last results are read
    with torch.no_grad():
        for b in batches:
            d, e = piped.step(b)
        piped.synchronize()
        torch.cuda.synchronize()
    assert torch.equal(d, want[-1][0]) and torch.equal(e, want[-1][1])
    m.gnn.coresident = False


@pytest.mark.parametrize("opts", [dict(gnn_graph=True), dict(gnn_streams=2), dict(gnn_graph=True, gnn_streams=2, encoder_streams=1)])
def test_pipelined_path_options_match_serial(opts):
    """The pipelined step's options -- the GNN forward replayed as a captured hipGraph per rotating buffer, the GNN
    passes of consecutive batches on two streams, one or two encoder streams -- change scheduling only: every step's
    results equal the one-stream path's, bit for bit, over more steps than rotating buffers (a captured forward is keyed
    on its input buffer and the model's parameter versions: a weight update in between must recapture)."""
    from neural_spectral_codec_amd import distributed as nd, synth
    from neural_spectral_codec_amd.encoding import SpectralEncoder
    n = 96
    enc = SpectralEncoder(n_elevation=16).to("cuda")
    m = _model()
    poses = synth.make_pose_chain(n, 0)
    batches = [synth.make_clouds_device(n, 3000, "cuda", seed=s) for s in range(1, 11)]
    serial = nd.ShardedDescriptorPath(enc, m, n, poses)
    piped = nd.ShardedDescriptorPath(enc, m, n, poses, pipeline=True, **opts)
    with torch.no_grad():
        m.gnn.coresident = False
        want = [tuple(t.clone() for t in serial.step(b)) for b in batches]
        got = []
        for b in batches:
            d, e = piped.step(b)
            torch.cuda.current_stream().wait_event(piped.last_event)
            got.append((d.clone(), e.clone()))
        piped.synchronize()
        torch.cuda.synchronize()
        for (wd, we), (gd, ge) in zip(want, got):
            assert torch.equal(wd, gd) and torch.equal(we, ge)
        if opts.get("gnn_graph"):
            assert len(piped._gnn_graphs) == nd.ShardedDescriptorPath._PIPE_BUFFERS      # one capture per rotating buffer
            # new weights: the captures are stale and must not be replayed
            m.gnn.output_proj.bias.add_(0.25)
            m.gnn.coresident = False
            want2 = [tuple(t.clone() for t in serial.step(b)) for b in batches[:6]]
            got2 = []
            for b in batches[:6]:
                d, e = piped.step(b)
                torch.cuda.current_stream().wait_event(piped.last_event)
                got2.append((d.clone(), e.clone()))
            piped.synchronize()
            torch.cuda.synchronize()
            for (wd, we), (gd, ge) in zip(want2, got2):
                assert torch.equal(wd, gd) and torch.equal(we, ge)
            assert not torch.equal(want2[0][1], want[0][1])
    m.gnn.coresident = False


# ---- round 4: one-launch GATConv layers on banded graphs (gat_layer_banded_kernel) ---------------------------------------
def _banded_vs_generic(m, g, want_band=2):
    """The default forward takes the banded kernel when the graph qualifies; "generic" (NSC_GAT_GENERIC) forces lin GEMM +
    gat_aggregate_kernel.  Same bits."""
    use_edge = getattr(g, "edge_attr", None) is not None and m.gnn.edge_dim is not None
    with torch.no_grad():
        m.gnn.coresident = False
        a = m(g)
        assert m.gnn._csr(g, use_edge).band == want_band
        m.gnn.coresident = "generic"
        b = m(g)
        m.gnn.coresident = False
    assert torch.equal(a, b)
    return a


# sizes either side of the owned-row counts of the tiles (12, 28, 44, 60, 76, 92 rows) and of the tile switches
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 11, 12, 13, 28, 29, 57, 75, 76, 77, 153, 1023, 1024, 3000, 4541, 7000])
@pytest.mark.parametrize("edge_dim", [2, None])
def test_banded_layer_kernel_is_bit_identical(n, edge_dim):
    m = _model(edge_dim=edge_dim)
    g = gm.synthetic_chain_graph(n, device="cuda", seed=n + 17)
    out = _banded_vs_generic(m, g)
    if n <= 1024:
        _check(out, m, g)


@pytest.mark.parametrize("tn", [1, 3, 5])
def test_banded_other_bandwidths_and_hidden(tn):
    """temporal_neighbors 1 (self loops only), 3 (one neighbour each side), 5; hidden 64 / 128 (1 / 2 column blocks, 1 / 2
    k-chunks), no BatchNorm-free path exists, residual on the middle layer of 3."""
    for hidden in (64, 128):
        m = _model(edge_dim=2, hidden_dim=hidden, input_dim=48, output_dim=80)
        g = gm.synthetic_chain_graph(97, device="cuda", seed=tn, temporal_neighbors=tn)
        g.x = torch.rand((97, 48), device="cuda")
        out = _banded_vs_generic(m, g)
        _check(out, m, g)


def test_banded_shuffled_and_duplicate_edges():
    """A banded multigraph: the chain's edges in random order, some twice, explicit self loops (removed and re-added) --
    at most 8 entries per target, so it still takes the banded kernel; entry order = edge order decides the summation."""
    rng = np.random.default_rng(3)
    n = 203
    base = gm.chain_edges(n, 5)
    extra = base[rng.choice(len(base), 150, replace=False)]
    loops = np.stack([np.arange(0, n, 7), np.arange(0, n, 7)], 1)
    edges = np.concatenate([base, extra, loops], 0)
    rng.shuffle(edges)
    # no target above 7 real edges (+ the self loop = 8 slots)
    keep, cnt = [], np.zeros(n, int)
    for s, t in edges:
        if s == t or cnt[t] < 7:
            keep.append((s, t))
            cnt[t] += s != t
    ei = torch.tensor(np.array(keep), dtype=torch.long).t().contiguous().cuda()
    ea = torch.rand(ei.shape[1], 2).cuda()
    g = gm.Data(x=torch.rand(n, 800).cuda(), edge_index=ei, edge_attr=ea, num_nodes=n)
    m = _model()
    out = _banded_vs_generic(m, g)
    _check(out, m, g)


def test_not_banded_graphs_keep_the_generic_kernels():
    """A loop closure (|i - j| > 2), a target with more than 8 entries, edge_dim 3: band stays 0, results as before."""
    m = _model()
    g = gm.synthetic_chain_graph(120, device="cuda", seed=4)
    ei = torch.cat([g.edge_index, torch.tensor([[3, 90], [90, 3]], device="cuda")], 1)
    ea = torch.cat([g.edge_attr, torch.rand(2, 2, device="cuda")], 0)
    g2 = gm.Data(x=g.x, edge_index=ei, edge_attr=ea, num_nodes=120)
    _check(_banded_vs_generic(m, g2, want_band=0), m, g2)
    dup = torch.tensor([[11] * 6, [12] * 6], device="cuda")
    g3 = gm.Data(x=g.x, edge_index=torch.cat([g.edge_index, dup], 1),
                 edge_attr=torch.cat([g.edge_attr, torch.rand(6, 2, device="cuda")], 0), num_nodes=120)
    _check(_banded_vs_generic(m, g3, want_band=0), m, g3)
    m3 = _model(edge_dim=3)
    g4 = gm.Data(x=g.x, edge_index=g.edge_index, edge_attr=torch.rand(g.edge_index.shape[1], 3, device="cuda"), num_nodes=120)
    _check(_banded_vs_generic(m3, g4, want_band=0), m3, g4)


def test_banded_attention_coefficients():
    """forward_with_attention through the banded kernel returns the generic kernels' coefficients, bit for bit."""
    m = _model(edge_dim=None)
    g = gm.synthetic_chain_graph(150, device="cuda", seed=8)
    with torch.no_grad():
        out_a, att_a = m.gnn.forward_with_attention(g)
        m.gnn.coresident = "generic"
        out_b, att_b = m.gnn.forward_with_attention(g)
        m.gnn.coresident = False
    assert torch.equal(out_a, out_b)
    for (ea, aa), (eb, ab) in zip(att_a, att_b):
        assert torch.equal(ea, eb) and torch.equal(aa, ab)


def test_scratch_cache_is_bounded_and_never_replaces_a_buffer():
    """_lib.ScratchCache (the Python layer's CSR / forward / encoder scratch): a larger request ADDS a buffer (a captured
    hipGraph may replay the smaller one's address), keys of finished threads go first once the cap is reached, keys touched
    during a stream capture are pinned until release()."""
    import threading
    from neural_spectral_codec_amd import _lib
    c = _lib.ScratchCache(cap=4)
    a = c.get("cuda:0", 1000, "x")
    b = c.get("cuda:0", 5000, "x")
    assert a.data_ptr() != b.data_ptr() and b.numel() >= 5000
    assert c.get("cuda:0", 800, "x") is a and c.get("cuda:0", 4000, "x") is b and len(c) == 1

    def worker():
        c.get("cuda:0", 256, "y")
    for _ in range(10):
        t = threading.Thread(target=worker)
        t.start()
        t.join()
    assert len(c) <= 4 and c.get("cuda:0", 800, "x") is a          # the live thread's key survived the dead ones
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            pinned = c.get("cuda:0", 512, "captured")
    for i in range(8):
        c.get("cuda:0", 256, f"z{i}")
    with torch.cuda.stream(s):
        assert c.get("cuda:0", 512, "captured") is pinned
    c.release()
    assert len(c) == 0


def test_graph_whose_feature_matrix_passes_2_31_elements():
    """Maximum sizes: a 2.75 M-keyframe temporal chain -- x and the output hold 2.2e9 floats each (8.8 GB), so the element offset
    of the last 65 000 rows lies beyond 2^31.  Windows at the start, across row 2^31 / 800 and at the end are checked against
    the restatement evaluated on the window + a 6-node halo (3 layers x 2 hops: the interior rows are exact), and the generic
    kernel set must give the same bits as the one-launch banded layers everywhere."""
    n = 2_750_000
    m = _model()
    feats = torch.rand((n, 800), device="cuda")
    feats.pow_(4)
    feats.div_(feats.sum(1, keepdim=True))
    from neural_spectral_codec_amd import synth
    g = gm.build_chain_graph(feats, 5, "cuda", synth.make_pose_chain(n, 0))
    assert g.x.numel() > 2 ** 31 and g.x.data_ptr() == feats.data_ptr()
    with torch.no_grad():
        out = m(g)
        assert m.gnn._csr(g, True).band == 2
        m.gnn.coresident = "generic"
        gen = m(g)
        m.gnn.coresident = False
    torch.cuda.synchronize()
    assert out.shape == (n, 800) and torch.equal(out, gen)
    del gen
    cut = 2 ** 31 // 800
    for a, b in ((0, 150), (cut - 60, cut + 60), (n - 150, n)):
        wlo, whi = max(0, a - 6), min(n, b + 6)
        ei = g.edge_index
        keep = (ei[0] >= wlo) & (ei[0] < whi) & (ei[1] >= wlo) & (ei[1] < whi)
        sub = gm.Data(x=g.x[wlo:whi].cpu(), edge_index=(ei[:, keep] - wlo).cpu(), edge_attr=g.edge_attr[keep].cpu(),
                      num_nodes=whi - wlo)
        ref = go.forward_reference(m, sub, dtype=torch.float64)
        go.assert_within_bar(out[a:b], ref[a - wlo:b - wlo], what=f"rows {a}..{b} vs float64 restatement")
