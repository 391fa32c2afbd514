"""SpectralGNN on MI355X -- drop-in for the reference's src/gnn/model.py.

Same classes, constructor arguments, state-dict keys and ``forward(data)`` contract
(model.py:21-205 SpectralGNN, :208-281 LocalUpdateGNN, :284-324 create_spectral_gnn).  The forward
pass is one call into the C ABI (nsc_gat_forward, csrc/nsc_gat.hip): f32-MFMA projection GEMMs with
fused BatchNorm/ReLU/residual epilogues and a wavefront-per-node attention/aggregation kernel.
The module must live on a HIP device; there is no CPU fallback.
"""
import ctypes as C
import os
from typing import Optional

import torch
import torch.nn as nn

from .. import _lib
from .gat_conv import GATConv

def _ws(device, nbytes, tag):
    # one scratch buffer per (device, stream, host thread, use), bounded and capture-safe: _lib.ScratchCache (round 3: two
    # rank threads building their CSR in one shared scratch faulted the aggregate kernel; round 4: the dict only grew)
    return _lib.scratch.get(device, nbytes, tag)


class GraphCSR:
    """Device CSR-by-target arrays of one graph (built by nsc_graph_build_csr, cached per graph)."""

    def __init__(self, edge_index: torch.Tensor, edge_attr: Optional[torch.Tensor], n_nodes: int):
        dev = edge_index.device
        _lib.require_cuda(edge_index, "data.edge_index")
        L = _lib.lib()
        ei = edge_index.to(torch.int64).contiguous()
        E = int(ei.shape[1])
        self.n_nodes, self.n_edges = n_nodes, E
        self.capacity = E + n_nodes
        self.row_ptr = torch.empty(n_nodes + 1, dtype=torch.int32, device=dev)
        self.src = torch.empty(self.capacity, dtype=torch.int32, device=dev)
        self.eid = torch.empty(self.capacity, dtype=torch.int32, device=dev)
        self.edge_dim = 0
        self.loop_attr = None
        ea = None
        if edge_attr is not None:
            ea = edge_attr.to(device=dev, dtype=torch.float32).contiguous()
            if ea.dim() == 1:
                ea = ea.unsqueeze(1)
            self.edge_dim = int(ea.shape[1])
            self.loop_attr = torch.empty((n_nodes, self.edge_dim), dtype=torch.float32, device=dev)
        self.edge_attr = ea
        nbytes = L.nsc_graph_workspace_bytes(n_nodes, E)
        ws = _ws(dev, nbytes, "graph")
        with torch.cuda.device(dev):
            st = L.nsc_graph_build_csr(_lib.ptr(ei), E, n_nodes, _lib.ptr(ea), self.edge_dim,
                                       _lib.ptr(self.row_ptr), _lib.ptr(self.src), _lib.ptr(self.eid),
                                       _lib.ptr(self.loop_attr), _lib.ptr(ws), nbytes,
                                       _lib.stream_ptr(dev))
        _lib.check(st, "nsc_graph_build_csr")

        self.t_ptr = self.t_entry = self.tgt = None
        # Banded form (nsc_graph_band_entries): the temporal chain of every reference caller has its sources within 2 rows
        # of the target (graph_manager.py:520-532), and such a graph runs a GATConv layer as ONE launch.  Whether THIS graph
        # qualifies is read back once per graph (two int32; the CSR is cached per graph by SpectralGNN._csr).
        self.band, self.band_entries = 0, None
        if n_nodes > 0 and self.edge_dim in (0, 2):
            ent = torch.empty((n_nodes, 8, 4), dtype=torch.float32, device=dev)
            info = torch.empty(2, dtype=torch.int32, device=dev)
            g = self.struct()
            with torch.cuda.device(dev):
                st = L.nsc_graph_band_entries(C.byref(g), _lib.ptr(ea), self.edge_dim, _lib.ptr(ent), _lib.ptr(info),
                                              _lib.stream_ptr(dev))
            _lib.check(st, "nsc_graph_band_entries")
            max_off, max_deg = info.tolist()
            if max_off <= 2 and max_deg <= 8:
                self.band, self.band_entries = 2, ent

    def ensure_transpose(self):
        """Entries grouped by source (nsc_graph_transpose); the backward needs it, built once."""
        if self.t_ptr is not None:
            return
        dev = self.row_ptr.device
        L = _lib.lib()
        self.t_ptr = torch.empty(self.n_nodes + 1, dtype=torch.int32, device=dev)
        self.t_entry = torch.empty(self.capacity, dtype=torch.int32, device=dev)
        self.tgt = torch.empty(self.capacity, dtype=torch.int32, device=dev)
        nbytes = L.nsc_graph_transpose_workspace_bytes(self.n_nodes)
        ws = _ws(dev, nbytes, "graph")
        g = self.struct()
        with torch.cuda.device(dev):
            st = L.nsc_graph_transpose(C.byref(g), _lib.ptr(self.t_ptr), _lib.ptr(self.t_entry),
                                       _lib.ptr(self.tgt), _lib.ptr(ws), nbytes, _lib.stream_ptr(dev))
        _lib.check(st, "nsc_graph_transpose")

    def struct(self) -> _lib.Graph:
        g = _lib.Graph()
        g.n_nodes = self.n_nodes
        g.nnz = self.capacity
        g.row_ptr = self.row_ptr.data_ptr()
        g.src = self.src.data_ptr()
        g.eid = self.eid.data_ptr()
        g.loop_attr = self.loop_attr.data_ptr() if self.loop_attr is not None else None
        if self.t_ptr is not None:
            g.t_ptr, g.t_entry, g.tgt = self.t_ptr.data_ptr(), self.t_entry.data_ptr(), self.tgt.data_ptr()
        if getattr(self, "band", 0):
            g.band_entries, g.band = self.band_entries.data_ptr(), self.band
        return g


def _train_forward_raw(gnn, x, csr, dropout_p, seed):
    """nsc_gat_forward_train: returns (out, state); state holds the workspace with the saved activations for
    _train_backward_raw.  Used by the autograd Function below and, without autograd, by GNNTrainer's direct step."""
    L = _lib.lib()
    dev = x.device
    m = gnn._train_struct()
    csr.ensure_transpose()
    g = csr.struct()
    cfg = _lib.GatTrainCfg()
    cfg.dropout_p, cfg.bn_momentum = float(dropout_p), float(gnn.input_norm.momentum or 0.1)
    cfg.seed, cfg.update_running_stats = int(seed), 1
    sd = getattr(gnn, "_seed_dev", None)          # a device word the kernels read the seed from (captured steps)
    cfg.seed_dev = sd.data_ptr() if sd is not None else None
    n = int(x.shape[0])
    out = torch.empty((n, gnn.output_dim), dtype=torch.float32, device=dev)
    nbytes = L.nsc_gat_train_workspace_bytes(C.byref(m), C.byref(g))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)      # holds the saved activations
    if os.environ.get("NSC_DEBUG_WS_NAN") == "1":              # debugging: a read of never-written workspace shows up as NaN
        ws.fill_(0xFF)
    with torch.cuda.device(dev):
        st = L.nsc_gat_forward_train(C.byref(m), C.byref(g), _lib.ptr(x), _lib.ptr(csr.edge_attr),
                                     C.byref(cfg), _lib.ptr(out), _lib.ptr(ws), nbytes, _lib.stream_ptr(dev))
    _lib.check(st, "nsc_gat_forward_train")
    _bump_batches_tracked(gnn)
    return out, (csr, cfg, ws, nbytes, x, sd)


def _bump_batches_tracked(gnn):
    """``num_batches_tracked += 1`` of every BatchNorm (what torch's own train-mode forward does) as ONE launch: the counters
    are 0-d views of one int64 vector (set up outside a capture, re-made whenever a ``.to()`` / a new buffer broke the aliasing;
    ``load_state_dict`` copies in place and keeps it).  Four 4 us kernels per batch of the captured training step were these."""
    bns = [bn for bn in [gnn.input_norm] + list(gnn.batch_norms) if bn.num_batches_tracked is not None]
    if not bns:
        return
    bufs = [bn.num_batches_tracked for bn in bns]
    flat = gnn.__dict__.get("_nbt_flat")
    ok = (flat is not None and flat.numel() == len(bufs) and flat.device == bufs[0].device
          and all(b_.dtype == flat.dtype and b_.dim() == 0 and b_.device == flat.device
                  and b_.data_ptr() == flat.data_ptr() + i * flat.element_size() for i, b_ in enumerate(bufs)))
    if not ok:
        same = all(b_.dim() == 0 and b_.dtype == bufs[0].dtype and b_.device == bufs[0].device for b_ in bufs)
        if not same or (bufs[0].is_cuda and torch.cuda.is_current_stream_capturing()):
            for b_ in bufs:                                  # unusual buffers, or no allocation wanted inside a capture
                b_ += 1
            return
        flat = torch.stack([b_.detach() for b_ in bufs])
        for i, bn in enumerate(bns):
            bn._buffers["num_batches_tracked"] = flat[i]
        gnn.__dict__["_nbt_flat"] = flat
    flat += 1


def _train_backward_raw(gnn, state, grad_out, grads, accumulate, need_x):
    """nsc_gat_backward into ``grads`` (tensors in _train_params() order; accumulate: added to what they hold).  Returns the
    gradient w.r.t. x or None."""
    csr, cfg, ws, nbytes, x, _sd = state
    L = _lib.lib()
    dev = x.device
    m = gnn._train_struct()
    g = csr.struct()
    cfg.accumulate_grads = 1 if accumulate else 0
    gs = _lib.GatGrads()
    it = iter(grads)
    gs.in_w, gs.in_b = next(it).data_ptr(), next(it).data_ptr()
    gs.in_bn_w, gs.in_bn_b = next(it).data_ptr(), next(it).data_ptr()
    gs.out_w, gs.out_b = next(it).data_ptr(), next(it).data_ptr()
    if gnn.residual_proj is not None:
        gs.res_w, gs.res_b = next(it).data_ptr(), next(it).data_ptr()
    for l, conv in enumerate(gnn.convs):
        gl = gs.layers[l]
        gl.lin_w, gl.att_src, gl.att_dst = next(it).data_ptr(), next(it).data_ptr(), next(it).data_ptr()
        if conv.lin_edge is not None:
            gl.lin_edge_w, gl.att_edge = next(it).data_ptr(), next(it).data_ptr()
        gl.bias, gl.bn_w, gl.bn_b = next(it).data_ptr(), next(it).data_ptr(), next(it).data_ptr()
    gx = torch.empty_like(x) if need_x else None
    gs.x = gx.data_ptr() if gx is not None else None
    with torch.cuda.device(dev):
        st = L.nsc_gat_backward(C.byref(m), C.byref(g), _lib.ptr(x), _lib.ptr(csr.edge_attr),
                                C.byref(cfg), _lib.ptr(grad_out), C.byref(gs), _lib.ptr(ws), nbytes,
                                _lib.stream_ptr(dev))
    _lib.check(st, "nsc_gat_backward")
    return gx


class _GatTrainFunction(torch.autograd.Function):
    """model.train(); model(data) with autograd: nsc_gat_forward_train / nsc_gat_backward."""

    @staticmethod
    def forward(ctx, gnn, x, csr, dropout_p, seed, *params):
        out, state = _train_forward_raw(gnn, x, csr, dropout_p, seed)
        ctx.gnn, ctx.state = gnn, state
        ctx.need_x = x.requires_grad
        return out

    @staticmethod
    def backward(ctx, grad_out):
        gnn = ctx.gnn
        params = gnn._train_params()
        # GNNTrainer's steps: the kernels ADD into the existing .grad tensors (NscGatTrainCfg.accumulate_grads) and autograd
        # gets no parameter gradients back -- no AccumulateGrad axpy per parameter (30 small kernels per batch)
        direct = bool(getattr(gnn, "_direct_grads", False)) and all(p.grad is not None and p.grad.is_contiguous() for p in params)
        grads = [p.grad for p in params] if direct else [torch.empty_like(p) for p in params]
        go = grad_out.contiguous().float()
        gx = _train_backward_raw(gnn, ctx.state, go, grads, direct, ctx.need_x)
        ctx.state = None
        if direct:
            return (None, gx, None, None, None) + (None,) * len(grads)
        return (None, gx, None, None, None, *grads)


class SpectralGNN(nn.Module):
    """Input(800) -> Proj(256) -> GAT x3 (256) -> Proj(800) -> Output(800)   (model.py:21-94)"""

    def __init__(self, input_dim: int = 800, hidden_dim: int = 256, output_dim: int = 800,
                 n_layers: int = 3, n_heads: int = 1, dropout: float = 0.1, residual: bool = True,
                 edge_dim: int = None):
        super().__init__()
        self.input_dim, self.hidden_dim, self.output_dim = input_dim, hidden_dim, output_dim
        self.n_layers, self.n_heads, self.dropout = n_layers, n_heads, dropout
        self.residual, self.edge_dim = residual, edge_dim
        self.input_proj = nn.Linear(input_dim, hidden_dim)                   # :67
        self.input_norm = nn.BatchNorm1d(hidden_dim)                         # :68
        self.convs = nn.ModuleList()
        self.batch_norms = nn.ModuleList()
        for _ in range(n_layers):                                            # :74-85
            self.convs.append(GATConv(hidden_dim, hidden_dim, heads=n_heads, concat=False,
                                      dropout=dropout, edge_dim=edge_dim))
            self.batch_norms.append(nn.BatchNorm1d(hidden_dim))
        self.output_proj = nn.Linear(hidden_dim, output_dim)                 # :88
        if residual and input_dim != output_dim:                             # :91-94
            self.residual_proj = nn.Linear(input_dim, output_dim)
        else:
            self.residual_proj = None
        self._csr_cache = {}
        self._struct_cache = None          # (key, GatModel, folded tensor)
        # True: launch the LDS-free, low-VGPR kernel set (NSC_GAT_CORESIDENT) whose workgroups fit beside a
        # resident encoder grid -- used by distributed.ShardedDescriptorPath(pipeline=True).  "shared_b": that set with
        # the small-LDS GEMMs (NSC_GAT_SHARED_B).  "lds_tiled": the stand-alone forward with the round-2 GEMMs
        # (NSC_GAT_LDS_TILED) instead of the LDS-DMA GEMM.  "generic": the stand-alone forward without the one-launch
        # layers of a banded graph (NSC_GAT_GENERIC).  Same output, bit for bit.
        self.coresident = False
        self._seed_dev = None              # device int64[1]: dropout seed read by the kernels at run time (captured steps)
        self._direct_grads = False         # backward adds straight into the parameters' .grad tensors (GNNTrainer's steps)

    # -- plumbing ---------------------------------------------------------------------------
    def __getstate__(self):
        # device-pointer caches are rebuilt on demand; keep them out of deepcopy / pickle / torch.save
        state = self.__dict__.copy()
        state["_csr_cache"] = {}
        state["_struct_cache"] = None
        state["_train_struct_cache"] = None
        state["_seed_dev"] = None
        state["_fold_generation"] = 0
        state.pop("_live_slots", None)
        state.pop("_nbt_flat", None)
        state["_direct_grads"] = False
        return state

    def _csr(self, data, use_edge_attr: bool) -> GraphCSR:
        ei = data.edge_index
        ea = getattr(data, "edge_attr", None) if use_edge_attr else None
        n = int(data.x.shape[0])
        key = (ei.data_ptr(), ei._version, tuple(ei.shape), n,
               None if ea is None else (ea.data_ptr(), ea._version, tuple(ea.shape)))
        csr = self._csr_cache.get(key)
        if csr is None:
            if len(self._csr_cache) >= 8:
                self._csr_cache.clear()
            csr = GraphCSR(ei, ea, n)
            csr._keepalive = (ei, ea)        # pointers in the key stay valid while cached
            self._csr_cache[key] = csr
        return csr

    @staticmethod
    def _p(t):
        if t is None:
            return None
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise _lib.NscError("GNN parameters must be contiguous float32 tensors on the HIP device")
        return t.data_ptr()

    def _live_tensors(self):
        """Every tensor NscGatModel points into.  Looked up through the modules' own ``_parameters`` / ``_buffers`` dicts (the
        (dict, name) pairs are collected once: the module tree is fixed after __init__): always the current tensor, without
        nn.Module.__getattr__ -- 113 attribute walks per call were a third of the host time of a pipelined step (round 4;
        nn.Module.parameters() / buffers() before that: 40 %, round 3)."""
        slots = self.__dict__.get("_live_slots")
        if slots is None:
            slots = [(self.input_proj._parameters, "weight"), (self.input_proj._parameters, "bias")]
            for bn in [self.input_norm] + list(self.batch_norms):
                slots += [(bn._parameters, "weight"), (bn._parameters, "bias"), (bn._buffers, "running_mean"),
                          (bn._buffers, "running_var")]
            slots += [(self.output_proj._parameters, "weight"), (self.output_proj._parameters, "bias")]
            if self.residual_proj is not None:
                slots += [(self.residual_proj._parameters, "weight"), (self.residual_proj._parameters, "bias")]
            for conv in self.convs:
                slots += [(conv.lin_src._parameters, "weight"), (conv._parameters, "att_src"), (conv._parameters, "att_dst"),
                          (conv._parameters, "bias")]
                if conv.lin_edge is not None:
                    slots += [(conv.lin_edge._parameters, "weight"), (conv._parameters, "att_edge")]
            self.__dict__["_live_slots"] = slots
        return [d[n] for d, n in slots]

    def _model_struct(self, live=None) -> _lib.GatModel:
        """NscGatModel over the live parameter storage.  Rebuilt (and the attention vectors re-folded
        by nsc_gat_fold_weights) only when a parameter was modified or moved.  ``live``: the caller's _live_tensors()."""
        key = tuple((t.data_ptr(), t._version) for t in (live if live is not None else self._live_tensors()))
        if self._struct_cache is not None and self._struct_cache[0] == key:
            return self._struct_cache[1]
        m = self._build_struct()
        L = _lib.lib()
        dev = self.input_proj.weight.device
        folded = torch.empty(int(L.nsc_gat_folded_floats(C.byref(m))), dtype=torch.float32, device=dev)
        m.folded = folded.data_ptr()
        with torch.cuda.device(dev):
            st = L.nsc_gat_fold_weights(C.byref(m), _lib.ptr(folded), _lib.stream_ptr(dev))
        _lib.check(st, "nsc_gat_fold_weights")
        self._struct_cache = (key, m, folded)
        self._fold_generation = getattr(self, "_fold_generation", 0) + 1     # a captured forward bakes `folded` in: see
        return m                                                              # ShardedDescriptorPath._enhance

    def _train_struct(self) -> _lib.GatModel:
        """NscGatModel for nsc_gat_forward_train / nsc_gat_backward: pointers into the live parameter storage, no folded
        attention vectors (the training kernels compute the attention dot products themselves, so an optimizer step
        does not force a re-fold before the next forward).  Rebuilt only when a parameter's storage moves."""
        key = tuple(t.data_ptr() for t in self._live_tensors())
        c = getattr(self, "_train_struct_cache", None)
        if c is None or c[0] != key:
            c = (key, self._build_struct())
            self._train_struct_cache = c
        return c[1]

    def _build_struct(self) -> _lib.GatModel:
        m = _lib.GatModel()
        m.in_dim, m.hidden, m.out_dim = self.input_dim, self.hidden_dim, self.output_dim
        m.n_layers = self.n_layers
        m.edge_dim = self.edge_dim or 0
        m.residual = int(bool(self.residual))
        m.bn_eps = float(self.input_norm.eps)
        m.negative_slope = float(self.convs[0].negative_slope)
        p = self._p
        m.in_w, m.in_b = p(self.input_proj.weight), p(self.input_proj.bias)
        bn = self.input_norm
        m.in_bn_w, m.in_bn_b = p(bn.weight), p(bn.bias)
        m.in_bn_mean, m.in_bn_var = p(bn.running_mean), p(bn.running_var)
        m.out_w, m.out_b = p(self.output_proj.weight), p(self.output_proj.bias)
        if self.residual_proj is not None:
            m.res_w, m.res_b = p(self.residual_proj.weight), p(self.residual_proj.bias)
        for l, (conv, bn) in enumerate(zip(self.convs, self.batch_norms)):
            ly = m.layers[l]
            ly.lin_w = p(conv.lin_src.weight)
            ly.att_src, ly.att_dst = p(conv.att_src), p(conv.att_dst)
            if conv.lin_edge is not None:
                ly.lin_edge_w, ly.att_edge = p(conv.lin_edge.weight), p(conv.att_edge)
            ly.bias = p(conv.bias)
            ly.bn_w, ly.bn_b = p(bn.weight), p(bn.bias)
            ly.bn_mean, ly.bn_var = p(bn.running_mean), p(bn.running_var)
        return m

    def _train_params(self):
        """Parameters in the order NscGatGrads lists them."""
        ps = [self.input_proj.weight, self.input_proj.bias, self.input_norm.weight, self.input_norm.bias,
              self.output_proj.weight, self.output_proj.bias]
        if self.residual_proj is not None:
            ps += [self.residual_proj.weight, self.residual_proj.bias]
        for conv, bn in zip(self.convs, self.batch_norms):
            ps += [conv.lin_src.weight, conv.att_src, conv.att_dst]
            if conv.lin_edge is not None:
                ps += [conv.lin_edge.weight, conv.att_edge]
            ps += [conv.bias, bn.weight, bn.bias]
        return ps

    def _run_train(self, data, use_edge_attr: bool):
        """model.train() forward with autograd (trainer.py:205): batch-statistics BatchNorm, feature and
        attention dropout (counter-based masks seeded from torch's CPU generator)."""
        x = data.x
        _lib.require_cuda(x, "data.x")
        x = x.to(torch.float32).contiguous()
        csr = self._csr(data, use_edge_attr)
        # seed of the counter-based dropout masks from torch's CPU generator; a captured step (GNNTrainer) keeps it in a
        # device word instead (self._seed_dev), rewritten before every replay
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (self.dropout > 0 and getattr(self, "_seed_dev", None) is None) else 0
        return _GatTrainFunction.apply(self, x, csr, float(self.dropout), seed, *self._train_params())

    def train_step_direct(self, data, anchor_idx, positive_idx, negative_idx, margin: float, scale: float):
        """forward (train mode) + TripletLoss + backward WITHOUT autograd (GNNTrainer's steps, round 4): nsc_gat_forward_train,
        nsc_triplet_loss (which returns d loss / d emb) and nsc_gat_backward adding into the parameters' existing .grad
        tensors -- what ``loss = criterion(model(graph)[a], ...); loss.backward()`` (trainer.py:205-213) computes, minus the
        autograd plumbing around the three ABI calls (a (N, 800) multiply by the upstream scalar, two dtype / layout copies and
        the engine's bookkeeping per batch).  Indices: int64 device tensors.  Returns the loss (0-d tensor)."""
        x = data.x
        _lib.require_cuda(x, "data.x")
        x = x.detach().to(torch.float32).contiguous()
        edge_attr = getattr(data, 'edge_attr', None)
        csr = self._csr(data, edge_attr is not None and self.edge_dim is not None)
        params = self._train_params()
        if any(p.grad is None or not p.grad.is_contiguous() for p in params):
            raise _lib.NscError("train_step_direct adds into existing contiguous .grad tensors")
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (self.dropout > 0 and getattr(self, "_seed_dev", None) is None) else 0
        out, state = _train_forward_raw(self, x, csr, float(self.dropout), seed)
        L = _lib.lib()
        dev = x.device
        n, d, t = int(out.shape[0]), int(out.shape[1]), int(anchor_idx.numel())
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        grad = torch.empty_like(out)
        nbytes = L.nsc_triplet_workspace_bytes(t)
        ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            st = L.nsc_triplet_loss(_lib.ptr(out), _lib.ptr(anchor_idx), _lib.ptr(positive_idx), _lib.ptr(negative_idx), t, n, d,
                                    float(margin), float(scale), _lib.ptr(loss), _lib.ptr(grad), _lib.ptr(ws), nbytes,
                                    _lib.stream_ptr(dev))
        _lib.check(st, "nsc_triplet_loss")
        _train_backward_raw(self, state, grad, [p.grad for p in params], True, False)
        return loss[0]

    def _run(self, data, use_edge_attr: bool, want_alpha: bool):
        x = data.x
        _lib.require_cuda(x, "data.x")
        if self.input_proj.weight.device != x.device:
            raise _lib.NscError("SpectralGNN and data must be on the same HIP device")
        if self.training:
            raise _lib.NscError("forward_with_attention is an inference helper: call model.eval() first")
        dev = x.device
        x = x.detach().to(torch.float32).contiguous()
        n = int(x.shape[0])
        csr = self._csr(data, use_edge_attr)
        L = _lib.lib()
        m = self._model_struct()
        g = csr.struct()
        out = torch.empty((n, self.output_dim), dtype=torch.float32, device=dev)
        alpha = (torch.zeros((self.n_layers, csr.capacity), dtype=torch.float32, device=dev)
                 if want_alpha else None)
        nbytes = L.nsc_gat_workspace_bytes(C.byref(m), n)
        ws = _ws(dev, nbytes, "gat")
        with torch.cuda.device(dev):
            st = L.nsc_gat_forward_ex(C.byref(m), C.byref(g), _lib.ptr(x), _lib.ptr(csr.edge_attr),
                                      _lib.ptr(out), _lib.ptr(alpha), _lib.ptr(ws), nbytes,
                                      {False: 0, True: 1, "shared_b": 3, "lds_tiled": 4, "generic": 8}[getattr(self, "coresident", False)],
                                      _lib.stream_ptr(dev))
        _lib.check(st, "nsc_gat_forward_ex")
        return out, alpha, csr

    # -- reference API ----------------------------------------------------------------------
    def forward(self, data) -> torch.Tensor:
        """model.py:96-153: data.x (N,in), data.edge_index (2,E), optional data.edge_attr (E,edge_dim)."""
        edge_attr = getattr(data, 'edge_attr', None)
        use_edge = edge_attr is not None and self.edge_dim is not None       # :126
        if self.training:
            return self._run_train(data, use_edge)
        out, _, _ = self._run(data, use_edge, False)
        return out

    def forward_with_attention(self, data) -> tuple:
        """model.py:155-201: convs are called WITHOUT edge_attr there; returns
        (embeddings, [(edge_index_with_self_loops (2,E'), alpha (E',1))] * n_layers) in PyG's
        order (kept edges in input order, then the N self loops)."""
        out, alpha, csr = self._run(data, False, True)
        nnz = int(csr.row_ptr[-1].item())
        eid = csr.eid[:nnz].long()
        ei = data.edge_index
        keep = ei[0] != ei[1]
        rank = torch.cumsum(keep.long(), 0) - 1                              # position among kept edges
        n_kept = int(keep.sum().item())
        tgt = torch.repeat_interleave(torch.arange(csr.n_nodes, device=ei.device),
                                      (csr.row_ptr[1:] - csr.row_ptr[:-1]).long())
        pos = torch.where(eid >= 0, rank[eid.clamp(min=0)], n_kept + tgt)
        loops = torch.arange(csr.n_nodes, device=ei.device)
        ei_full = torch.cat([ei[:, keep], torch.stack([loops, loops])], 1)
        weights = []
        for l in range(self.n_layers):
            a = torch.empty(nnz, dtype=torch.float32, device=ei.device)
            a[pos] = alpha[l, :nnz]
            weights.append((ei_full, a.unsqueeze(1)))
        return out, weights

    def get_embedding_dim(self) -> int:
        return self.output_dim


class LocalUpdateGNN(nn.Module):
    """model.py:208-281: wrapper whose local update is a stub in the reference -- always full graph."""

    def __init__(self, gnn: SpectralGNN, k_hops: int = 3):
        super().__init__()
        self.gnn = gnn
        self.k_hops = k_hops

    def forward(self, data, update_nodes: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self.gnn(data)                                                # :248-255

    def forward_local(self, data, center_node: int, k_hops: Optional[int] = None) -> torch.Tensor:
        embeddings = self.gnn(data)                                          # :277-281
        return embeddings[center_node:center_node + 1]


def create_spectral_gnn(input_dim: int = 800, hidden_dim: int = 256, output_dim: int = 800,
                        n_layers: int = 3, dropout: float = 0.1, use_local_updates: bool = True,
                        local_update_hops: int = 3, edge_dim: int = None) -> nn.Module:
    """model.py:284-324"""
    base_gnn = SpectralGNN(input_dim=input_dim, hidden_dim=hidden_dim, output_dim=output_dim,
                           n_layers=n_layers, n_heads=1, dropout=dropout, residual=True,
                           edge_dim=edge_dim)
    if use_local_updates:
        return LocalUpdateGNN(base_gnn, k_hops=local_update_hops)
    return base_gnn
