"""CPU restatement (pure torch, float32) of the reference GNN forward for the GAT hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by the product package.

Parity status: PARITY UNPINNED BY THE REFERENCE.  The arithmetic of this stage lives in the
third-party dependency torch-geometric==2.4.0 (reference requirements.txt:3, setup.py:26; call
sites src/gnn/model.py:16,75-84,127,129,180), which is absent from /root/reference and from this
image, and the reference holds no tests or golden vectors at that boundary.  This file restates
the published GATConv(heads=1, concat=False, add_self_loops=True, fill_value='mean',
negative_slope=0.2, bias=True) algorithm and src/gnn/model.py:96-153 around
it; it is pinned by an independent dense-adjacency formulation (gatconv_dense) and by invariants
(tests/test_gat_oracle.py), not by reference outputs.
"""
import torch
import torch.nn.functional as F


def add_self_loops_mean(edge_index, edge_attr, n):
    """PyG remove_self_loops + add_self_loops(fill_value='mean'): self loops are dropped, N loops
    (i,i) are appended LAST, their attribute = mean of the attributes of edges entering i."""
    keep = edge_index[0] != edge_index[1]
    ei = edge_index[:, keep]
    loops = torch.arange(n, dtype=edge_index.dtype, device=edge_index.device)
    ea = None
    if edge_attr is not None:
        ea = edge_attr[keep]
        d = ea.shape[1]
        s = torch.zeros((n, d), dtype=ea.dtype, device=ea.device).index_add_(0, ei[1], ea)
        cnt = torch.zeros(n, dtype=ea.dtype, device=ea.device).index_add_(
            0, ei[1], torch.ones(ei.shape[1], dtype=ea.dtype, device=ea.device))
        loop_ea = s / cnt.clamp(min=1).unsqueeze(1)
        ea = torch.cat([ea, loop_ea], 0)
    ei = torch.cat([ei, torch.stack([loops, loops])], 1)
    return ei, ea


def gatconv_reference(x, edge_index, edge_attr, lin_w, att_src, att_dst, lin_edge_w, att_edge, bias,
                      negative_slope=0.2, return_alpha=False):
    """GATConv 2.4.0 forward, heads=1.  edge_index[0]=source j, [1]=target i."""
    n = x.shape[0]
    h = x @ lin_w.t()
    a_src = (h * att_src.view(1, -1)).sum(-1)
    a_dst = (h * att_dst.view(1, -1)).sum(-1)
    ei, ea = add_self_loops_mean(edge_index, edge_attr if lin_edge_w is not None else None, n)
    j, i = ei[0], ei[1]
    logit = a_src[j] + a_dst[i]
    if ea is not None and lin_edge_w is not None:
        logit = logit + ((ea @ lin_edge_w.t()) * att_edge.view(1, -1)).sum(-1)
    logit = F.leaky_relu(logit, negative_slope)
    m = torch.full((n,), float("-inf"), dtype=x.dtype).scatter_reduce(0, i, logit.detach(), "amax",
                                                                      include_self=True)
    p = torch.exp(logit - m[i])                       # the max shift carries no gradient (softmax identity)
    den = torch.zeros(n, dtype=x.dtype).index_add_(0, i, p)
    alpha = p / (den[i] + 1e-16)
    out = torch.zeros_like(h).index_add_(0, i, alpha.unsqueeze(1) * h[j])
    out = out + bias
    return (out, ei, alpha) if return_alpha else out


def gatconv_dense(x, edge_index, edge_attr, lin_w, att_src, att_dst, lin_edge_w, att_edge, bias,
                  negative_slope=0.2):
    """Independent formulation: dense (N,N) masked softmax.  Requires a simple graph (no duplicate
    edges); used only to cross-check gatconv_reference on small graphs."""
    n = x.shape[0]
    h = (x.double() @ lin_w.double().t())
    s = h @ att_src.double().view(-1)
    d = h @ att_dst.double().view(-1)
    logits = torch.full((n, n), float("-inf"), dtype=torch.float64)        # [target i, source j]
    has_edge = torch.zeros((n, n), dtype=torch.bool)
    ea_dense = None
    if edge_attr is not None and lin_edge_w is not None:
        ea_dense = torch.zeros((n, n, edge_attr.shape[1]), dtype=torch.float64)
    for e in range(edge_index.shape[1]):
        j, i = int(edge_index[0, e]), int(edge_index[1, e])
        if i == j:
            continue
        assert not has_edge[i, j], "dense cross-check needs a simple graph"
        has_edge[i, j] = True
        if ea_dense is not None:
            ea_dense[i, j] = edge_attr[e].double()
    if ea_dense is not None:
        cnt = has_edge.sum(1).clamp(min=1).double()
        loop = (ea_dense * has_edge.unsqueeze(-1)).sum(1) / cnt.unsqueeze(1)
        for i in range(n):
            ea_dense[i, i] = loop[i]
    for i in range(n):
        has_edge[i, i] = True
    v = None
    if ea_dense is not None:
        v = lin_edge_w.double().t() @ att_edge.double().view(-1)         # (edge_dim,)
    raw = s.view(1, n) + d.view(n, 1)
    if v is not None:
        raw = raw + ea_dense @ v
    raw = F.leaky_relu(raw, negative_slope)
    logits = torch.where(has_edge, raw, logits)
    alpha = torch.softmax(logits, dim=1)
    return (alpha @ h + bias.double()).float()


def _bn(x, bn, training):
    if training:
        mean = x.mean(0)
        var = x.var(0, unbiased=False)
    else:
        mean, var = bn.running_mean, bn.running_var
    return (x - mean) / torch.sqrt(var + bn.eps) * bn.weight + bn.bias


def forward_from_state(sd, gnn_cfg, x, ei, ea, training=False):
    """src/gnn/model.py:96-153 from a state dict (tensors may require grad -> torch autograd gives
    the reference gradients).  gnn_cfg = (n_layers, residual, edge_dim)."""
    n_layers, residual, edge_dim = gnn_cfg
    use_edge = ea is not None and edge_dim is not None                   # :126

    class _B:                                                            # tiny BN view over the state dict
        def __init__(self, prefix):
            self.weight, self.bias = sd[prefix + ".weight"], sd[prefix + ".bias"]
            self.running_mean, self.running_var = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
            self.eps = 1e-5

    x_input = x
    h = x @ sd["input_proj.weight"].t() + sd["input_proj.bias"]          # :116
    h = F.relu(_bn(h, _B("input_norm"), training))                       # :117-118
    for l in range(n_layers):
        h_prev = h
        pre = f"convs.{l}."
        h = gatconv_reference(
            h, ei, ea if use_edge else None, sd[pre + "lin_src.weight"], sd[pre + "att_src"],
            sd[pre + "att_dst"], sd.get(pre + "lin_edge.weight") if use_edge else None,
            sd.get(pre + "att_edge") if use_edge else None, sd[pre + "bias"])          # :126-129
        h = _bn(h, _B(f"batch_norms.{l}"), training)                     # :132
        if l < n_layers - 1:
            h = F.relu(h)                                                # :135-137 (dropout off)
        if residual and 0 < l < n_layers - 1:
            h = h + h_prev                                               # :140-141
    out = h @ sd["output_proj.weight"].t() + sd["output_proj.bias"]      # :144
    if residual:
        if "residual_proj.weight" in sd:
            out = out + x_input @ sd["residual_proj.weight"].t() + sd["residual_proj.bias"]
        else:
            out = out + x_input                                          # :147-151
    return out


def forward_reference(model, data, training=False, dtype=torch.float32):
    """src/gnn/model.py:96-153 with the module's own parameters, on the CPU in float32 (the reference's
    arithmetic) or, with dtype=torch.float64, the same restatement evaluated in double precision (the
    checker for element-wise bounds: its own rounding error is negligible against the 1e-4 bar).
    training=True uses batch statistics in BatchNorm and NO dropout (dropout is stochastic and
    unseeded in the reference; parity runs use dropout=0)."""
    gnn = getattr(model, "gnn", model)                                  # LocalUpdateGNN wrapper :230
    sd = {k: (v.detach().cpu().to(dtype) if v.dtype.is_floating_point else v.detach().cpu())
          for k, v in gnn.state_dict().items()}
    x = data.x.detach().cpu().to(dtype)
    ei = data.edge_index.detach().cpu()
    ea = getattr(data, "edge_attr", None)
    ea = ea.detach().cpu().to(dtype) if ea is not None else None
    return forward_from_state(sd, (gnn.n_layers, gnn.residual, gnn.edge_dim), x, ei, ea, training)


def assert_within_bar(out, ref, rtol=1e-4, atol=1e-6, what="GAT output"):
    """north_star bar for the GAT output, ELEMENT-WISE: |out - ref| <= rtol * |ref| + atol for every element."""
    out, ref = out.detach().cpu().double(), ref.detach().cpu().double()
    assert out.shape == ref.shape, f"{what}: shape {tuple(out.shape)} vs {tuple(ref.shape)}"
    d = (out - ref).abs()
    bound = rtol * ref.abs() + atol
    worst = (d / bound).max().item() if d.numel() else 0.0
    assert bool((d <= bound).all()), (f"{what}: {(d > bound).sum().item()} of {d.numel()} elements outside "
                                      f"|d| <= {rtol}*|ref| + {atol} (worst {worst:.2f}x the bound, max |d| {d.max().item():.3g})")
    return worst


def reference_gradients(model, data, loss_fn, dtype=torch.float32):
    """Reference gradients by torch autograd through the restatement (train-mode BatchNorm, no
    dropout).  loss_fn(embeddings) -> scalar.  Returns (embeddings, {state-dict key: grad}, grad_x, loss).
    dtype=torch.float64 evaluates the same restatement in double precision (used to measure how far float32
    arithmetic itself sits from the exact result on ill-conditioned inputs)."""
    gnn = getattr(model, "gnn", model)
    sd = {}
    for k, v in gnn.state_dict().items():
        t = v.detach().cpu().to(dtype).clone() if v.dtype.is_floating_point else v.detach().cpu().clone()
        if v.dtype.is_floating_point and "running_" not in k:
            t.requires_grad_(True)
        sd[k] = t
    for l in range(gnn.n_layers):                       # lin_dst aliases lin_src (one parameter)
        sd[f"convs.{l}.lin_dst.weight"] = sd[f"convs.{l}.lin_src.weight"]
    x = data.x.detach().cpu().to(dtype).clone().requires_grad_(True)
    ei = data.edge_index.detach().cpu()
    ea = getattr(data, "edge_attr", None)
    ea = ea.detach().cpu().to(dtype) if ea is not None else None
    emb = forward_from_state(sd, (gnn.n_layers, gnn.residual, gnn.edge_dim), x, ei, ea, training=True)
    loss = loss_fn(emb)
    loss.backward()
    grads = {k: v.grad for k, v in sd.items() if v.requires_grad and "lin_dst" not in k and v.grad is not None}
    return emb.detach(), grads, x.grad, loss.detach()


def triplet_loss_reference(emb, ia, ip, in_, margin):
    """src/gnn/trainer.py:62-68"""
    a, p, n = emb[ia], emb[ip], emb[in_]
    pos = torch.sum((a - p) ** 2, dim=1)
    neg = torch.sum((a - n) ** 2, dim=1)
    return torch.relu(pos - neg + margin).mean()


def randomize_bn_stats(model, seed=1):
    """The synthetic-setup helper lives in the product package (synth.randomize_bn_stats): bench.py and the multi-rank
    workers need it for SETUP and must not import oracle/ for that.  Kept here as an alias for the tests."""
    from neural_spectral_codec_amd import synth
    return synth.randomize_bn_stats(model, seed)
