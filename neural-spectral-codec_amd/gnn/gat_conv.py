"""Parameter container for one GATConv layer with torch-geometric 2.4.0's state-dict layout.

The reference builds ``torch_geometric.nn.GATConv(hidden, hidden, heads=1, concat=False,
dropout=p, edge_dim=edge_dim)`` (src/gnn/model.py:75-84).  PyG is a third-party dependency that is
not vendored; this module reproduces its parameter names, shapes and initialisation so checkpoints
written by the reference (src/gnn/trainer.py:480-495) load unchanged:
    lin_src.weight (H,H)   lin_dst.weight (same tensor: PyG sets lin_dst = lin_src for int in_channels)
    att_src, att_dst (1,1,H)   lin_edge.weight (H,edge_dim)   att_edge (1,1,H)   bias (H)
The arithmetic runs inside nsc_gat_forward (csrc/nsc_gat.hip); the layer is never called alone.
"""
import math

import torch
import torch.nn as nn


def _glorot(t: torch.Tensor):
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-a, a)


class GATConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, concat: bool = True,
                 negative_slope: float = 0.2, dropout: float = 0.0, add_self_loops: bool = True,
                 edge_dim=None, fill_value="mean", bias: bool = True):
        super().__init__()
        if heads != 1:
            raise NotImplementedError("the hot path uses heads=1 (src/gnn/model.py:315)")
        if not add_self_loops or fill_value != "mean" or not bias:
            raise NotImplementedError("only GATConv defaults are on the hot path")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.heads, self.concat = heads, concat
        self.negative_slope, self.dropout = negative_slope, dropout
        self.edge_dim = edge_dim
        self.lin_src = nn.Linear(in_channels, out_channels, bias=False)
        self.lin_dst = self.lin_src
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        if edge_dim is not None:
            self.lin_edge = nn.Linear(edge_dim, out_channels, bias=False)
            self.att_edge = nn.Parameter(torch.empty(1, heads, out_channels))
        else:
            self.lin_edge = None
            self.register_parameter("att_edge", None)
        self.bias = nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        _glorot(self.lin_src.weight)
        if self.lin_edge is not None:
            _glorot(self.lin_edge.weight)
            _glorot(self.att_edge)
        _glorot(self.att_src)
        _glorot(self.att_dst)
        with torch.no_grad():
            self.bias.zero_()

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # PyG >= 2.5 stores a single `lin.weight`; 2.0-2.4 store lin_src/lin_dst (aliases)
        k = prefix + "lin.weight"
        if k in state_dict:
            w = state_dict.pop(k)
            state_dict.setdefault(prefix + "lin_src.weight", w)
        if prefix + "lin_src.weight" in state_dict:
            state_dict.setdefault(prefix + "lin_dst.weight", state_dict[prefix + "lin_src.weight"])
        elif prefix + "lin_dst.weight" in state_dict:
            state_dict[prefix + "lin_src.weight"] = state_dict[prefix + "lin_dst.weight"]
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, *a, **k):
        raise NotImplementedError("GATConv layers run fused inside SpectralGNN.forward (nsc_gat_forward)")

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, heads={self.heads}, edge_dim={self.edge_dim}"
