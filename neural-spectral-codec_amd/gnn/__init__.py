"""Mirror of the reference's ``gnn`` package for the hot path (src/gnn/model.py)."""
from .model import SpectralGNN, LocalUpdateGNN, create_spectral_gnn
from .gat_conv import GATConv

__all__ = ["SpectralGNN", "LocalUpdateGNN", "create_spectral_gnn", "GATConv"]
