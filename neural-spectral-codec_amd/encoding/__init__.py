"""Mirror of the reference's ``encoding`` package for the hot path (src/encoding/__init__.py)."""
from .range_image import RangeImageProjector, interpolate_range_image
from .spectral_encoder import SpectralEncoder

__all__ = ["SpectralEncoder", "RangeImageProjector", "interpolate_range_image"]
