"""Range-image projection and gap interpolation on MI355X.

Mirrors the hot-path surface of the reference's src/encoding/range_image.py:
``RangeImageProjector`` (:92-232) and ``interpolate_range_image`` (:15-89).  Both run inside the
HIP encoder kernels (csrc/nsc_encoder.hip); these wrappers exist so callers that reach for
``encoder.projector.project(...)`` keep working (range image and, for (N,4) input, the intensity image
of keep_intensity=True).  ``unproject``/visualisation helpers are not part of the path and are not provided.
"""
from typing import Optional, Tuple

import numpy as np
import torch

from .. import _lib


def _as_points(points, device):
    """(N,3|4) array-like -> contiguous float32 device tensor + stride."""
    if isinstance(points, torch.Tensor):
        t = points.detach()
        if t.dtype != torch.float32:
            t = t.float()
        t = t.to(device).contiguous()
    else:
        a = np.ascontiguousarray(points, dtype=np.float32)
        t = torch.from_numpy(a).to(device)
    if t.dim() != 2 or t.shape[1] not in (3, 4):
        raise ValueError(f"points must be (N,3) or (N,4), got {tuple(t.shape)}")
    return t, int(t.shape[1])


class RangeImageProjector:
    """Same constructor and attributes as the reference (range_image.py:102-127)."""

    def __init__(self, n_elevation: int = 64, n_azimuth: int = 360,
                 elevation_range: Tuple[float, float] = (-24.8, 2.0),
                 max_range: float = 80.0, min_range: float = 1.0, device="cuda",
                 elev_float64: bool = True):
        self.n_elevation = n_elevation
        self.n_azimuth = n_azimuth
        self.max_range = max_range
        self.min_range = min_range
        self.elevation_min = np.deg2rad(elevation_range[0])
        self.elevation_max = np.deg2rad(elevation_range[1])
        self.device = torch.device(device)
        self.elev_float64 = elev_float64

    def _params(self, n_bins=50, target_rows=None, epsilon=1e-8, interpolate=True):
        p = _lib.EncParams()
        p.n_elevation = self.n_elevation
        p.n_azimuth = self.n_azimuth
        p.n_bins = n_bins
        p.target_rows = target_rows if target_rows is not None else min(self.n_elevation, 16)
        p.elev_min_rad = float(self.elevation_min)
        p.elev_max_rad = float(self.elevation_max)
        p.min_range = float(self.min_range)
        p.max_range = float(self.max_range)
        p.epsilon = float(epsilon)
        p.interpolate = int(bool(interpolate))
        p.elev_f64 = int(bool(self.elev_float64))
        return p

    def project_images(self, points, interpolate=False) -> Tuple[torch.Tensor, torch.Tensor]:
        """Device-side projection of ONE cloud: returns (raw, interpolated) (E,A) float32 tensors."""
        from .spectral_encoder import _run_encode_clouds, _default_lut
        dev = self.device
        t, stride = _as_points(points, dev)
        off = torch.tensor([0, t.shape[0]], dtype=torch.int64, device=dev)
        p = self._params(interpolate=interpolate)
        lut = _default_lut(dev)
        _, raw, itp = _run_encode_clouds(t, off, 1, int(t.shape[0]), stride, p, lut, want_images=True)
        return raw[0], itp[0]

    def project(self, points: np.ndarray, keep_intensity: bool = True
                ) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        """range_image.py:129-232.  Returns (range_image, intensity_image) as (E,A) float32 ndarrays; the
        intensity image is None for (N,3) input or keep_intensity=False, as in the reference."""
        from .spectral_encoder import _run_encode_clouds, _default_lut
        dev = self.device
        t, stride = _as_points(points, dev)
        n = int(t.shape[0])
        off = torch.tensor([0, n], dtype=torch.int64, device=dev)
        p = self._params(interpolate=False)
        _, raw, _ = _run_encode_clouds(t, off, 1, n, stride, p, _default_lut(dev), want_images=True)
        if not (keep_intensity and stride == 4):
            return raw[0].cpu().numpy(), None
        inten = torch.empty_like(raw)
        with torch.cuda.device(dev):
            st = _lib.lib().nsc_project_intensity(_lib.ptr(t), _lib.ptr(off), 1, n, p, _lib.ptr(raw), _lib.ptr(inten),
                                                  _lib.stream_ptr(dev))
        _lib.check(st, "nsc_project_intensity")
        return raw[0].cpu().numpy(), inten[0].cpu().numpy()


def project_to_range_image(points: np.ndarray, n_elevation: int = 64, n_azimuth: int = 360,
                           device="cuda") -> np.ndarray:
    """Convenience wrapper of range_image.py:302-323: (N,3|4) points -> (n_elevation, n_azimuth) range image."""
    return RangeImageProjector(n_elevation=n_elevation, n_azimuth=n_azimuth, device=device).project(
        points, keep_intensity=False)[0]


def interpolate_range_image(range_image: np.ndarray, method: str = "linear",
                            device="cuda") -> np.ndarray:
    """range_image.py:15-89 on the device: (rows, 360) image with 0 for empty pixels -> interpolated image,
    bit-identical to the reference, method 'linear' (:52-64) or 'nearest' (:66-75)
    (nsc_interpolate_range_images_ex)."""
    if method not in ("linear", "nearest"):
        # (the reference silently skips the column interpolation for any other string and only copies empty rows)
        raise ValueError(f"method must be 'linear' or 'nearest', got {method!r}")
    from .spectral_encoder import _default_lut
    dev = torch.device(device)
    img = torch.as_tensor(np.ascontiguousarray(range_image, dtype=np.float32)).to(dev)
    batched = img.dim() == 3
    x = img if batched else img.unsqueeze(0)
    if x.shape[2] != 360:
        raise ValueError("range images must have 360 azimuth columns")
    out = torch.empty_like(x)
    with torch.cuda.device(dev):
        st = _lib.lib().nsc_interpolate_range_images_ex(_lib.ptr(x), int(x.shape[0]), int(x.shape[1]),
                                                        _lib.ptr(_default_lut(dev)), 1 if method == "linear" else 2,
                                                        _lib.ptr(out), _lib.stream_ptr(dev))
    _lib.check(st, "nsc_interpolate_range_images_ex")
    res = out if batched else out[0]
    return res.cpu().numpy()
