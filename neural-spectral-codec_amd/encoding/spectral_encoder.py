"""SpectralEncoder on MI355X -- drop-in for the reference's src/encoding/spectral_encoder.py:24-261.

Same constructor, attributes and methods; every method launches the hand-written HIP kernels of
csrc/nsc_encoder.hip through the C ABI (include/nsc.h).  The module must live on a HIP device
(``.to('cuda')`` as pipeline.py:73 / train_multi_dataset.py:271 do); there is no CPU fallback.

Additive API (not in the reference): ``encode_points_batch`` for packed batches of clouds.
"""
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .range_image import RangeImageProjector, _as_points

_LUT_CACHE = {}


def compute_bin_edges(alpha: torch.Tensor, n_bins: int, n_freqs: int, epsilon: float) -> torch.Tensor:
    """spectral_encoder.py:93-116, op for op (float32 torch ops on the host)."""
    t = torch.linspace(0, 1, n_bins + 1, device=alpha.device)
    bin_edges = (torch.exp(alpha * t) - 1) / (torch.exp(alpha) - 1 + epsilon)
    return bin_edges * n_freqs


def compute_bin_lut(alpha: float, n_bins: int, n_freqs: int, epsilon: float) -> torch.Tensor:
    """spectral_encoder.py:136-145: frequency index -> histogram bin (int32, host)."""
    a = torch.tensor(float(alpha), dtype=torch.float32)
    edges = compute_bin_edges(a, n_bins, n_freqs, epsilon)
    freq = torch.arange(n_freqs, dtype=torch.float32)
    lut = torch.searchsorted(edges, freq, right=True) - 1
    lut = torch.clamp(lut, 0, n_bins - 1).to(torch.int32)
    if bool((lut[1:] < lut[:-1]).any()):
        raise _lib.NscError("bin LUT is not monotone (alpha must be > 0)")
    return lut


def _lut_on(device, alpha, n_bins, n_freqs, epsilon):
    key = (str(device), float(alpha), n_bins, n_freqs, float(epsilon))
    lut = _LUT_CACHE.get(key)
    if lut is None:
        lut = compute_bin_lut(alpha, n_bins, n_freqs, epsilon).to(device)
        _LUT_CACHE[key] = lut
    return lut


def _default_lut(device):
    return _lut_on(device, 2.0, 50, 181, 1e-8)


def _workspace(device, nbytes):
    if nbytes == 0:
        return None
    # one scratch buffer per (device, stream, host thread): launches issued on different streams may overlap (the
    # pipelined step alternates its encoder launches over two streams); bounded, capture-safe (_lib.ScratchCache)
    return _lib.scratch.get(device, nbytes, "enc")


def _run_encode_clouds(pts, offsets, n_clouds, total_points, stride, p, lut, want_images=False,
                       out=None):
    """nsc_encode_clouds on the current stream.  pts/offsets/lut are device tensors."""
    L = _lib.lib()
    dev = pts.device
    D = p.target_rows * p.n_bins
    desc = out if out is not None else torch.empty((n_clouds, D), dtype=torch.float32, device=dev)
    raw = itp = None
    if want_images:
        raw = torch.empty((n_clouds, p.n_elevation, p.n_azimuth), dtype=torch.float32, device=dev)
        itp = torch.empty_like(raw)
    nbytes = L.nsc_encode_clouds_workspace_bytes(n_clouds, total_points, p)
    ws = _workspace(dev, nbytes)
    with torch.cuda.device(dev):
        st = L.nsc_encode_clouds(_lib.ptr(pts), _lib.ptr(offsets), n_clouds, total_points, stride, p,
                                 _lib.ptr(lut), _lib.ptr(desc), _lib.ptr(raw), _lib.ptr(itp),
                                 _lib.ptr(ws), nbytes, _lib.stream_ptr(dev))
    _lib.check(st, "nsc_encode_clouds")
    return desc, raw, itp


class SpectralEncoder(nn.Module):
    """Neural Spectral Histogram Encoder (reference spectral_encoder.py:24-261)."""

    def __init__(self, n_elevation: int = 64, n_azimuth: int = 360, n_bins: int = 50,
                 alpha: float = 2.0, learnable_alpha: bool = True, epsilon: float = 1e-8,
                 target_elevation_bins: int = 16, interpolate_empty: bool = True,
                 elevation_range: tuple = (-24.8, 2.0), device: str = 'cpu',
                 elev_float64: bool = True):
        super().__init__()
        self.n_elevation = n_elevation
        self.n_azimuth = n_azimuth
        self.n_bins = n_bins
        self.epsilon = epsilon
        self.target_elevation_bins = target_elevation_bins
        self.interpolate_empty = interpolate_empty
        self._device = device
        if learnable_alpha:                                       # :74-77
            self.alpha = nn.Parameter(torch.tensor(alpha, dtype=torch.float32))
        else:
            self.register_buffer('alpha', torch.tensor(alpha, dtype=torch.float32))
        self.projector = RangeImageProjector(n_elevation=n_elevation, n_azimuth=n_azimuth,
                                             elevation_range=elevation_range,
                                             elev_float64=elev_float64)          # :80-84
        self.n_freqs = n_azimuth // 2 + 1                         # :88
        self.output_dim = target_elevation_bins * n_bins          # :91

    # -- host-side helpers ------------------------------------------------------------------
    def _compute_bin_edges(self, alpha: torch.Tensor) -> torch.Tensor:
        return compute_bin_edges(alpha, self.n_bins, self.n_freqs, self.epsilon)

    def _dev(self) -> torch.device:
        dev = self.alpha.device                                   # :224 alpha.device decides placement
        if dev.type != "cuda":
            raise _lib.NscError(
                "SpectralEncoder is on %s: move it to the MI355X with .to('cuda') "
                "(pipeline.py:73, train_multi_dataset.py:271); there is no CPU fallback" % dev)
        self.projector.device = dev
        return dev

    def _params(self) -> _lib.EncParams:
        return self.projector._params(n_bins=self.n_bins, target_rows=self.target_elevation_bins,
                                      epsilon=self.epsilon, interpolate=self.interpolate_empty)

    def _lut(self, dev) -> torch.Tensor:
        # alpha lives on the device; read it back only when it was modified (a .item() is a full
        # host-device sync and would stall the launch pipeline on every call)
        key = (self.alpha.data_ptr(), self.alpha._version)
        if getattr(self, "_alpha_key", None) != key:
            self._alpha_host = float(self.alpha.detach())
            self._alpha_key = key
        return _lut_on(dev, self._alpha_host, self.n_bins, self.n_freqs, self.epsilon)

    # -- reference API ------------------------------------------------------------------------
    def encode_range_image(self, range_image: torch.Tensor) -> torch.Tensor:
        """(rows, n_azimuth) -> (target_elevation_bins * n_bins,)   spectral_encoder.py:160-204"""
        return self.forward(range_image.unsqueeze(0))[0]

    def encode_points(self, points: np.ndarray) -> torch.Tensor:
        """(N,3)|(N,4) -> (target_elevation_bins * n_bins,)          spectral_encoder.py:206-229"""
        return self.encode_points_batch([points])[0]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(batch, rows, n_azimuth) -> (batch, output_dim)            spectral_encoder.py:231-249"""
        dev = self._dev()
        if x.dim() != 3 or x.shape[2] != self.n_azimuth:
            raise ValueError(f"expected (batch, rows, {self.n_azimuth}), got {tuple(x.shape)}")
        x = x.detach().to(device=dev, dtype=torch.float32).contiguous()
        n, rows = int(x.shape[0]), int(x.shape[1])
        out = torch.empty((n, self.output_dim), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = _lib.lib().nsc_encode_range_images(_lib.ptr(x), n, rows, self._params(),
                                                    _lib.ptr(self._lut(dev)), _lib.ptr(out),
                                                    _lib.stream_ptr(dev))
        _lib.check(st, "nsc_encode_range_images")
        return out

    def encode_batch(self, range_images: torch.Tensor) -> torch.Tensor:
        """spectral_encoder.py:251-261"""
        return self.forward(range_images)

    # -- additive API -------------------------------------------------------------------------
    def encode_points_batch(self, clouds: Union[Sequence[np.ndarray], Tuple[torch.Tensor, torch.Tensor]],
                            return_images: bool = False, out: Optional[torch.Tensor] = None):
        """Encode many clouds in one launch.

        clouds: a list of (N_i,3|4) arrays, or (points, offsets) with points (sum N,3|4) float32 and
        offsets (B+1,) int64 -- device tensors are used in place (no copy).
        Returns (B, output_dim) [, raw (B,E,A), interpolated (B,E,A)]."""
        dev = self._dev()
        if isinstance(clouds, tuple) and len(clouds) == 2 and isinstance(clouds[1], torch.Tensor):
            pts, stride = _as_points(clouds[0], dev)
            off = clouds[1].to(device=dev, dtype=torch.int64).contiguous()
            n = int(off.numel()) - 1
            total = int(pts.shape[0])
        elif len(clouds) > 0 and all(isinstance(c, torch.Tensor) for c in clouds):
            ts = [c.detach().to(device=dev, dtype=torch.float32) for c in clouds]       # device tensors stay put
            n = len(ts)
            if len({int(t.shape[1]) for t in ts}) != 1:
                raise ValueError("all clouds of a batch must have the same width (3 or 4)")
            pts, stride = _as_points(ts[0] if n == 1 else torch.cat(ts, 0), dev)
            off = torch.tensor(np.concatenate([[0], np.cumsum([int(t.shape[0]) for t in ts])]),
                               dtype=torch.int64, device=dev)
            total = int(pts.shape[0])
        else:
            arrs = [np.ascontiguousarray(c, dtype=np.float32) for c in clouds]
            n = len(arrs)
            if n == 0:
                return torch.empty((0, self.output_dim), dtype=torch.float32, device=dev)
            widths = {a.shape[1] for a in arrs}
            if len(widths) != 1:
                raise ValueError("all clouds of a batch must have the same width (3 or 4)")
            sizes = [a.shape[0] for a in arrs]
            packed = np.concatenate(arrs, 0) if n > 1 else arrs[0]
            pts, stride = _as_points(packed, dev)
            off = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int64, device=dev)
            total = int(packed.shape[0])
        desc, raw, itp = _run_encode_clouds(pts, off, n, total, stride, self._params(), self._lut(dev),
                                            want_images=return_images, out=out)
        return (desc, raw, itp) if return_images else desc
