"""ctypes binding of csrc/libnsc_hip.so (include/nsc.h).  No fallback: if the library is missing
or a call fails, this raises."""
import collections
import ctypes as C
import os
import threading

import torch  # noqa: F401  -- loads torch's libamdhip64 first so the library shares its HIP runtime

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libnsc_hip.so")


class NscError(RuntimeError):
    pass


class EncParams(C.Structure):
    """struct NscEncParams (include/nsc.h)"""
    _fields_ = [
        ("n_elevation", C.c_int32), ("n_azimuth", C.c_int32), ("n_bins", C.c_int32),
        ("target_rows", C.c_int32), ("elev_min_rad", C.c_double), ("elev_max_rad", C.c_double),
        ("min_range", C.c_float), ("max_range", C.c_float), ("epsilon", C.c_float),
        ("interpolate", C.c_int32), ("elev_f64", C.c_int32),
    ]


class GatLayer(C.Structure):
    """struct NscGatLayer"""
    _fields_ = [(n, C.c_void_p) for n in (
        "lin_w", "att_src", "att_dst", "lin_edge_w", "att_edge", "bias",
        "bn_w", "bn_b", "bn_mean", "bn_var")]


ABI_VERSION = 4            # NSC_ABI_VERSION of include/nsc.h
GAT_MAX_LAYERS = 8
GAT_MAX_EDGE_DIM = 8


class GatModel(C.Structure):
    """struct NscGatModel"""
    _fields_ = [
        ("in_dim", C.c_int32), ("hidden", C.c_int32), ("out_dim", C.c_int32), ("n_layers", C.c_int32),
        ("edge_dim", C.c_int32), ("residual", C.c_int32), ("bn_eps", C.c_float),
        ("negative_slope", C.c_float),
        ("in_w", C.c_void_p), ("in_b", C.c_void_p),
        ("in_bn_w", C.c_void_p), ("in_bn_b", C.c_void_p), ("in_bn_mean", C.c_void_p), ("in_bn_var", C.c_void_p),
        ("out_w", C.c_void_p), ("out_b", C.c_void_p), ("res_w", C.c_void_p), ("res_b", C.c_void_p),
        ("folded", C.c_void_p),
        ("layers", GatLayer * GAT_MAX_LAYERS),
    ]


class Graph(C.Structure):
    """struct NscGraph"""
    _fields_ = [("n_nodes", C.c_int32), ("nnz", C.c_int32), ("row_ptr", C.c_void_p),
                ("src", C.c_void_p), ("eid", C.c_void_p), ("loop_attr", C.c_void_p),
                ("t_ptr", C.c_void_p), ("t_entry", C.c_void_p), ("tgt", C.c_void_p),
                ("band_entries", C.c_void_p), ("band", C.c_int32)]


class GatTrainCfg(C.Structure):
    """struct NscGatTrainCfg"""
    _fields_ = [("dropout_p", C.c_float), ("bn_momentum", C.c_float), ("seed", C.c_uint64),
                ("update_running_stats", C.c_int32), ("accumulate_grads", C.c_int32), ("seed_dev", C.c_void_p)]


class MineParams(C.Structure):
    """struct NscMineParams"""
    _fields_ = [("positive_distance_max", C.c_double), ("negative_distance_min", C.c_double),
                ("negative_distance_max", C.c_double), ("positive_temporal_min", C.c_int32),
                ("negative_temporal_min", C.c_int32), ("strategy", C.c_int32),
                ("triplets_per_anchor", C.c_int32), ("seed", C.c_uint64)]


class GatGradLayer(C.Structure):
    """struct NscGatGradLayer"""
    _fields_ = [(n, C.c_void_p) for n in (
        "lin_w", "att_src", "att_dst", "lin_edge_w", "att_edge", "bias", "bn_w", "bn_b")]


class GatGrads(C.Structure):
    """struct NscGatGrads"""
    _fields_ = [(n, C.c_void_p) for n in ("in_w", "in_b", "in_bn_w", "in_bn_b", "out_w", "out_b", "x",
                                          "res_w", "res_b")] + [
        ("layers", GatGradLayer * GAT_MAX_LAYERS)]


_lib = None

# every symbol include/nsc.h declares: (restype, argtypes)
_vp, _i32, _i64, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
_pp = C.POINTER(EncParams)
SYMBOLS = {
    "nsc_abi_version": (C.c_int, []),
    "nsc_status_string": (C.c_char_p, [C.c_int]),
    "nsc_enc_default_params": (None, [_pp]),
    "nsc_encode_clouds_workspace_bytes": (_sz, [_i32, _i64, _pp]),
    "nsc_encode_clouds_path": (C.c_int, [_i32, _i64, _i32, _pp]),
    "nsc_encode_clouds": (C.c_int, [_vp, _vp, _i32, _i64, _i32, _pp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "nsc_project_intensity": (C.c_int, [_vp, _vp, _i32, _i64, _pp, _vp, _vp, _vp]),
    "nsc_scatter_clouds": (C.c_int, [_vp, _vp, _i32, _i64, _i32, _pp, _vp, _vp]),
    "nsc_finish_images": (C.c_int, [_vp, _i32, _pp, _vp, _vp, _vp, _vp, _vp]),
    "nsc_interpolate_range_images": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "nsc_interpolate_range_images_ex": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _vp, _vp]),
    "nsc_encode_range_images": (C.c_int, [_vp, _i32, _i32, _pp, _vp, _vp, _vp]),
    "nsc_graph_workspace_bytes": (_sz, [_i32, _i64]),
    "nsc_graph_build_csr": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "nsc_graph_band_entries": (C.c_int, [C.POINTER(Graph), _vp, _i32, _vp, _vp, _vp]),
    "nsc_gat_folded_floats": (_sz, [C.POINTER(GatModel)]),
    "nsc_gat_fold_weights": (C.c_int, [C.POINTER(GatModel), _vp, _vp]),
    "nsc_gat_workspace_bytes": (_sz, [C.POINTER(GatModel), _i32]),
    "nsc_gat_forward": (C.c_int, [C.POINTER(GatModel), C.POINTER(Graph), _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "nsc_gat_forward_ex": (C.c_int, [C.POINTER(GatModel), C.POINTER(Graph), _vp, _vp, _vp, _vp, _vp, _sz, C.c_uint32,
                                     _vp]),
    "nsc_gat_gemm_tile": (C.c_int, [_i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "nsc_graph_transpose_workspace_bytes": (_sz, [_i32]),
    "nsc_graph_transpose": (C.c_int, [C.POINTER(Graph), _vp, _vp, _vp, _vp, _sz, _vp]),
    "nsc_gat_train_workspace_bytes": (_sz, [C.POINTER(GatModel), C.POINTER(Graph)]),
    "nsc_gat_forward_train": (C.c_int, [C.POINTER(GatModel), C.POINTER(Graph), _vp, _vp, C.POINTER(GatTrainCfg),
                                        _vp, _vp, _sz, _vp]),
    "nsc_gat_backward": (C.c_int, [C.POINTER(GatModel), C.POINTER(Graph), _vp, _vp, C.POINTER(GatTrainCfg), _vp,
                                   C.POINTER(GatGrads), _vp, _sz, _vp]),
    "nsc_w1_cdf": (C.c_int, [_vp, _i32, _i32, C.c_float, _i32, _vp, _vp]),
    "nsc_w1_distances": (C.c_int, [_vp, _i32, _i32, C.c_float, _vp, _i32, _vp, _vp, C.c_float, _vp, _vp]),
    "nsc_w1_distances_cdf": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _vp, _vp, C.c_float, _vp, _vp]),
    "nsc_revisit_queries": (C.c_int, [_vp, _i32, _i32, C.c_double, _vp, _vp]),
    "nsc_pairwise_l2": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "nsc_recall_rank": (C.c_int, [_vp, _vp, _vp, _i32, _i32, C.c_double, _vp, _vp]),
    "nsc_mine_triplets": (C.c_int, [_vp, _vp, _i32, _i32, C.POINTER(MineParams), _vp, _vp, _vp, _vp]),
    "nsc_mine_workspace_bytes": (_sz, [_i32, _i32]),
    "nsc_mine_triplets_ws": (C.c_int, [_vp, _vp, _i32, _i32, C.POINTER(MineParams), _vp, _vp, _vp, _vp, _sz, _vp]),
    "nsc_topk_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "nsc_topk_smallest": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "nsc_chain_graph_num_edges": (_i64, [_i32, _i32, _i32]),
    "nsc_build_chain_graph": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp]),
    "nsc_quantize_descriptors": (C.c_int, [_vp, _i32, _i32, C.c_float, _vp, _vp]),
    "nsc_dequantize_descriptors": (C.c_int, [_vp, _i32, _i32, C.c_float, _vp, _vp]),
    "nsc_record_bytes": (_sz, [_i32]),
    "nsc_pack_records": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "nsc_unpack_records": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "nsc_voxel_overlap_workspace_bytes": (_sz, [_i64, _i64]),
    "nsc_voxel_overlap": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _i32, _vp, C.c_double, _vp, _vp,
                                    _vp, _sz, _vp]),
    "nsc_triplet_workspace_bytes": (_sz, [_i32]),
    "nsc_triplet_loss": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, C.c_float, C.c_float, _vp, _vp, _vp,
                                   _sz, _vp]),
}


# include/nsc_debug.h: diagnostics outside the product ABI, bound when the library exports them (nsc_debug_burn: development
# builds only)
DEBUG_SYMBOLS = {
    "nsc_debug_point_bins": (C.c_int, [_vp, _i64, _i32, _pp, _vp, _vp, _vp]),
    "nsc_debug_burn": (C.c_int, [_i32, _i32, _i32, _vp, _sz, _vp]),
}


def lib():
    """Load the HIP library (once).  Raises NscError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NscError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in DEBUG_SYMBOLS.items():
            fn = getattr(L, name, None)
            if fn is not None:
                fn.restype = res
                fn.argtypes = args
        if L.nsc_abi_version() != ABI_VERSION:
            raise NscError("libnsc_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().nsc_status_string(status).decode()
        raise NscError(f"{what} failed: {msg} (status {status})")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(t, name):
    if not t.is_cuda:
        raise NscError(f"{name} must live on a HIP device (got {t.device}); the MI355X path has no "
                       "CPU fallback -- move the module/tensors to 'cuda'")


class ScratchCache:
    """Device scratch buffers of the Python layer (CSR build, GNN forward workspace, split encoder), one set per
    (device, stream, host thread, use): work on different streams may overlap and two host threads issuing on one stream
    interleave their launches, work issued by one thread on one stream is ordered.

    Bounded (round 4): at most ``cap`` keys, least recently used first, keys of host threads that no longer exist before
    any other.  A buffer is NEVER replaced by a bigger one -- a larger request adds a buffer to the key and the smaller
    ones stay, because a captured hipGraph replays the address it recorded; keys touched while a stream capture is open
    are pinned until ``release()`` (a replay does not come through here, so use-recency says nothing about them)."""

    def __init__(self, cap: int = 48):
        self.cap = cap
        self._lock = threading.Lock()
        self._d = collections.OrderedDict()          # key -> [pinned, [tensors, ascending size]]

    def get(self, device, nbytes: int, tag: str = ""):
        key = (str(device), torch.cuda.current_stream(device).cuda_stream, threading.get_ident(), tag)
        capturing = torch.cuda.is_current_stream_capturing()
        with self._lock:
            ent = self._d.get(key)
            if ent is None:
                ent = self._d[key] = [False, []]
            else:
                self._d.move_to_end(key)
            ent[0] = ent[0] or capturing
            for t in ent[1]:
                if t.numel() >= nbytes:
                    return t
            big = ent[1][-1].numel() if ent[1] else 0
            t = torch.empty(max(int(nbytes), 256, 2 * big if big else 0), dtype=torch.uint8, device=device)
            ent[1].append(t)
            if len(self._d) > self.cap:
                self._prune()
            return t

    def _prune(self):
        alive = {th.ident for th in threading.enumerate()}
        for dead_first in (True, False):
            for k in list(self._d):
                if len(self._d) <= self.cap:
                    return
                pinned = self._d[k][0]
                if pinned or (dead_first and k[2] in alive):
                    continue
                del self._d[k]

    def release(self, device=None):
        """Drop every buffer (of ``device``), pinned ones included: call when no captured hipGraph that ran through this
        cache will be replayed any more (e.g. after tearing down a ShardedDescriptorPath / GNNTrainer)."""
        with self._lock:
            for k in list(self._d):
                if device is None or k[0] == str(device):
                    del self._d[k]

    def __len__(self):
        return len(self._d)


scratch = ScratchCache()
