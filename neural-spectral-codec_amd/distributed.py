"""Multi-GPU layout of the descriptor path: one process per GPU, ``torch.distributed`` (backend
``nccl`` = RCCL over xGMI on ROCm; ``gloo`` in the CPU tests).

The reference is single-process; this is the MI355X-side design:

* keyframes shard CONTIGUOUSLY over ranks -- every cloud is independent, so the encoder needs no
  communication;
* ONE exchange step: an all-gather of the (N_r, 800) float32 descriptor shards (3 200 B per keyframe),
  after which every rank holds the full descriptor matrix (what stage-1 retrieval consumes,
  reference src/retrieval/two_stage_retrieval.py:145-202);
* the GNN runs on each rank's own node range plus a halo: the temporal graph is a chain with
  |i - j| <= M//2 (src/keyframe/graph_manager.py:520-532), so L GAT layers see at most L*(M//2)
  nodes beyond the shard (6 for the reference's L = 3, M = 5).  With eval-mode BatchNorm (pointwise)
  the owned rows are exactly the rows of the full-graph forward.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist

from .keyframe.graph_manager import Data, chain_edges, edge_features


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank: the first n_total % world ranks get one extra keyframe."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_descriptors(local: torch.Tensor, n_total: Optional[int] = None,
                           group=None, single_rank_too: bool = False) -> torch.Tensor:
    """All-gather (N_r, D) shards laid out by shard_range() into the full (N, D) matrix.

    Equal shards use a single all_gather_into_tensor (one RCCL collective, no copies); ragged
    shards are padded to the largest shard and trimmed."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not single_rank_too):
        return local
    world = dist.get_world_size(group)
    n_local, d = int(local.shape[0]), int(local.shape[1])
    if n_total is None:
        n_total = n_local * world
    base, rem = divmod(n_total, world)
    local = local.contiguous()
    if rem == 0:
        out = torch.empty((n_total, d), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    cap = base + 1
    padded = torch.zeros((cap, d), dtype=local.dtype, device=local.device)
    padded[:n_local] = local
    buf = torch.empty((world * cap, d), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        parts.append(buf[r * cap:r * cap + (hi - lo)])
    return torch.cat(parts, 0)


def halo_window(n_total: int, lo: int, hi: int, n_layers: int = 3, temporal_neighbors: int = 5
                ) -> Tuple[int, int]:
    """Node window [wlo, whi) a rank needs so rows [lo, hi) of an n_layers-deep GAT are exact."""
    halo = n_layers * (temporal_neighbors // 2)
    return max(0, lo - halo), min(n_total, hi + halo)


def shard_graph(desc_all: torch.Tensor, lo: int, hi: int, poses=None, n_layers: int = 3,
                temporal_neighbors: int = 5):
    """Sub-chain graph over the shard + halo window (node ids relative to the window start).
    Returns (Data, first owned row inside the window)."""
    n_total = int(desc_all.shape[0])
    wlo, whi = halo_window(n_total, lo, hi, n_layers, temporal_neighbors)
    n = whi - wlo
    edges = chain_edges(n, temporal_neighbors)
    dev = desc_all.device
    edge_index = (torch.from_numpy(edges.T.copy()).to(dev) if len(edges)
                  else torch.empty((2, 0), dtype=torch.long, device=dev))
    edge_attr = None
    if poses is not None and len(edges):
        edge_attr = torch.from_numpy(edge_features(poses[wlo:whi], edges)).to(dev)
    g = Data(x=desc_all[wlo:whi], edge_index=edge_index, edge_attr=edge_attr, num_nodes=n)
    return g, lo - wlo


def hw_queue_classes(streams, spin_cycles: int = 250_000):
    """Which of ``streams`` share a hardware queue?  HIP multiplexes its streams over a few HSA queues
    (GPU_MAX_HW_QUEUES, 4 by default), and which streams end up together depends on the addresses the runtime's queues
    happen to get -- it differs from process to process.  Two streams on one queue execute in order: launches issued
    on them cannot overlap.  Probe (setup only, a few milliseconds): a one-thread spin kernel on each of two streams --
    distinct queues run them side by side, a shared queue one after the other.  Returns a class id per stream (equal
    ids share a queue), or None when the spin kernel is not available."""
    import time
    if not streams or not hasattr(torch.cuda, "_sleep"):
        return None
    dev = streams[0].device

    def timed(idx):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in idx:
            with torch.cuda.stream(streams[i]):
                torch.cuda._sleep(spin_cycles)
                torch.cuda._sleep(spin_cycles)
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0

    for i in range(len(streams)):
        timed([i])                                            # first use of a stream binds its queue
    one = min(timed([0]) for _ in range(3))
    reps, cls = [], []
    for i in range(len(streams)):
        found = None
        for r in reps:
            if min(timed([r, i]), timed([r, i])) > 1.6 * one:
                found = cls[r]
                break
        if found is None:
            found = len(reps)
            reps.append(i)
        cls.append(found)
    return cls


def concurrent_streams(device, n: int, candidates: int = 12):
    """``n`` streams that sit on ``n`` different hardware queues (see hw_queue_classes), none of them the queue of the
    caller's current stream when that can be had.  Falls back to plain new streams when the probe is unavailable or the
    device has fewer queues than asked for.  All at the default priority: a high-priority encoder (or GNN) stream was
    measured 40-60 % slower per step (round 3: the other side starves and the buffer rotation stalls)."""
    cur = torch.cuda.current_stream(device)
    pool = [torch.cuda.Stream(device) for _ in range(candidates)]
    cls = hw_queue_classes([cur] + pool)
    if cls is None:
        return pool[:n], None
    picked, used = [], {cls[0]}
    for st, c in zip(pool, cls[1:]):
        if c not in used:
            picked.append(st)
            used.add(c)
        if len(picked) == n:
            break
    if len(picked) < n:                                       # not enough queues beside the caller's: share the caller's
        for st, c in zip(pool, cls[1:]):
            if st not in picked and c not in {cls[1 + pool.index(q)] for q in picked}:
                picked.append(st)
            if len(picked) == n:
                break
    for st in pool:                                           # still short: whatever is left
        if len(picked) == n:
            break
        if st not in picked:
            picked.append(st)
    return picked, {"caller": cls[0], "picked": [cls[1 + pool.index(q)] for q in picked], "queues_seen": len(set(cls))}


class ShardedDescriptorPath:
    """encode (local shard) -> all-gather descriptors -> GNN on shard + halo (local rows out).

    ``encoder`` needs ``encode_points_batch(clouds)``; ``gnn`` is called as ``gnn(data)``.  The graph
    of a fixed (n_total, poses) layout is built once and reused: only ``x`` changes per step.

    With equal shards the GNN does not wait for the full all-gather: the ranks first exchange only
    their 2 x halo boundary rows (one tiny all-gather, 38 KB per rank for halo 6), the big all-gather
    (3.3 MB per rank at 1 024 keyframes) is issued asynchronously and overlaps the GNN forward, and the
    step waits for it at the end (the gathered matrix is what stage-1 retrieval consumes).

    ``pipeline=True`` software-pipelines consecutive steps on HIP streams of their own: the encoder of batch k+1 is
    issued without waiting for batch k, and the exchange + GNN of batch k run on a second stream under it (results stay
    valid for ``_PIPE_BUFFERS - 1`` further steps).  ``encoder_streams=2`` (the default) alternates consecutive encoder
    launches over TWO streams, so that they overlap: while the four workgroups per CU of launch k drain through their
    finish phase, workgroups of launch k+1 already stream (the finish of one launch, its launch ramp and the completion
    marker of its stream hide under the streaming of the next).  The resident grid is then up to five encoder
    workgroups per CU -- 5 x 27.9 KB of the 160 KB of LDS, 5 x 80 of the 512 VGPRs of a SIMD lane -- so the GNN is
    launched in its LDS-free, <= 56-VGPR form (``gnn.coresident``, NSC_GAT_CORESIDENT), two waves of which fit in the
    112 registers left (tests/test_abi_cpu.py::test_coresident_register_budget).  Descriptor buffers rotate;
    ``step`` returns without waiting and its results are valid after ``synchronize()`` (or once the caller's stream
    has waited on ``last_event``)."""

    def __init__(self, encoder, gnn, n_total: int, poses=None, temporal_neighbors: int = 5,
                 n_layers: int = 3, group=None, overlap: bool = True, pipeline: bool = False,
                 encoder_streams: int = 2, gnn_streams: int = 1, gnn_graph: Optional[bool] = None,
                 rehearse_collectives: bool = False):
        self.encoder, self.gnn, self.group = encoder, gnn, group
        self.pipeline = pipeline
        self.encoder_streams = max(1, int(encoder_streams))
        self.gnn_streams = max(1, int(gnn_streams))
        # replay the GNN forward as a captured hipGraph (pipeline mode, eval).  Default: on for one rank (round 3: +1.3 %
        # per step at 200 steps, host issue 0.12 -> 0.09 ms per step), off beside RCCL (its threads issue HIP calls of their
        # own while a capture is open; not measurable here)
        self.gnn_graph = gnn_graph
        self._gnn_graphs, self._gnn_warm = {}, {}
        self.coresident_gnn = True         # pipeline mode: launch the GNN in its NSC_GAT_CORESIDENT form
        self._k = 0
        self._streams = None
        self.queue_classes = None          # hardware-queue classes of the pipeline's streams (set with the streams)
        self.gnn_waits = 0                 # times the encoder stream had to wait for a GNN pass (buffer still being read)
        self.collective_events = None      # optional [(start, end) HIP events]: step k times its descriptor all-gather with
                                           # pair k % len on the stream the collective is issued on (bench.py, N > 1)
        self.last_event = None
        self.n_total, self.poses = n_total, poses
        self.M, self.L = temporal_neighbors, n_layers
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.lo, self.hi = shard_range(n_total, self.rank, self.world)
        # rehearse_collectives: a ONE-rank process group still issues every collective of the N > 1 step (RCCL world-1
        # rehearsal on a one-GPU box: the backend's stream semantics and its kernels beside the encoder grid, bench.py
        # NSC_BENCH_RCCL_WORLD1=1 / tests); without it one rank skips the exchange
        self._collect = self.world > 1 or (bool(rehearse_collectives) and dist.is_initialized())
        if self._collect:
            self.gnn_streams = 1           # one communicator: its collectives stay on one stream, in one order on all ranks
        if self.gnn_graph is None:
            self.gnn_graph = not self._collect
        self.halo = n_layers * (temporal_neighbors // 2)
        n_local = self.hi - self.lo
        # pipeline mode hides the whole exchange + GNN under the next encoder: ONE all-gather per step then (every
        # RCCL kernel has to find room beside a resident encoder grid), no boundary-row pre-exchange
        self.overlap = (overlap and not pipeline and self._collect and n_total % self.world == 0
                        and n_local >= self.halo)
        self._graph = None
        self._own0 = 0
        self._wlo = 0
        self._desc_all = {}                # gathered-matrix buffers, one per pipeline slot

    def _window_graph(self, like: torch.Tensor):
        """Sub-chain graph over [lo-halo, hi+halo) n [0, n_total) with a placeholder x."""
        wlo, whi = halo_window(self.n_total, self.lo, self.hi, self.L, self.M)
        n = whi - wlo
        edges = chain_edges(n, self.M)
        dev = like.device
        edge_index = (torch.from_numpy(edges.T.copy()).to(dev) if len(edges)
                      else torch.empty((2, 0), dtype=torch.long, device=dev))
        edge_attr = None
        if self.poses is not None and len(edges):
            edge_attr = torch.from_numpy(edge_features(self.poses[wlo:whi], edges)).to(dev)
        self._graph = Data(x=None, edge_index=edge_index, edge_attr=edge_attr, num_nodes=n)
        self._own0, self._wlo = self.lo - wlo, wlo

    # -- two-stream software pipeline -----------------------------------------------------------
    _PIPE_BUFFERS = 4      # descriptor buffers in rotation: the encoder stream never has to wait for the GNN
    probe_queues = True    # pick the pipeline's streams on distinct hardware queues (concurrent_streams); a test that
                           # runs several ranks as threads of one process turns the timing probe off

    def _pipe_setup(self, device):
        inner = getattr(self.gnn, "gnn", self.gnn)
        if hasattr(inner, "coresident"):
            # the kernel set this path asked for, whatever was set before (True, False or "shared_b")
            inner.coresident = self.coresident_gnn if self.coresident_gnn == "shared_b" else bool(self.coresident_gnn)
        n_local, d = self.hi - self.lo, int(getattr(self.encoder, "output_dim", 800))
        nb = self._PIPE_BUFFERS
        if device.type == "cuda":
            # the encoder streams must be able to run side by side, and the GNN beside them: one hardware queue each
            ns = self.encoder_streams + self.gnn_streams
            if self.probe_queues:
                sts, self.queue_classes = concurrent_streams(device, ns)
            else:
                sts = [torch.cuda.Stream(device) for _ in range(ns)]
            sE, sG = sts[:self.encoder_streams], sts[self.encoder_streams:]
            cur = torch.cuda.current_stream(device)
            for st in sE + sG:
                st.wait_stream(cur)
            self._ev_enc = [torch.cuda.Event() for _ in range(nb)]
            self._ev_gnn = [torch.cuda.Event() for _ in range(nb)]
        else:
            # host tensors (the gloo tests of the buffer rotation and the exchange): no streams, every launch is
            # synchronous, so the same issue order runs as a plain sequence
            sE, sG = None, None
            self._ev_enc = self._ev_gnn = [None] * nb
        self._streams = (sE, sG)
        self._desc = [torch.empty((n_local, d), dtype=torch.float32, device=device) for _ in range(nb)]

    def _step_pipelined(self, clouds, encoder_events, inputs_ready):
        device = torch.device(self.encoder.alpha.device)
        if self._streams is None:
            self._pipe_setup(device)
        sEs, sGs = self._streams
        nb = self._PIPE_BUFFERS
        i = self._k % nb
        if sEs is None:
            local = self.encoder.encode_points_batch(clouds, out=self._desc[i])
            res = self._exchange_and_enhance(local)
            self._k += 1
            return res
        sE = sEs[self._k % len(sEs)]                          # consecutive launches alternate over the encoder streams
        sG = sGs[self._k % len(sGs)]                          # ... and the exchange + GNN passes over the GNN streams
        caller = torch.cuda.current_stream(device)
        if not inputs_ready:
            # the clouds may still be being written on the caller's stream: order the encoder behind it
            sE.wait_stream(caller)
        with torch.cuda.stream(sE):
            # batch k-nb must have been read out of this buffer.  With nb buffers in rotation that GNN pass
            # has normally finished long ago: ask the host first, so that the encoder stream carries no
            # cross-stream barrier packet (each one costs ~20 us of idle between two encoder launches).
            if self._k >= nb and not self._ev_gnn[i].query():
                sE.wait_event(self._ev_gnn[i])
                self.gnn_waits += 1
            if encoder_events is not None and encoder_events[0] is not None:
                encoder_events[0].record(sE)
            local = self.encoder.encode_points_batch(clouds, out=self._desc[i])
            done = encoder_events[1] if encoder_events is not None else self._ev_enc[i]
            done.record(sE)
        with torch.cuda.stream(sG):
            sG.wait_event(done)
            res = self._exchange_and_enhance(local)
            self._ev_gnn[i].record(sG)
        # emb (and a gathered matrix that is not one of the rotating slots) came out of the caching allocator on
        # stream G and are read by the caller on ITS stream after synchronize(): tell the allocator, or the block
        # could be handed to a later step on stream G while those reads are still in flight
        for t in res:
            if t.is_cuda and not (self.gnn_graph and self._gnn_graphs.get(i) is not None):
                t.record_stream(caller)      # (a captured forward's output lives in its graph's own pool)
        self.last_event = self._ev_gnn[i]
        self._k += 1
        return res

    def synchronize(self):
        """Make the caller's current stream wait for every step issued so far (pipeline mode)."""
        if self._streams is not None and self._streams[0] is not None:
            cur = torch.cuda.current_stream(self._streams[0][0].device)
            for st in list(self._streams[0]) + list(self._streams[1]):
                cur.wait_stream(st)

    def step(self, clouds, encoder_events=None, inputs_ready: bool = False):
        """clouds: this rank's shard (list of arrays or (points, offsets) device tensors).
        Returns (all descriptors (n_total, D), enhanced embeddings of the owned rows (hi-lo, D)).
        ``encoder_events``: optional (start, end) torch.cuda.Event pair recorded around the encoder launch on the stream
        it is issued on (start may be None: only the completion of the launch is time-stamped).
        ``inputs_ready`` (pipeline mode): the clouds are already complete in HBM (nothing pending on the caller's
        stream writes them), so the encoder stream does not have to wait for the caller's stream -- that
        cross-stream wait costs ~10 us of idle between two encoder launches."""
        if self.pipeline:
            return self._step_pipelined(clouds, encoder_events, inputs_ready)
        if encoder_events is not None and encoder_events[0] is not None:
            encoder_events[0].record()
        local = self.encoder.encode_points_batch(clouds)
        if encoder_events is not None:
            encoder_events[1].record()
        res = self._exchange_and_enhance(local)
        self._k += 1
        return res

    def _gather(self, out, local):
        """The step's descriptor all-gather (blocking form), bracketed by the caller's HIP events when asked for."""
        ce = self.collective_events
        if ce and local.is_cuda:
            a, b = ce[self._k % len(ce)]
            a.record()
            dist.all_gather_into_tensor(out, local, group=self.group)
            b.record()
        else:
            dist.all_gather_into_tensor(out, local, group=self.group)

    def _exchange_and_enhance(self, local):
        if self._graph is None:
            self._window_graph(local)
        n_local, h = self.hi - self.lo, self.halo
        work = None
        if self.overlap:
            # 1. boundary rows only (tiny, blocking): [first h rows | last h rows] of every rank
            d = int(local.shape[1])
            mine = torch.cat([local[:h], local[n_local - h:]], 0)
            edges_all = torch.empty((self.world * 2 * h, d), dtype=local.dtype, device=local.device)
            ce = self.collective_events if local.is_cuda else None
            if ce:                                   # (this form's window spans the GNN forward the big gather overlaps)
                ce[self._k % len(ce)][0].record()
            dist.all_gather_into_tensor(edges_all, mine, group=self.group)
            # 2. the full matrix, asynchronously
            slot = 0                                 # (overlap is off in pipeline mode, see __init__)
            if slot not in self._desc_all:
                self._desc_all[slot] = torch.empty((self.n_total, d), dtype=local.dtype, device=local.device)
            gathered = self._desc_all[slot]
            work = dist.all_gather_into_tensor(gathered, local.contiguous(), group=self.group,
                                               async_op=True)
            parts = []
            if self.rank > 0:                       # previous rank's last h rows
                parts.append(edges_all[(self.rank - 1) * 2 * h + h:(self.rank - 1) * 2 * h + 2 * h])
            parts.append(local)
            if self.rank < self.world - 1:          # next rank's first h rows
                parts.append(edges_all[(self.rank + 1) * 2 * h:(self.rank + 1) * 2 * h + h])
            self._graph.x = torch.cat(parts, 0) if len(parts) > 1 else local
            desc_all = gathered
        elif self.pipeline and self._collect and self.n_total % self.world == 0:
            # pipeline mode: ONE all-gather straight into this step's slot of the rotating gathered-matrix buffers
            # (valid for _PIPE_BUFFERS - 1 further steps, like the descriptor buffers; no allocation per step)
            slot = self._k % self._PIPE_BUFFERS
            if slot not in self._desc_all:
                self._desc_all[slot] = torch.empty((self.n_total, int(local.shape[1])), dtype=local.dtype,
                                                   device=local.device)
            desc_all = self._desc_all[slot]
            self._gather(desc_all, local.contiguous())
            self._graph.x = desc_all[self._wlo:self._wlo + self._graph.num_nodes]
        else:
            ce = self.collective_events if (local.is_cuda and self._collect) else None
            if ce:
                ce[self._k % len(ce)][0].record()
            desc_all = all_gather_descriptors(local, self.n_total, self.group, self._collect)    # fresh tensor per step
            if ce:
                ce[self._k % len(ce)][1].record()
            self._graph.x = desc_all[self._wlo:self._wlo + self._graph.num_nodes]
        emb = self._enhance()
        if work is not None:
            work.wait()
            if self.collective_events and local.is_cuda:
                self.collective_events[self._k % len(self.collective_events)][1].record()
        return desc_all, emb[self._own0:self._own0 + n_local]

    def _enhance(self):
        """GNN forward over the window graph.  Pipeline mode on a HIP device: the forward's kernel launches are captured
        once per rotating buffer into a hipGraph (torch.cuda.CUDAGraph: the C ABI only enqueues kernels on the stream it
        is given, so a capturing stream records them) and replayed -- one graph launch per step instead of eight kernel
        launches plus their Python plumbing (host issue 77 -> 19 us at 1 024 keyframes, device 77 -> 57 us alone; round
        3).  A capture is keyed on the input's storage and the model's parameter versions: new weights or a new input
        buffer recapture; an input that is not one of the rotating buffers (ragged shards) runs eagerly."""
        x = self._graph.x
        inner = getattr(self.gnn, "gnn", self.gnn)
        if not (self.gnn_graph and self.pipeline and x.is_cuda and not torch.is_grad_enabled()
                and not getattr(self.gnn, "training", False)
                and (hasattr(inner, "_live_tensors") or hasattr(inner, "parameters"))):     # (a plain callable runs eagerly)
            return self.gnn(self._graph)
        slot = self._k % self._PIPE_BUFFERS
        folded = None
        live = inner._live_tensors() if hasattr(inner, "_live_tensors") else list(inner.parameters()) + list(inner.buffers())
        if hasattr(inner, "_model_struct"):
            # the capture bakes in the folded attention vectors the model caches: make sure they exist for the current
            # weights, and key on their generation -- an optimizer that writes through a multi-tensor kernel does not bump
            # the parameters' version counters, GNNTrainer drops the cache after every step instead (a new generation)
            inner._model_struct(live)
            folded = inner._struct_cache[2]
        key = (x.data_ptr(), tuple(x.shape), tuple((t.data_ptr(), t._version) for t in live),
               getattr(inner, "coresident", False), getattr(inner, "_fold_generation", 0))
        ent = self._gnn_graphs.get(slot)
        if ent is not None and ent[0] == key:
            ent[1].replay()
            return ent[2]
        if self._gnn_warm.get(slot) != key:              # first sight of this (buffer, weights): run it eagerly once
            self._gnn_warm[slot] = key                   # (lazy set-up inside the forward must not be captured)
            return self.gnn(self._graph)
        try:
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg, stream=torch.cuda.current_stream(x.device), capture_error_mode="thread_local"):
                out = self.gnn(self._graph)
            cg.replay()                                  # the capture itself ran nothing
        except Exception as ex:  # noqa: BLE001 -- whatever keeps a capture from closing: issue the forward eagerly from now on
            import logging
            logging.getLogger(__name__).warning("hipGraph capture of the GNN forward failed (%s: %s); issuing it eagerly "
                                                "from now on", type(ex).__name__, ex)
            self.gnn_graph = False
            self._gnn_graphs.clear()
            return self.gnn(self._graph)
        self._gnn_graphs[slot] = (key, cg, out, folded)      # (folded: kept alive as long as the capture that reads it)
        return out

    def release(self):
        """Drop the captured GNN forwards and the rotating buffers; the scratch buffers of the Python layer that captures
        ran through are pinned until ``_lib.scratch.release()``."""
        self._gnn_graphs.clear()
        self._gnn_warm.clear()
        self._desc_all.clear()


def all_reduce_gradients(params, group=None, average: bool = False):
    """Sum (or average) the .grad of ``params`` over the ranks with ONE collective: gradients are
    flattened into a single bucket (the GNN has 613 920 float32 parameters = 2.46 MB, far below the
    size where several buckets would overlap anything), all-reduced, and copied back."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    o = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[o:o + n].view_as(g))
        o += n


def split_triplets(triplets, rank: int, world: int):
    """Contiguous slice of a triplet batch for this rank + the weight local/total that makes the sum of
    the per-rank mean losses equal the reference's global mean (trainer.py:68 loss.mean())."""
    n = len(triplets)
    lo, hi = shard_range(n, rank, world)
    return triplets[lo:hi], ((hi - lo) / n if n else 0.0)
