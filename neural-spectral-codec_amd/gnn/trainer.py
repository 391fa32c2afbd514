"""Training of the GNN on MI355X -- mirror of the reference's src/gnn/trainer.py.

``TripletLoss`` (:27-68) and ``GNNTrainer`` (:70-495) keep their names, constructor arguments, methods
(``train_epoch``, ``validate``, ``train``, ``save_checkpoint``, ``load_checkpoint``) and the checkpoint keys; the
compute inside them runs through the C ABI: full-graph forward + backward per 1 024-triplet batch
(nsc_gat_forward_train / nsc_gat_backward / nsc_triplet_loss), hard-negative mining on the device
(gnn/triplet_miner.py), loop-closure recall on the device.  The epoch / early-stopping / checkpoint loop around
them is host code, as in the reference.
"""
import logging
import os
import time
from pathlib import Path
from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from .. import _lib


class _TripletFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, ia, ip, in_, margin, scale):
        L = _lib.lib()
        dev = emb.device
        emb = emb.contiguous()
        n, d, t = int(emb.shape[0]), int(emb.shape[1]), int(ia.numel())
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        grad = torch.empty_like(emb) if emb.requires_grad else None
        nbytes = L.nsc_triplet_workspace_bytes(t)
        ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            st = L.nsc_triplet_loss(_lib.ptr(emb), _lib.ptr(ia), _lib.ptr(ip), _lib.ptr(in_), t, n, d,
                                    float(margin), float(scale), _lib.ptr(loss), _lib.ptr(grad), _lib.ptr(ws),
                                    nbytes, _lib.stream_ptr(dev))
        _lib.check(st, "nsc_triplet_loss")
        ctx.grad = grad
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        return (ctx.grad * g if ctx.grad is not None else None), None, None, None, None, None


class TripletLoss(nn.Module):
    """L(a,p,n) = mean(relu(|a-p|^2 - |a-n|^2 + margin))           trainer.py:27-68"""

    def __init__(self, margin: float = 0.1):
        super().__init__()
        self.margin = margin

    def forward_indexed(self, embeddings: torch.Tensor, anchor_idx, positive_idx, negative_idx,
                        scale: float = 1.0) -> torch.Tensor:
        """Fused gather + loss (+ gradient): embeddings (N,D), three index vectors (T,)."""
        _lib.require_cuda(embeddings, "embeddings")
        dev = embeddings.device
        n = int(embeddings.shape[0])
        idx = []
        for i in (anchor_idx, positive_idx, negative_idx):
            if not (isinstance(i, torch.Tensor) and i.is_cuda):
                # host-side indices (the miner's triplet list): embeddings[idx] would raise here in the reference
                # (trainer.py:207-209).  Device-resident indices are checked by the kernel instead (an index
                # outside [-N, N) turns the loss into NaN and writes nothing) -- no device sync on this path.
                h = np.asarray(i.cpu() if isinstance(i, torch.Tensor) else i).astype(np.int64)
                if h.size and (h.min() < -n or h.max() >= n):
                    raise IndexError(f"triplet index out of range for {n} embeddings")
                i = torch.from_numpy(h)
            idx.append(i.to(device=dev, dtype=torch.int64).contiguous())
        return _TripletFn.apply(embeddings.float(), idx[0], idx[1], idx[2], self.margin, scale)

    def forward(self, anchors: torch.Tensor, positives: torch.Tensor, negatives: torch.Tensor) -> torch.Tensor:
        """Reference signature: three already-gathered (T,D) tensors."""
        t = int(anchors.shape[0])
        emb = torch.cat([anchors, positives, negatives], 0)
        ar = torch.arange(t, device=emb.device, dtype=torch.int64)
        return self.forward_indexed(emb, ar, ar + t, ar + 2 * t)


class GNNTrainer:
    """Reference GNNTrainer (trainer.py:70-495): Adam(lr, weight_decay) over the model parameters, TripletLoss,
    gradient accumulation over ``accumulation_steps`` batches of ``batch_size`` triplets (1 024 / 4 are literals
    in the reference, :187-188).  ``use_multi_gpu`` is accepted for signature compatibility: the reference wraps
    the model in nn.DataParallel (:106-108); here multi-GPU training is one process per GPU -- under an initialised
    ``torch.distributed`` group every batch's triplets are split over the ranks and the gradients are all-reduced
    once per optimizer step."""

    def __init__(self, model: nn.Module, device: str = 'cuda', learning_rate: float = 5e-4,
                 weight_decay: float = 1e-5, margin: float = 0.1, checkpoint_dir: Optional[str] = None,
                 log_interval: int = 10, use_multi_gpu: bool = True, patience: int = 10,
                 batch_size: int = 1024, accumulation_steps: int = 4, use_graph: Optional[bool] = None,
                 direct_grads: bool = True):
        self.model = model.to(device)
        self.device = device
        self.patience = patience                                                                      # :112
        self.epochs_without_improvement = 0
        # Adam(lr, weight_decay) over the model parameters (:115-119).  On a HIP device: torch's FUSED implementation (one
        # multi-tensor kernel for the 25 parameter tensors instead of ~25 small launches per foreach op: the optimizer step was
        # 0.34 ms per 4 batches, round 3).  It updates parameters without bumping their version counters, which the eval-mode
        # forward's cache of folded attention vectors is keyed on: train_batches drops that cache after every step, and a
        # captured GNN forward is keyed on the fold generation (distributed.ShardedDescriptorPath._enhance).  Same state-dict
        # layout as the default implementation: checkpoints stay interchangeable with the reference's.
        fused = torch.device(device).type == "cuda" and torch.cuda.is_available() and os.environ.get("NSC_TRAINER_FUSED_ADAM", "1") == "1"
        try:
            self.optimizer = optim.Adam(model.parameters(), lr=learning_rate, weight_decay=weight_decay, fused=fused)
        except (TypeError, RuntimeError):       # an older torch without the fused path
            self.optimizer = optim.Adam(model.parameters(), lr=learning_rate, weight_decay=weight_decay)
        self.criterion = TripletLoss(margin=margin)                                                   # :121
        self.batch_size, self.accumulation_steps = batch_size, accumulation_steps
        # replay the per-batch step (forward + loss + backward) as a captured hipGraph from its second occurrence on.
        # None = decide per train_batches() call: on for one rank, OFF under an initialised process group of more than one
        # rank -- RCCL's own threads issue HIP calls while a capture is open (ShardedDescriptorPath turns its GNN capture
        # off beside RCCL for the same reason); True forces it (captures are opened in thread-local error mode).
        self.use_graph = use_graph
        self.max_captures = 4                      # captured steps kept (LRU); each pins a hipGraph + its workspace pool
        # the backward adds into the existing .grad tensors itself instead of handing gradients to autograd's AccumulateGrad
        self.direct_grads = direct_grads
        import collections
        self._captured, self._seen_once, self._capture_failed = collections.OrderedDict(), set(), False
        self._full_T = None                        # size of a full batch (slice) of the current train_batches() call
        # the reference creates 'checkpoints/' eagerly (:123-124); here the directory appears with the first save
        self.checkpoint_dir = Path(checkpoint_dir if checkpoint_dir is not None else 'checkpoints')
        self.log_interval = log_interval
        self.epoch = 0                                                                                # :129-135
        self.global_step = 0
        self.best_val_metric = 0.0
        self.train_losses = []
        self.val_metrics = []

    # -- validation (trainer.py:238-387) ------------------------------------------------------
    def _recall_ranks(self, embeddings: torch.Tensor, poses: np.ndarray, max_k: int,
                      distance_threshold: float, skip_frames: int):
        """Rank of the first correct candidate among the max_k nearest, per loop-closure query."""
        from ..retrieval.wasserstein import _topk
        L = _lib.lib()
        dev = embeddings.device
        emb = embeddings.detach().float().contiguous()
        n, d = int(emb.shape[0]), int(emb.shape[1])
        pos = torch.as_tensor(np.asarray(poses)[:, :3, 3].astype(np.float64)).to(dev).contiguous()
        first = torch.empty(n, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.nsc_revisit_queries(_lib.ptr(pos), n, skip_frames, float(distance_threshold),
                                             _lib.ptr(first), _lib.stream_ptr(dev)), "nsc_revisit_queries")
        qidx = first[first >= 0].contiguous()                     # one query per earlier frame i (:342-348)
        nq = int(qidx.numel())
        if nq == 0:
            return torch.empty(0, dtype=torch.int32, device=dev)
        ranks = torch.empty(nq, dtype=torch.int32, device=dev)
        k = min(max_k, n)
        chunk = 4096
        for q0 in range(0, nq, chunk):
            qs = qidx[q0:q0 + chunk].contiguous()
            dist = torch.empty((int(qs.numel()), n), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(L.nsc_pairwise_l2(_lib.ptr(emb), _lib.ptr(qs), int(qs.numel()), n, d, skip_frames,
                                             _lib.ptr(dist), _lib.stream_ptr(dev)), "nsc_pairwise_l2")
            idx, val = _topk(dist, k)
            idx = torch.where(torch.isinf(val), torch.full_like(idx, -1), idx)    # fewer than k candidates
            with torch.cuda.device(dev):
                _lib.check(L.nsc_recall_rank(_lib.ptr(pos), _lib.ptr(qs), _lib.ptr(idx.contiguous()),
                                             int(qs.numel()), k, float(distance_threshold),
                                             _lib.ptr(ranks[q0:q0 + chunk]), _lib.stream_ptr(dev)),
                           "nsc_recall_rank")
        return ranks

    def _compute_recall_loop_closure(self, embeddings, poses: np.ndarray, k: int,
                                     distance_threshold: float, skip_frames: int = 30):
        """trainer.py:306-387 -> (recall@k, n_queries)."""
        if not isinstance(embeddings, torch.Tensor):
            embeddings = torch.as_tensor(np.asarray(embeddings), dtype=torch.float32).to(self.device)
        ranks = self._recall_ranks(embeddings, poses, k, distance_threshold, skip_frames)
        nq = int(ranks.numel())
        if nq == 0:
            return 0.0, 0
        return float(((ranks > 0) & (ranks <= k)).sum().item()) / nq, nq

    def validate(self, val_graph, val_poses: np.ndarray, distance_threshold: float = 5.0,
                 skip_frames: int = 30):
        """trainer.py:238-304: eval forward + loop-closure recall@1/5/10 (one top-10 pass serves all three)."""
        self.model.eval()
        with torch.no_grad():
            embeddings = self.model(val_graph.to(self.device))
        ranks = self._recall_ranks(embeddings, val_poses, 10, distance_threshold, skip_frames)
        nq = int(ranks.numel())
        rec = lambda k: (float(((ranks > 0) & (ranks <= k)).sum().item()) / nq) if nq else 0.0   # noqa: E731
        return {'recall@1': rec(1), 'recall@5': rec(5), 'recall@10': rec(10), 'n_queries': nq}

    def train_epoch(self, graph, triplet_miner, poses: np.ndarray, descriptors: np.ndarray,
                    sequence_ids: np.ndarray = None, n_triplets_per_anchor: int = 1) -> float:
        """trainer.py:137-236: mine triplets for the epoch, shuffle them, run the accumulation loop."""
        t0 = time.perf_counter()
        triplets = triplet_miner.mine_triplets(descriptors=descriptors, poses=poses,
                                               n_triplets_per_anchor=n_triplets_per_anchor,
                                               sequence_ids=sequence_ids)                             # :161-166
        if len(triplets) == 0:
            logging.warning("No valid triplets mined!")
            return 0.0
        logging.info(f"Mined {len(triplets):,} triplets in {time.perf_counter() - t0:.2f}s")
        graph = graph.to(self.device)                                                                 # :180
        triplets = np.array(triplets)
        np.random.shuffle(triplets)                                                                   # :183-184
        avg_loss = self.train_batches(graph, triplets)
        self.train_losses.append(avg_loss)                                                            # :234
        return avg_loss

    def train(self, train_graph, train_poses: np.ndarray, train_descriptors: np.ndarray,
              train_sequence_ids: np.ndarray = None, val_graph=None, val_poses: Optional[np.ndarray] = None,
              n_epochs: int = 50, triplet_miner=None):
        """trainer.py:389-476: epochs of train_epoch + validate, best / periodic / final checkpoints,
        early stopping on recall@1."""
        if triplet_miner is None:
            from .triplet_miner import create_triplet_miner
            triplet_miner = create_triplet_miner()
        for epoch in range(n_epochs):
            self.epoch = epoch
            t0 = time.perf_counter()
            avg_loss = self.train_epoch(train_graph, triplet_miner, train_poses, train_descriptors,
                                        sequence_ids=train_sequence_ids)
            if val_graph is not None and val_poses is not None:
                metrics = self.validate(val_graph, val_poses)
                self.val_metrics.append(metrics)
                logging.info(f"Epoch {epoch + 1}/{n_epochs} | Loss: {avg_loss:.4f} | R@1: {metrics['recall@1']:.4f} | "
                             f"Time: {time.perf_counter() - t0:.1f}s")
                if metrics['recall@1'] > self.best_val_metric:                                        # :446-455
                    self.best_val_metric = metrics['recall@1']
                    self.save_checkpoint('best_model.pth')
                    self.epochs_without_improvement = 0
                else:
                    self.epochs_without_improvement += 1
                if self.epochs_without_improvement >= self.patience:                                  # :457-461
                    logging.info(f"Early stopping after {self.patience} epochs without improvement")
                    break
            else:
                logging.info(f"Epoch {epoch + 1}/{n_epochs} | Loss: {avg_loss:.4f} | "
                             f"Time: {time.perf_counter() - t0:.1f}s")
            if (epoch + 1) % 10 == 0:                                                                 # :467-468
                self.save_checkpoint(f'checkpoint_epoch_{epoch + 1}.pth')
        self.save_checkpoint('final_model.pth')                                                       # :472

    def save_checkpoint(self, filename: str):
        """trainer.py:478-495: same keys, so the files are interchangeable with the reference's."""
        self.checkpoint_dir.mkdir(parents=True, exist_ok=True)
        torch.save({
            'epoch': self.epoch,
            'global_step': self.global_step,
            'model_state_dict': self.model.state_dict(),
            'optimizer_state_dict': self.optimizer.state_dict(),
            'best_val_metric': self.best_val_metric,
            'train_losses': self.train_losses,
            'val_metrics': self.val_metrics,
            'epochs_without_improvement': self.epochs_without_improvement,
        }, self.checkpoint_dir / filename)

    def load_checkpoint(self, filename: str):
        """trainer.py:497-518."""
        load_path = self.checkpoint_dir / filename
        if not load_path.exists():
            raise FileNotFoundError(f"Checkpoint not found: {load_path}")
        ck = torch.load(load_path, map_location=self.device, weights_only=False)
        self.model.load_state_dict(ck['model_state_dict'])
        self.optimizer.load_state_dict(ck['optimizer_state_dict'])
        self.epoch = ck['epoch']
        self.global_step = ck['global_step']
        self.best_val_metric = ck['best_val_metric']
        self.train_losses = ck.get('train_losses', [])
        self.val_metrics = ck.get('val_metrics', [])
        self.epochs_without_improvement = ck.get('epochs_without_improvement', 0)

    # -- the per-batch step, captured --------------------------------------------------------------------------
    def _eager_step(self, graph, ia, ip, in_, scale):
        inner = getattr(self.model, "gnn", self.model)
        # the backward adds straight into the .grad tensors (they exist: zero_grad keeps them) -- no AccumulateGrad pass
        direct = (self.direct_grads and hasattr(inner, "_direct_grads")
                  and all(p.grad is not None for p in self.model.parameters() if p.requires_grad))
        if direct and os.environ.get("NSC_TRAINER_NO_DIRECT") != "1" and hasattr(inner, "train_step_direct") and self.model.training and all(
                isinstance(i, torch.Tensor) and i.is_cuda and i.dtype == torch.int64 for i in (ia, ip, in_)):
            # device-resident indices (the captured step's static buffers, an epoch's uploaded triplets): the three ABI calls
            # back to back, no autograd graph (round 4)
            return inner.train_step_direct(graph, ia.contiguous(), ip.contiguous(), in_.contiguous(),
                                           self.criterion.margin, scale).detach()
        if direct:
            inner._direct_grads = True
        try:
            embeddings = self.model(graph)                                                    # :205
            loss = self.criterion.forward_indexed(embeddings, ia, ip, in_, scale=scale)       # :207-212
            loss.backward()                                                                   # :213
        finally:
            if direct:
                inner._direct_grads = False
        return loss.detach()

    def _captured_step(self, graph, bt, scale, bt_dev=None):
        """forward + TripletLoss + backward of one batch as ONE hipGraph launch.  The step is ~75 small kernels
        (1.2-1.8 ms per batch issued one by one, launch-bound; 0.73-1.04 ms replayed, round 3): everything it touches
        is static -- the replicated graph, the parameter and gradient storage, the workspace of the capture's own
        memory pool -- except three things fed through device buffers the capture reads: the triplet indices, and the
        dropout seed (NscGatTrainCfg.seed_dev).  Gradients ACCUMULATE into the existing .grad tensors (the capture
        holds their addresses: zero_grad must keep them, set_to_none=False).  Returns the loss, or None when capture is
        not possible (then the caller runs the step eagerly)."""
        if not self._graph_enabled() or not torch.cuda.is_available() or torch.device(self.device).type != "cuda":
            return None
        inner = getattr(self.model, "gnn", self.model)
        params = [p for p in self.model.parameters() if p.requires_grad]
        T = int(len(bt))
        # only FULL batches are captured: the ragged last batch of an epoch has a different triplet count every epoch
        # (and per rank), and each capture pins a hipGraph plus a private pool with the whole training workspace
        if self._full_T is not None and T != self._full_T:
            return None
        ea = getattr(graph, "edge_attr", None)
        use_edge = ea is not None and getattr(inner, "edge_dim", None) is not None
        csr = inner._csr(graph, use_edge) if hasattr(inner, "_csr") else None     # the capture bakes its arrays in
        key = (id(graph), graph.x.data_ptr(), graph.edge_index.data_ptr(), None if ea is None else ea.data_ptr(), id(csr),
               T, float(scale), tuple(p.data_ptr() for p in params))
        ent = self._captured.get(key)
        dev = graph.x.device
        if ent is None:
            if self._capture_failed:
                return None
            if key not in self._seen_once:
                if len(self._seen_once) > 64:
                    self._seen_once.clear()
                self._seen_once.add(key)                    # first batch of this shape runs eagerly (lazy set-up, warm-up)
                return None
            try:
                idx_all = torch.empty((3, T), dtype=torch.int64, device=dev)      # one static buffer: one copy per batch feeds it
                idx = [idx_all[0], idx_all[1], idx_all[2]]
                seed = torch.zeros(1, dtype=torch.int64, device=dev)
                self._ensure_flat_grads(params)             # static gradient storage
                for p in params:
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                inner._seed_dev = seed
                torch.cuda.synchronize(dev)
                cg = torch.cuda.CUDAGraph()
                # thread-local error mode: HIP calls of OTHER threads (RCCL's proxy / watchdog) do not invalidate the capture
                with torch.cuda.graph(cg, capture_error_mode="thread_local"):   # records the launches; nothing runs until replay()
                    loss = self._eager_step(graph, idx[0], idx[1], idx[2], scale)
                # the entry keeps what the capture baked in alive (CSR arrays, graph tensors): while it exists their ids
                # and addresses in the key cannot be recycled
                ent = (cg, idx, seed, loss, [p.grad for p in params], (csr, graph, ea), idx_all)
                self._captured[key] = ent
                while len(self._captured) > self.max_captures:
                    self._captured.popitem(last=False)      # least recently used capture: graph + pool are dropped
            except Exception as ex:  # noqa: BLE001 -- any capture problem: fall back to issuing the step eagerly
                logging.warning(f"hipGraph capture of the training step failed ({type(ex).__name__}: {ex}); running eagerly")
                self._capture_failed = True
                return None
            finally:
                inner._seed_dev = None
        self._captured.move_to_end(key)
        cg, idx, seed, loss, grads = ent[:5]
        for p, g in zip(params, grads):
            if p.grad is not g:                             # someone dropped / replaced the static gradient tensor
                if p.grad is not None:
                    g.copy_(p.grad)
                else:
                    g.zero_()
                p.grad = g
        h = np.ascontiguousarray(np.asarray(bt), dtype=np.int64)
        n = int(graph.x.shape[0])
        if h.size and (h.min() < -n or h.max() >= n):
            raise IndexError(f"triplet index out of range for {n} embeddings")
        # bt_dev: the batch's (3, T) index rows already on the device (train_batches uploads an epoch's triplets once: a
        # per-batch copy from pageable host memory would block the host until the stream has drained); ONE copy for the three rows
        ent[6].copy_(bt_dev if bt_dev is not None else torch.from_numpy(np.ascontiguousarray(h.T)), non_blocking=True)
        seed.fill_(int(torch.randint(0, 2 ** 62, (1,)).item()) if float(getattr(inner, "dropout", 0.0)) > 0 else 0)
        cg.replay()                                         # (num_batches_tracked is incremented inside the capture)
        return loss.detach().clone()

    def _graph_enabled(self) -> bool:
        if self.use_graph is not None:
            return bool(self.use_graph)
        import torch.distributed as dist
        return not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)

    def release(self):
        """Drop the captured steps (hipGraphs + their workspace pools)."""
        self._captured.clear()
        self._seen_once.clear()

    def _ensure_flat_grads(self, params) -> None:
        """Gradient storage as views of ONE flat buffer (256-byte aligned slices): zeroing the gradients after an optimizer
        step is one fill instead of one per parameter (25 launches of 2.3 us per step in the round-3 trace).  Existing
        gradients are carried over; anything unusual (mixed devices / dtypes) keeps per-tensor storage."""
        views = getattr(self, "_flat_views", None)
        if views is not None and len(views) == len(params) and all(p.grad is v for p, v in zip(params, views)):
            return
        if not params or any(p.device != params[0].device or p.dtype != torch.float32 or not p.is_cuda for p in params):
            self._flat_grad, self._flat_views = None, None
            return
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 63) // 64 * 64
        flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
        views = []
        for p, o in zip(params, offs):
            v = flat[o:o + p.numel()].view_as(p)
            if p.grad is not None:
                v.copy_(p.grad)
            p.grad = v
            views.append(v)
        self._flat_grad, self._flat_views = flat, views

    def _zero_grads(self, params) -> None:
        views = getattr(self, "_flat_views", None)
        if views is not None and len(views) == len(params) and all(p.grad is v for p, v in zip(params, views)):
            self._flat_grad.zero_()
        else:
            self.optimizer.zero_grad(set_to_none=False)

    def train_batches(self, graph, triplets: Sequence) -> float:
        """trainer.py:186-231 for an (n,3) array of (anchor, positive, negative) triplets."""
        self.model.train()
        triplets = np.asarray(triplets)
        n_batches = (len(triplets) + self.batch_size - 1) // self.batch_size
        losses = []
        from .. import distributed as nd
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        params = [p for p in self.model.parameters() if p.requires_grad]
        # gradients keep their storage across optimizer steps (a captured step holds their addresses)
        use_graph = self._graph_enabled()
        if use_graph and torch.device(self.device).type == "cuda" and torch.cuda.is_available():
            self._ensure_flat_grads(params)
        self._zero_grads(params)
        trip_dev = None
        self._full_T = None
        if len(triplets):
            flo, fhi = nd.shard_range(min(self.batch_size, len(triplets)), rank, world) if world > 1 else (0, min(self.batch_size, len(triplets)))
            self._full_T = fhi - flo
        if use_graph and len(triplets) and torch.device(self.device).type == "cuda" and torch.cuda.is_available():
            trip_dev = torch.from_numpy(np.ascontiguousarray(triplets.T, dtype=np.int64)).to(graph.x.device)   # (3, n), once
        for b in range(n_batches):
            b0 = b * self.batch_size
            bt = triplets[b0:b0 + self.batch_size]
            lo, hi = 0, len(bt)
            # data parallel over ranks: the graph forward is replicated, each rank takes a slice of the
            # triplet batch; weighting by slice size makes the summed gradients those of the global mean
            weight = 1.0
            if world > 1:
                lo, hi = nd.shard_range(len(bt), rank, world)
                bt, weight = nd.split_triplets(bt, rank, world)
            scale = weight / self.accumulation_steps
            bt_dev = trip_dev[:, b0 + lo:b0 + hi] if trip_dev is not None else None
            loss = self._captured_step(graph, bt, scale, bt_dev) if len(bt) else None
            if loss is None:
                loss = self._eager_step(graph, bt[:, 0], bt[:, 1], bt[:, 2], scale)
            losses.append(loss)
            self.global_step += 1
            if (b + 1) % self.accumulation_steps == 0 or (b + 1) == n_batches:                # :219-221
                nd.all_reduce_gradients(params)                            # one 2.46 MB RCCL all-reduce
                self.optimizer.step()
                self._zero_grads(params)
                # the eval-mode forward caches folded attention vectors keyed on the parameters' version counters; an
                # optimizer that writes through a multi-tensor kernel need not bump them (torch's fused Adam does not:
                # measured, round 3) -- drop the cache explicitly
                inner_ = getattr(self.model, "gnn", self.model)
                if hasattr(inner_, "_struct_cache"):
                    inner_._struct_cache = None
        if not losses:
            return 0.0
        return float(torch.stack(losses).mean().item() * self.accumulation_steps)


def create_trainer(model: Optional[nn.Module] = None, device: str = 'cuda', **kwargs) -> GNNTrainer:
    """trainer.py:521-541"""
    if model is None:
        from .model import create_spectral_gnn
        model = create_spectral_gnn()
    return GNNTrainer(model=model, device=device, **kwargs)
