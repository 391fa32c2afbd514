"""Child process of tests/test_a_multirank_gpu.py: ranks of the sharded descriptor path on a HIP device.

All ranks share cuda:0 (the GPU boxes of the test pool have one card).  Two transports:

* ``--rank R --world W``: ONE rank per process, exchange over gloo -- the layout of ``NSC_BENCH_REHEARSAL``;
* ``--threads --world W``: ALL W ranks as threads of this one process, exchange through torch's in-process
  "threaded" process group.  The GPU boxes allow at most 6 processes on the card at once, so BASELINE configs[3]'s
  8-way shard cannot be rehearsed with a process per rank; the threads each run the product's
  ``ShardedDescriptorPath`` / ``ShardedTwoStageRetrieval`` code unchanged, with their own rank, streams and buffers.

The kernels, streams and buffer rotation are the product's; only the transport differs from the 8-GPU node (RCCL).
Every rank writes desc_all / emb of the step computed from the reference batch to an .npz.
"""
import argparse
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch
import torch.distributed as dist


_INIT_LOCK = threading.Lock()


def run_rank(a, out_path):
    """One rank (process or thread); torch.distributed is already initialised for world > 1."""
    from neural_spectral_codec_amd import distributed as nd
    from neural_spectral_codec_amd import synth
    from neural_spectral_codec_amd.encoding import SpectralEncoder
    from neural_spectral_codec_amd.gnn.model import create_spectral_gnn

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    enc = SpectralEncoder(n_elevation=16, n_azimuth=360, n_bins=50, alpha=2.0, target_elevation_bins=16).to(dev)
    with _INIT_LOCK:                                          # threads share torch's global CPU generator
        torch.manual_seed(0)
        model = create_spectral_gnn(edge_dim=2)
        synth.randomize_bn_stats(model)
    model = model.to(dev).eval()
    poses = synth.make_pose_chain(a.n_total, 3)
    pipelined = a.mode == "pipelined"
    path = nd.ShardedDescriptorPath(enc, model, a.n_total, poses, pipeline=pipelined)
    lo, hi = path.lo, path.hi

    def batch(seed0, pts, kind):
        p, o = synth.make_clouds_packed(range(seed0 + lo, seed0 + hi), pts, kind)
        return torch.from_numpy(p).to(dev), torch.from_numpy(o).to(dev)

    ref_batch = batch(1000, a.points, "uniform")
    other = batch(7000, a.points // 2, "ring")
    torch.cuda.synchronize(dev)
    n_steps = nd.ShardedDescriptorPath._PIPE_BUFFERS + 3      # odd: the last step encodes ref_batch
    kept = None
    with torch.no_grad():
        for k in range(n_steps):
            res = path.step(ref_batch if k % 2 == 0 else other, inputs_ready=True)
            if k == n_steps - 3:
                kept = res            # pipelined: two steps old, still valid (4 buffers in rotation)
        path.synchronize()
        torch.cuda.synchronize(dev)
    desc_all, emb = res
    # the consumer of the gathered matrix: stage-1 retrieval with the database rows sharded over the ranks
    # (two_stage_retrieval.py:145-202) -- every rank scores its own rows, ONE all-gather of k candidates, merge
    from neural_spectral_codec_amd.retrieval import ShardedTwoStageRetrieval, WassersteinRetriever
    sh = ShardedTwoStageRetrieval(WassersteinRetriever(device=dev), a.n_total, top_k=10, spatial_filter_distance=8.0)
    pos = torch.from_numpy(poses[:, :3, 3].astype(np.float32)).to(dev)
    sh.add_local_rows(desc_all[lo:hi], pos[lo:hi])
    qsel = torch.tensor([0, a.n_total // 3, a.n_total // 2, a.n_total - 1], device=dev)
    r_idx, r_val = sh.query_batch(desc_all[qsel], pos[qsel])
    torch.cuda.synchronize(dev)
    out = {"lo": lo, "hi": hi, "desc_all": desc_all.cpu().numpy(), "emb": emb.cpu().numpy(),
           "retr_idx": r_idx.cpu().numpy(), "retr_val": r_val.cpu().numpy(),
           "coresident": int(bool(getattr(getattr(model, "gnn", model), "coresident", False))),
           "overlap": int(bool(path.overlap))}
    if pipelined:
        out["desc_all_kept"] = kept[0].cpu().numpy()
        out["emb_kept"] = kept[1].cpu().numpy()
    np.savez(out_path, **out)


def _gpu_safe_collectives():
    """The in-process group copies between the ranks' tensors on ONE thread's current stream (rank 0's) once every
    rank has arrived; the ranks' own streams know nothing of that copy.  Bracket every collective with a device
    synchronise: what a rank hands in is complete, and what it takes out has landed, whatever stream did the copy."""
    def wrap(fn):
        def inner(*args, **kw):
            torch.cuda.synchronize()
            r = fn(*args, **kw)
            torch.cuda.synchronize()
            return r
        return inner
    for name in ("all_gather_into_tensor", "all_gather", "all_reduce", "broadcast", "barrier"):
        setattr(dist, name, wrap(getattr(dist, name)))


def run_threads(a):
    from torch.testing._internal.distributed import multi_threaded_pg as mtp
    from neural_spectral_codec_amd import distributed as nd
    nd.ShardedDescriptorPath.probe_queues = False          # W ranks probing at once would only time each other
    torch.cuda.init()
    mtp._install_threaded_pg()
    torch._C._distributed_c10d._set_thread_isolation_mode(True)
    _gpu_safe_collectives()
    store = dist.HashStore()
    errors = []

    def worker(rank):
        try:
            dist.init_process_group(backend="threaded", rank=rank, world_size=a.world, store=store)
            run_rank(a, a.out.replace("{rank}", str(rank)))
            dist.barrier()
        except BaseException as ex:  # noqa: B036 -- reported by the parent through the exit status
            import traceback
            errors.append((rank, traceback.format_exc()))
            mtp.ProcessLocalGroup.exception_handle(ex)     # wake the ranks waiting in a collective
        finally:
            try:
                dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass

    ths = [threading.Thread(target=worker, args=(r,)) for r in range(a.world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errors:
        for rank, tb in errors:
            print(f"rank {rank} failed:\n{tb}", flush=True)
        sys.exit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, default=0)
    ap.add_argument("--threads", action="store_true")
    ap.add_argument("--mode", choices=["serial", "pipelined"], required=True)
    ap.add_argument("--n-total", type=int, required=True)
    ap.add_argument("--points", type=int, default=6000)
    ap.add_argument("--out", required=True)              # --threads: a pattern containing {rank}
    a = ap.parse_args()

    if a.threads:
        run_threads(a)
        return
    if a.world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(a.port)
        dist.init_process_group("gloo", rank=a.rank, world_size=a.world)
    run_rank(a, a.out)
    if a.world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
