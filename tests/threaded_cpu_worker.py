"""Child process of tests/test_distributed_cpu.py::test_world8_kitti00_layout: BASELINE configs[3]'s layout -- 8
contiguous shards of the 4 541-keyframe set (568 / 567 rows: ragged, so the padded all-gather and the halo window at
7 interior boundaries) -- with the product's ShardedDescriptorPath / ShardedTwoStageRetrieval on every rank and the
oracle standing in for the kernels.  The 8 ranks are threads of this process over torch's in-process "threaded"
process group (the same harness the GPU test uses, tests/multirank_worker.py --threads)."""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch
import torch.distributed as dist


def main():
    world, n_total, pipeline, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == "pipelined", sys.argv[4]
    from torch.testing._internal.distributed import multi_threaded_pg as mtp
    from neural_spectral_codec_amd import distributed as nd
    from neural_spectral_codec_amd import synth
    from neural_spectral_codec_amd.gnn.model import create_spectral_gnn
    from neural_spectral_codec_amd.retrieval import ShardedTwoStageRetrieval
    import gat_oracle as go
    from test_distributed_cpu import OracleEncoder, OracleGnn, OracleLocalRetriever

    torch.set_num_threads(1)
    mtp._install_threaded_pg()
    torch._C._distributed_c10d._set_thread_isolation_mode(True)
    store = dist.HashStore()
    torch.manual_seed(0)
    model = create_spectral_gnn(edge_dim=2).eval()
    go.randomize_bn_stats(model)
    poses = synth.make_pose_chain(n_total, 3)
    results, errors = {}, []

    def worker(rank):
        try:
            dist.init_process_group(backend="threaded", rank=rank, world_size=world, store=store)
            path = nd.ShardedDescriptorPath(OracleEncoder(), OracleGnn(model), n_total, poses, pipeline=pipeline)
            lo, hi = path.lo, path.hi
            clouds = [synth.make_cloud(1000 + i, 400, "uniform") for i in range(lo, hi)]
            other = [synth.make_cloud(5000 + i, 300, "ring") for i in range(lo, hi)]
            n_steps = nd.ShardedDescriptorPath._PIPE_BUFFERS + 1 if pipeline else 3     # odd: the last encodes `clouds`
            for k in range(n_steps):
                desc_all, emb = path.step(clouds if k % 2 == 0 else other)
            path.synchronize()
            sh = ShardedTwoStageRetrieval(OracleLocalRetriever(), n_total, top_k=10, spatial_filter_distance=8.0)
            pos = poses[:, :3, 3].astype(np.float32)
            sh.add_local_rows(desc_all[lo:hi].numpy(), pos[lo:hi])
            qsel = [0, n_total // 3, n_total // 2, n_total - 1]
            idx, val = sh.query_batch(desc_all[qsel].numpy(), pos[qsel])
            results[rank] = dict(lo=lo, hi=hi, overlap=int(path.overlap), desc_all=desc_all.numpy().copy(),
                                 emb=emb.numpy().copy(), retr_idx=idx.numpy(), retr_val=val.numpy())
            dist.barrier()
        except BaseException as ex:  # noqa: B036
            import traceback
            errors.append((rank, traceback.format_exc()))
            mtp.ProcessLocalGroup.exception_handle(ex)
        finally:
            try:
                dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass

    ths = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errors:
        for rank, tb in errors:
            print(f"rank {rank} failed:\n{tb}", flush=True)
        sys.exit(1)
    flat = {}
    for r, d in results.items():
        for k, v in d.items():
            flat[f"r{r}_{k}"] = v
    np.savez(out, **flat)


if __name__ == "__main__":
    main()
