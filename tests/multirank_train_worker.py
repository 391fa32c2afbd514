"""Child process of tests/test_a_multirank_gpu.py: ONE rank of the data-parallel training step (BASELINE configs[4]'s
"8 x MI355X" leg, reference src/gnn/trainer.py:186-221 with the triplet batch split over the ranks) through the HIP
kernels.  Ranks share cuda:0 and exchange over gloo.  Every rank builds the same 4 541-keyframe dataset, mines the
same triplets, runs GNNTrainer.train_batches on ONE 1 024-triplet batch -- replicated graph forward, its slice of
the batch, nsc_gat_backward, all_reduce_gradients, Adam -- and writes the reduced gradients and the updated parameters."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--use-graph", type=int, default=-1)          # -1: the trainer's default (off under a > 1-rank group)
    ap.add_argument("--batches", type=int, default=1)             # full batches of --batch-size triplets (+ a ragged tail of 37)
    ap.add_argument("--batch-size", type=int, default=1024)
    a = ap.parse_args()
    if a.world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(a.port)
        dist.init_process_group("gloo", rank=a.rank, world_size=a.world)

    from neural_spectral_codec_amd import synth
    from neural_spectral_codec_amd.gnn.model import create_spectral_gnn
    from neural_spectral_codec_amd.gnn.trainer import GNNTrainer
    from neural_spectral_codec_amd.gnn.triplet_miner import create_triplet_miner
    from neural_spectral_codec_amd.keyframe.graph_manager import build_chain_graph
    from test_train_gpu import _config5_dataset, _key_map

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    desc, poses, seq_ids = _config5_dataset(dev)
    graph = build_chain_graph(desc, 5, "cuda", poses)
    np.random.seed(3)
    trip_all = np.asarray(create_triplet_miner().mine_triplets(desc.cpu().numpy(), poses, 1, sequence_ids=seq_ids))
    np.random.shuffle(trip_all)
    trip = trip_all[:1024] if a.batches == 1 else trip_all[:a.batches * a.batch_size + 37]
    torch.manual_seed(0)
    m = create_spectral_gnn(input_dim=800, hidden_dim=256, output_dim=800, n_layers=3, dropout=0.0, edge_dim=2)
    synth.randomize_bn_stats(m, 1)
    with torch.no_grad():
        for c in m.gnn.convs:
            c.bias.normal_(0, 0.1)
    m = m.to(dev)
    tr = GNNTrainer(m, device="cuda", learning_rate=5e-4, weight_decay=1e-5, margin=0.1, batch_size=a.batch_size,
                    accumulation_steps=4, use_graph=None if a.use_graph < 0 else bool(a.use_graph))
    grads = {}
    params = dict(m.gnn.named_parameters())
    step = tr.optimizer.step

    def snapshot_then_step(*args, **kw):                      # the gradients Adam sees = after all_reduce_gradients
        for k in _key_map(m.gnn):
            grads[k] = params[k].grad.detach().cpu().numpy().copy()
        if os.environ.get("NSC_DP_DEBUG") == "1":
            print("STEP grads", [(k, float(np.abs(grads[k]).sum())) for k in ("input_proj.weight", "convs.1.lin_src.weight", "output_proj.weight", "batch_norms.0.weight")],
                  "params", [float(params[k].detach().abs().sum()) for k in ("input_proj.weight", "output_proj.weight")], flush=True)
        return step(*args, **kw)
    tr.optimizer.step = snapshot_then_step
    if os.environ.get("NSC_DP_DEBUG") == "1":                  # per-batch losses (tools/dp_debug.sh)
        rec = []
        cs, es = tr._captured_step, tr._eager_step

        def cs2(*aa, **kw):
            r = cs(*aa, **kw)
            if r is not None:
                torch.cuda.synchronize()
                rec.append(("c", float(r)))
            return r

        def es2(*aa, **kw):
            r = es(*aa, **kw)
            if not torch.cuda.is_current_stream_capturing():
                torch.cuda.synchronize()
                rec.append(("e", float(r)))
            return r
        tr._captured_step, tr._eager_step = cs2, es2
    loss = tr.train_batches(graph, trip)
    if os.environ.get("NSC_DP_DEBUG") == "1":
        print("LOSSES", [(k, round(v, 6)) for k, v in rec], flush=True)
    torch.cuda.synchronize(dev)
    out = {"loss": np.float64(loss), "trip": trip, "n_local": np.int64(len(trip) if a.world == 1 else
                                                                         len(np.array_split(trip, a.world)[a.rank]))}
    for k in _key_map(m.gnn):
        out["g:" + k] = grads[k]
        out["p:" + k] = params[k].detach().cpu().numpy()
    out["bn_mean"] = m.gnn.input_norm.running_mean.cpu().numpy()
    out["captured"] = np.int64(len(tr._captured))
    out["capture_failed"] = np.int64(bool(tr._capture_failed))
    np.savez(a.out, **out)
    if a.world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
