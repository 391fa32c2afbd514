"""Child process of tests/test_a_multirank_gpu.py: the N > 1 step of ShardedDescriptorPath on a ONE-rank RCCL process group
(`rehearse_collectives=True`: every collective of the multi-rank step is issued -- the boundary-row gather, the asynchronous
full gather, the pipelined path's gather into its rotating buffers -- with the nccl backend's stream semantics: a collective is
ordered on the stream it is issued on and `wait()` does not block the host, unlike gloo's).  Each form must return what the
exchange-free single-process path returns, bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch
import torch.distributed as dist


def main():
    port = sys.argv[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from neural_spectral_codec_amd import distributed as nd, synth
    from neural_spectral_codec_amd.encoding import SpectralEncoder
    from neural_spectral_codec_amd.gnn.model import create_spectral_gnn
    n = 192
    enc = SpectralEncoder(n_elevation=16).to(dev)
    torch.manual_seed(0)
    m = create_spectral_gnn(edge_dim=2)
    synth.randomize_bn_stats(m)
    m = m.to(dev).eval()
    poses = synth.make_pose_chain(n, 0)
    batches = [synth.make_clouds_device(n, 4000, dev, seed=s) for s in range(6)]
    with torch.no_grad():
        plain = nd.ShardedDescriptorPath(enc, m, n, poses)                          # no process group yet: no exchange
        want = [tuple(t.clone() for t in plain.step(b)) for b in batches]
        dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
        assert dist.get_backend() == "nccl"
        forms = {"serial, two-phase exchange": dict(overlap=True), "serial, one gather": dict(overlap=False),
                 "pipelined": dict(pipeline=True), "pipelined, one encoder stream": dict(pipeline=True, encoder_streams=1)}
        for name, kw in forms.items():
            p = nd.ShardedDescriptorPath(enc, m, n, poses, rehearse_collectives=True, **kw)
            assert p._collect and p.overlap == bool(kw.get("overlap")) and not p.gnn_graph, name
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in batches]
            p.collective_events = evs
            got = []
            for b in batches:
                d, e = p.step(b)
                if p.pipeline:
                    torch.cuda.current_stream().wait_event(p.last_event)
                got.append((d.clone(), e.clone()))
            p.synchronize()
            torch.cuda.synchronize()
            for k, ((wd, we), (gd, ge)) in enumerate(zip(want, got)):
                assert torch.equal(wd, gd), (name, k, "descriptors")
                assert torch.equal(we, ge), (name, k, "embeddings")
            timed = [a.elapsed_time(b) for a, b in evs if a.query() and b.query()]
            assert len(timed) == len(batches) and all(t > 0 for t in timed), (name, timed)
            # fire and forget (what bench.py does): only the last step's results are read
            for b in batches:
                d, e = p.step(b)
            p.synchronize()
            torch.cuda.synchronize()
            assert torch.equal(d, want[-1][0]) and torch.equal(e, want[-1][1]), (name, "fire and forget")
            print(f"{name}: ok ({sum(timed) / len(timed) * 1e3:.0f} us per exchange)", flush=True)
    dist.destroy_process_group()
    print("RCCL_WORLD1_OK", flush=True)


if __name__ == "__main__":
    main()
