"""Experiment: K independent multi-stream chains (one persistent kernel each, on separate HIP streams) sharing the chip,
versus one chain with all the streams.  python tools/two_chains.py [total_streams] [chains] [samples]
WN_DEV_LIB=/path/to/lib.so selects a development build of the library."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-wavenet_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mi355_wavenet import _abi, engine, synth  # noqa: E402


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    chains = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    samples = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    lib = _abi.Library(os.environ["WN_DEV_LIB"]) if os.environ.get("WN_DEV_LIB") else None
    cfg = synth.CONFIGS["cfg3"]
    W = synth.init_weights(cfg, seed=0)
    per = total // chains
    engs = [engine.Engine(cfg, W, n_streams=per, lib=lib) for _ in range(chains)]
    streams = [torch.cuda.Stream() for _ in range(chains)]
    rs = np.random.RandomState(0)
    bufs = []
    for e in engs:
        bufs.append((e.mem.upload(np.full((per, 1), 128, dtype=np.int32)), e.mem.upload(rs.random_sample((per, samples))),
                     e.mem.empty((per, samples), np.int32)))
    print("chains %d x %d streams; per chain: %s" % (chains, per, {k: engs[0].info()[k] for k in ("n_workgroups", "lds_bytes", "layer_split", "head_split")}))

    def run():
        for e, st, (first, uni, out) in zip(engs, streams, bufs):
            with torch.cuda.stream(st):
                e.reset()
                e.launch(first, 1, samples, 1.0, None, uni, out, None, timeout_ms=20000)
        for e in engs:
            e.wait()
        torch.cuda.synchronize()

    run()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        run()
        best = min(best, time.perf_counter() - t0)
    print("total %d streams x %d samples: %.1f ms -> %.0f samples/s" % (total, samples, best * 1e3, total * samples / best))
    outs = [b[2].cpu().numpy() for b in bufs]
    print("checksum", int(sum(int(o.sum()) for o in outs)))


if __name__ == "__main__":
    main()
