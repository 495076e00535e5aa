"""dev tool: does the chain's per-item cycle depend on the dilations?  Same widths as cfg3 (128/128/512/256), same number of
layers (50), different dilation patterns: layers x blocks = 1 x 50 (all d = 1), 2 x 25 (d <= 2), 5 x 10 (d <= 16), 10 x 5 (cfg3).

    python tools/dilation_probe.py [n_streams=64]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-wavenet_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mi355_wavenet import engine, synth  # noqa: E402


def main():
    ns = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n = 1500
    for layers, blocks in ((1, 50), (2, 25), (5, 10), (7, 7), (8, 6), (9, 5), (10, 5)):
        cfg = dict(synth.CONFIGS["cfg3"], layers=layers, blocks=blocks)
        W = synth.init_weights(cfg, seed=0)
        eng = engine.Engine(cfg, W, n_streams=ns)
        first = eng.mem.upload(np.full((ns, 1), 128, dtype=np.int32))
        uni = eng.mem.upload(np.random.RandomState(1).random_sample((ns, n)))
        out = eng.mem.empty((ns, n), np.int32)
        best = 0.0
        for _ in range(3):
            eng.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.launch(first, 1, n, 1.0, None, uni, out, None, timeout_ms=20000)
            eng.wait()
            best = max(best, ns * n / (time.perf_counter() - t0))
        info = eng.info()
        nl = layers * blocks
        print("layers %2d x blocks %2d (%d layers, max dilation %3d) x%d: %.0f samples/s, %.2f us per timestep, %.3f us per item and stage (variant %d, %d workgroups)" % (
            layers, blocks, nl, 2 ** (layers - 1), ns, best, ns * 1e6 / best, 1e6 / best, info["kernel_variant"], info["n_workgroups"]))
        eng.close()


if __name__ == "__main__":
    main()
