"""dev tool: un-instrumented throughput of one engine job (no stamps): samples/s for a config / stream count.

    [WN_DEV_LIB=...] python tools/rate.py cfg3 64 [samples=2000] [reps=3]
"""
import os
import sys

os.environ.setdefault("WN_TESTING", "1")  # dev tool: WN_V3_MODE / WN_KERNEL pins are honoured
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-wavenet_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mi355_wavenet import _abi, engine, synth  # noqa: E402

if os.environ.get("WN_DEV_LIB"):
    _abi.PRODUCT_LIB = os.path.abspath(os.environ["WN_DEV_LIB"])


def main():
    cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    cfg = synth.CONFIGS[cfgname]
    W = synth.init_weights(cfg, seed=0)
    eng = engine.Engine(cfg, W, n_streams=ns)
    first = eng.mem.upload(np.full((ns, 1), 128, dtype=np.int32))
    uni = eng.mem.upload(np.random.RandomState(1).random_sample((ns, n)))
    out = eng.mem.empty((ns, n), np.int32)
    best = 0.0
    for _ in range(reps + 1):
        eng.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.launch(first, 1, n, 1.0, None, uni, out, None, timeout_ms=20000)
        eng.wait()
        dt = time.perf_counter() - t0
        best = max(best, ns * n / dt)
    info = eng.info()
    print("%s x%d: %.0f samples/s (%.2f us per timestep, %.0f per stream; variant %d, %d chain(s), %d workgroups)" % (
        cfgname, ns, best, 1e6 * ns / best, best / ns, info["kernel_variant"], info["n_chains"], info["n_workgroups"]))
    eng.close()


if __name__ == "__main__":
    main()
