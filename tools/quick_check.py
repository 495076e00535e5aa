"""dev tool: a short oracle check of the engine build named by WN_DEV_LIB (default: the product library) on one config.

    [WN_DEV_LIB=tools/variants/libX.so] python tools/quick_check.py cfg3 7 [N]
"""
import os
import sys

os.environ.setdefault("WN_TESTING", "1")  # dev tool: WN_V3_MODE / WN_KERNEL pins are honoured

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pytorch-wavenet_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

from mi355_wavenet import _abi, engine  # noqa: E402

if os.environ.get("WN_DEV_LIB"):
    _abi.PRODUCT_LIB = os.path.abspath(os.environ["WN_DEV_LIB"])
import c_oracle  # noqa: E402
from parity_common import make_case  # noqa: E402


def main():
    cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 120
    cfg, W, first, uniforms = make_case(cfgname, 91, ns, 9, N)
    eng = engine.Engine(cfg, W, n_streams=ns)
    info = eng.info()
    idx, logits = eng.generate(N, first, temperature=1.0, uniforms=uniforms, want_logits=True, timeout_ms=8000, batched_prime=False)
    bad = 0
    worst = 0.0
    for s in sorted(set((0, ns // 2, ns - 1))):
        o_idx, o_log = c_oracle.generate(cfg, W, N, first[s], 1.0, 0.0, uniforms[s])
        if not np.array_equal(idx[s], o_idx):
            bad += 1
        else:
            worst = max(worst, float(np.abs(logits[s] - o_log).max()) / max(1.0, float(np.abs(o_log).max())))
    print("quick_check %s x%d variant %d: %s (worst relative logit deviation %.2e)" % (cfgname, ns, info["kernel_variant"], "OK" if bad == 0 and worst <= 1e-5 else "MISMATCH in %d streams" % bad, worst))
    eng.close()
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
