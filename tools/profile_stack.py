"""Anatomy of the stacked generation chain (kernel variant 4, csrc/wn_kernel_v4.h) from in-kernel wall-clock stamps.

    python tools/profile_stack.py [cfg2] [n_streams]

Per stack workgroup: wait for its input, time in its layers (and per layer), skip chunk, rest of the tail; per hop: x' published ->
the consumer has staged it; the ring's tail: last skip lane published -> head staged -> logits published -> sampler has them ->
start_conv row published -> stack workgroup 0 staged.  100 MHz stamps (10 ns).
"""
import os
import sys

os.environ.setdefault("WN_TESTING", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-wavenet_amd"))
import numpy as np  # noqa: E402

from mi355_wavenet import _abi, engine, synth  # noqa: E402

if os.environ.get("WN_DEV_LIB"):
    _abi.PRODUCT_LIB = os.path.abspath(os.environ["WN_DEV_LIB"])


def main():
    cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cfg = synth.CONFIGS[cfgname]
    W = synth.init_weights(cfg, seed=0)
    eng = engine.Engine(cfg, W, n_streams=ns)
    info = eng.info()
    assert info["kernel_variant"] == 4, info
    NL, LPW, PA, n_smp = info["n_layers"], info["layers_per_workgroup"], info["head_split"], info["n_samplers"]
    n_stack = (NL + LPW - 1) // LPW
    N = 400
    u = np.random.RandomState(0).random_sample((ns, N))
    eng.generate(N, None, temperature=1.0, uniforms=u)
    eng.profile_next(N * ns)
    eng.generate(N, None, temperature=1.0, uniforms=u)
    raw = eng.profile_read(N * ns).astype(np.float64) * 0.01   # [workgroup][item][8] us
    eng.close()
    lo, hi = N * ns // 4, N * ns - ns
    st = raw[:n_stack, lo:hi]
    period = np.diff(raw[0, lo:hi:ns, 1]).mean()
    print("%s x%d: variant 4, %d stack workgroups of %d layers, %d head, %d sampler workgroup(s); %.2f us per timestep (%.0f samples/s per stream, %.0f total)" % (
        cfgname, ns, n_stack, LPW, PA, n_smp, period, 1e6 / period, ns * 1e6 / period))
    nl = np.array([min(LPW, NL - w * LPW) for w in range(n_stack)])
    wait = (st[:, :, 1] - st[:, :, 0]).mean(axis=1)
    layers = (st[:, :, 3] - st[:, :, 1]).mean(axis=1)
    first = (st[:, :, 2] - st[:, :, 1]).mean(axis=1)
    skip = (st[:, :, 4] - st[:, :, 3]).mean(axis=1)
    rest = (st[:, :, 5] - st[:, :, 4]).mean(axis=1)
    print("  stack workgroups: layers (staged -> x' published) %.3f us = %.3f per layer (first layer %.3f); skip chunk %.3f; taps + tap-0 dots %.3f; busy %.3f of the %.2f us period" % (
        layers.mean(), (layers / nl).mean(), first.mean(), skip.mean(), rest.mean(), (layers + skip + rest).mean(), period))
    print("    per workgroup layers: %s" % np.array2string(layers, precision=2, max_line_width=200))
    hop = (st[1:, :, 1] - st[:-1, :, 3])
    print("  hop x' published -> consumer staged: mean %.3f us (p10 %.3f p50 %.3f p90 %.3f); by consumer %s" % (
        hop.mean(), np.percentile(hop, 10), np.percentile(hop, 50), np.percentile(hop, 90), np.array2string(hop.mean(axis=1), precision=2, max_line_width=200)))
    # the tail of the ring, per (evaluation, stream)
    seg = []
    for e in range(N // 4, N - 2):
        for s in range(ns):
            it = e * ns + s
            last = raw[n_stack - 1, it]
            hrow = raw[n_stack:n_stack + PA, it]
            smp = raw[n_stack + PA + s % n_smp, it]
            nxt = raw[0, (e + 1) * ns + s]
            seg.append([last[4] - last[3], hrow[:, 1].max() - last[4], hrow[:, 2].max() - hrow[:, 1].max(), smp[1] - hrow[:, 2].max(), smp[2] - smp[1], nxt[1] - smp[2],
                        nxt[1] - last[3]])
    seg = np.array(seg)
    print("  ring tail per token (us): last layers done -> skip lane published %.3f | -> head staged %.3f | -> logits published %.3f | -> sampler has them %.3f | "
          "-> row published %.3f | -> stack 0 staged %.3f || total %.3f" % tuple(seg.mean(axis=0)))
    print("  ring = %d x %.3f (layers) + %d x %.3f (hops) + %.3f (tail) = %.2f us" % (
        NL, (layers / nl).mean(), n_stack - 1, hop.mean(), seg[:, 6].mean(), NL * (layers / nl).mean() + (n_stack - 1) * hop.mean() + seg[:, 6].mean()))


if __name__ == "__main__":
    main()
