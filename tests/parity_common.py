"""Shared differential checks: an engine (the HIP library on the GPU box; in host-logic tests the test double of the C
ABI) against the C oracle (oracle/wn_oracle.c).

Tolerances:
  * logits, teacher-forced on the oracle's index sequence:  max|d| <= 1e-5 * max(1, |logits|_inf)
    (the reference's own fp32-vs-fp64 noise floor is ~2e-7 .. 7e-7 at these scales)
  * greedy indices: bit-exact, asserted only when the oracle's smallest top-2 logit gap exceeds 10x the
    logit tolerance (otherwise an argmax flip would be legitimate rounding, and the test says so)
  * sampled indices: identical for the same uniforms; a mismatch is accepted only if the uniform sits within
    1e-6 of a CDF boundary of the oracle at the first diverging step (never observed; reported if it happens)
"""
import numpy as np

import c_oracle
from mi355_wavenet import engine, synth

LOGIT_RTOL = 1e-5


def oracle_run(cfg, W, N, first, temperature, regularize, uniforms, forced=None):
    return c_oracle.generate(cfg, W, N, first, temperature, regularize, uniforms, forced=forced)


def softmax_cdf(logits_row, temperature, reg):
    x = logits_row.astype(np.float32) - (reg if reg is not None else 0)
    x = (x / np.float32(temperature)).astype(np.float32)
    p = np.exp(x - x.max()).astype(np.float32)
    p = (p * (np.float32(1) / p.sum(dtype=np.float32))).astype(np.float64)
    cdf = np.cumsum(p)
    return cdf / cdf[-1]


def check_engine(eng, cfg, W, N, first, temperature=0.0, regularize=0.0, uniforms=None, label=""):
    """first: (ns, n_given); uniforms: (ns, N) or None.  Returns dict of measured deviations."""
    ns = eng.n_streams
    first = np.asarray(first)
    if first.ndim == 1:
        first = np.broadcast_to(first[None], (ns, first.shape[0]))
    idx, logits = eng.generate(N, first, temperature=temperature, regularize=regularize, uniforms=uniforms,
                               want_logits=True, timeout_ms=4000)
    assert idx.shape == (ns, N) and logits.shape == (ns, N, cfg.get("classes", 256))
    worst = 0.0
    min_gap = np.inf
    reg = c_oracle.regularizer_array(cfg.get("classes", 256), regularize) if regularize else None
    for s in range(ns):
        u = uniforms[s] if (uniforms is not None and temperature > 0) else None
        o_idx, o_log = oracle_run(cfg, W, N, first[s], temperature, regularize, u)
        # logits: free-running while the sequences agree, teacher-forced on the engine's sequence after
        same = np.array_equal(idx[s], o_idx)
        if not same:
            _, o_log_f = oracle_run(cfg, W, N, first[s], temperature, regularize, u, forced=idx[s])
        else:
            o_log_f = o_log
        tol = LOGIT_RTOL * max(1.0, float(np.abs(o_log_f).max()))
        dev = float(np.abs(logits[s] - o_log_f).max())
        worst = max(worst, dev)
        assert dev <= tol, "%s stream %d: logits deviate %.3g > %.3g" % (label, s, dev, tol)
        if u is None:  # greedy: bit-exact when the gap allows the claim
            xs = o_log - (reg if reg is not None else 0)
            top2 = np.sort(xs, axis=1)
            gap = float((top2[:, -1] - top2[:, -2]).min())
            min_gap = min(min_gap, gap)
            if gap > 10 * tol:
                assert same, "%s stream %d: greedy indices differ at step %d (gap %.3g)" % (
                    label, s, int(np.argmax(idx[s] != o_idx)), gap)
            elif not same:
                first_bad = int(np.argmax(idx[s] != o_idx))
                row = np.sort(xs[first_bad])
                assert row[-1] - row[-2] <= 10 * tol, "%s: greedy flip at a non-degenerate step" % label
        else:
            if not same:
                t = int(np.argmax(idx[s] != o_idx))
                cdf = softmax_cdf(o_log[t], temperature, reg)
                margin = float(np.abs(cdf - u[t]).min())
                assert margin < 1e-6, "%s stream %d: sampled indices differ at step %d, CDF margin %.3g" % (label, s, t, margin)
        assert np.array_equal(c_oracle.expand(idx[s]), c_oracle.expand(idx[s].astype(np.int64)))
    return {"max_logit_dev": worst, "min_gap": min_gap}


def make_case(cfg_name_or_dict, seed, ns, n_given, N):
    cfg = synth.CONFIGS[cfg_name_or_dict] if isinstance(cfg_name_or_dict, str) else cfg_name_or_dict
    W = synth.init_weights(cfg, seed=seed)
    rs = np.random.RandomState(seed + 1000)
    first = rs.randint(0, cfg.get("classes", 256), (ns, n_given))
    uniforms = rs.random_sample((ns, N))
    return cfg, W, first, uniforms
