"""Pins the oracle (oracle/restated.py = torch restatement, oracle/wn_oracle.c = plain C) against
(a) the committed fixtures produced by the real reference (tests/golden/make_golden.py) and
(b) the live reference when /root/reference exists (authoring container only).

Tolerances are stated where used.  fp32 noise floor of the reference itself (fp32 vs fp64 evaluation of
the same weights) is ~2e-7..7e-7 at these logit scales.
"""
import os
import sys

import numpy as np
import pytest
import torch

import c_oracle
import ref_shim
import restated
from mi355_wavenet import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GEN_CASES = {"tiny": "tiny", "tiny_bias": "tiny_bias", "cfg1": "cfg1", "cfg1_seed128": "cfg1",
             "cfg2": "cfg2", "cfg3": "cfg3"}  # cfg2 / cfg3: golden_v2.npz, 640 / 700 given samples (the d=512 queues wrap)
LOGIT_TOL = 1e-5  # max|dlogit| <= 1e-5 * max(1, |logits|_inf)   


# ---------------------------------------------------------------- DilatedQueue / known answers
def test_queue_known_answers_restated(golden):
    # /root/reference/tests/test_tensor_queue.py:13-24 test_enqueue
    q = restated.Queue(3, 8)
    e = torch.zeros(3)
    for _ in range(11):
        e = e + 1
        q.enqueue(e)
    assert q.data[0, 0] == 9 and q.data[0, 2] == 11 and q.data[0, 7] == 8
    assert np.array_equal(q.data.numpy(), golden["queue_enqueue_data"])
    # :26-40 test_dequeue
    q = restated.Queue(1, 8)
    e = torch.zeros(1)
    for _ in range(11):
        e = e + 1
        q.enqueue(e)
    seq = [q.dequeue(3, 2).numpy().copy() for _ in range(9)]
    assert list(seq[-1][0]) == [5, 7, 9]
    assert np.array_equal(np.stack(seq), golden["queue_dequeue_seq"])
    # :42-50 test_combined
    q = restated.Queue(1, 12)
    e = torch.zeros(1)
    seq = []
    for i in range(30):
        e = e + 1
        q.enqueue(e)
        d = q.dequeue(3, 4)
        assert d[0][0] == max(i - 7, 0)
        seq.append(d.numpy().copy())
    assert np.array_equal(np.stack(seq), golden["queue_combined_seq"])


# ---------------------------------------------------------------- generate_fast vs fixtures
def _case(golden, case):
    wseed, n_given, n, npseed = [int(v) for v in golden["gen_%s_meta" % case]]
    temp, regz = [float(v) for v in golden["gen_%s_tr" % case]]
    cfg = synth.CONFIGS[GEN_CASES[case]]
    W = synth.init_weights(cfg, seed=wseed)
    first = golden["gen_%s_first" % case].astype(np.int64)
    return cfg, W, first, n, temp, regz, npseed


@pytest.mark.parametrize("case", sorted(GEN_CASES))
def test_restated_equals_reference_fixture(golden, case):
    cfg, W, first, n, temp, regz, npseed = _case(golden, case)
    r = restated.RestatedWaveNet(cfg, W)
    np.random.seed(npseed)
    audio, idx, logits = r.generate_fast(n, first_samples=first, temperature=temp, regularize=regz, return_details=True)
    # same ATen ops in the same order => bit-identical float64 audio and indices
    assert np.array_equal(idx, golden["gen_%s_idx" % case].astype(np.int64))
    assert np.array_equal(audio, golden["gen_%s_audio" % case])
    rows = golden["gen_%s_logit_rows" % case]
    assert np.array_equal(logits[rows], golden["gen_%s_logits" % case])


@pytest.mark.parametrize("case", sorted(GEN_CASES))
def test_c_oracle_matches_reference_fixture(golden, case):
    cfg, W, first, n, temp, regz, npseed = _case(golden, case)
    ref_idx = golden["gen_%s_idx" % case].astype(np.int32)
    np.random.seed(npseed)
    u = np.random.random_sample(n)  # one uniform per generated sample, Appendix A item 10
    # teacher-forced on the reference's own index sequence: logits within tolerance
    idx_f, logits = c_oracle.generate(cfg, W, n, first, temp, regz, u, forced=ref_idx)
    rows = golden["gen_%s_logit_rows" % case]
    ref_logits = golden["gen_%s_logits" % case]
    tol = LOGIT_TOL * max(1.0, float(np.abs(ref_logits).max()))
    assert np.abs(logits[rows] - ref_logits).max() <= tol
    # free-running with the same uniforms: identical indices (the C softmax differs from SLEEF's by ulps,
    # so a CDF-boundary hit could in principle flip one; none does for these seeds)
    idx, _ = c_oracle.generate(cfg, W, n, first, temp, regz, u)
    assert np.array_equal(idx, ref_idx)
    assert np.array_equal(idx_f, ref_idx)
    assert np.array_equal(c_oracle.expand(idx), golden["gen_%s_audio" % case])


@pytest.mark.parametrize("cname,seed", [("tiny", 21), ("tiny_bias", 22), ("cfg1", 23)])
def test_greedy_c_oracle_equals_restated(cname, seed):
    """The unmodified reference cannot run its greedy branch (wavenet_model.py:292 IndexError); the torch
    restatement (proven == reference on the sampled branch above) is the anchor for greedy."""
    cfg = synth.CONFIGS[cname]
    W = synth.init_weights(cfg, seed=seed)
    first = np.random.RandomState(seed).randint(0, 256, 40)
    r = restated.RestatedWaveNet(cfg, W)
    audio, idx, logits = r.generate_fast(300, first_samples=first, temperature=0., return_details=True)
    cidx, clog = c_oracle.generate(cfg, W, 300, first, 0., 0.)
    top2 = np.sort(logits, axis=1)
    gap = float((top2[:, -1] - top2[:, -2]).min())
    assert gap > 10 * 1e-6, "weights too degenerate for a bit-exact greedy claim (gap %g)" % gap
    assert np.array_equal(idx.astype(np.int32), cidx)
    assert np.array_equal(audio, c_oracle.expand(cidx))


def test_f32_noise_floor():
    """fp32 vs fp64 evaluation of the same path (teacher forced): documents the tolerance we may claim."""
    cfg = synth.CONFIGS["cfg1"]
    W = synth.init_weights(cfg, seed=5)
    first = np.random.RandomState(5).randint(0, 256, 64)
    idx, l32 = c_oracle.generate(cfg, W, 200, first, 0., 0.)
    _, l64 = c_oracle.generate(cfg, W, 200, first, 0., 0., forced=idx, precision="f64")
    assert np.abs(l32 - l64).max() < 5e-6


def test_expand_matches_numpy_formula():
    idx = np.arange(256)
    assert np.array_equal(c_oracle.expand(idx), restated.mu_law_expansion((idx / 256) * 2. - 1, 256))


# ---------------------------------------------------------------- live reference (authoring container)
needs_ref = pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")


@needs_ref
@pytest.mark.reference
def test_restated_equals_live_reference_sampled():
    mdl, wm, ad = ref_shim.load()
    cfg = synth.CONFIGS["tiny"]
    W = synth.init_weights(cfg, seed=31)
    m = mdl.WaveNetModel(output_length=4, **cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    first = torch.from_numpy(np.random.RandomState(31).randint(0, 256, 25))
    np.random.seed(7)
    a = m.generate_fast(150, first_samples=first, temperature=0.9, regularize=0.003)
    np.random.seed(7)
    b = restated.RestatedWaveNet(cfg, W).generate_fast(150, first_samples=first, temperature=0.9, regularize=0.003)
    assert np.array_equal(a, b)


@needs_ref
@pytest.mark.reference
def test_golden_file_is_reproducible(golden):
    """re-run one generating case of make_golden.py against the live reference"""
    mdl, wm, ad = ref_shim.load()
    cfg, W, first, n, temp, regz, npseed = _case(golden, "tiny")
    m = mdl.WaveNetModel(output_length=8, **cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    np.random.seed(npseed)
    a = m.generate_fast(n, first_samples=torch.from_numpy(first), temperature=temp, regularize=regz)
    assert np.array_equal(a, golden["gen_tiny_audio"])


# ---------------------------------------------------------------- the bf16 training step's oracle (oracle/bf16_step.py)
@pytest.mark.parametrize("case", ["cfg2", "cfg3"])
def test_bf16_step_restatement_is_the_reference_when_nothing_is_rounded(golden, case):
    """oracle/bf16_step.py with round_operands=False against golden_v3.npz -- logits, loss and every parameter gradient the REAL reference
    produced (forward -> F.cross_entropy -> backward) on BASELINE configs[1] and the 50-layer cfg3 stack: 1e-4 / 1e-5 / 2e-5, the bars the
    product's fp32 step is held to.  With the roundings on, the same graph is the bf16 step's oracle (golden_v5.npz, next test)."""
    import os
    import sys
    import bf16_step
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import digest as dg
    wseed, N, out_len = [int(v) for v in golden["grad_%s_meta" % case]][:3]
    cfg = synth.CONFIGS[case]
    W = synth.init_weights(cfg, seed=wseed)
    logits, loss, grads = bf16_step.step(cfg, W, golden["grad_%s_ids" % case].astype(np.int64), golden["grad_%s_target" % case].astype(np.int64),
                                         out_len, round_operands=False)
    assert float(np.abs(logits - golden["grad_%s_out" % case]).max()) <= 1e-4
    assert abs(loss - float(golden["grad_%s_loss" % case][0])) <= 1e-5 * max(1.0, abs(loss))
    want = {k: golden["grad_%s_d_%s" % (case, k)] for k in grads}
    dg.compare(want, dg.digest(grads), 2e-5)


def test_bf16_step_fixture_is_reproducible():
    """golden_v5.npz is what oracle/bf16_step.py computes here and now (cfg2 case; numpy / torch CPU arithmetic in float64 accumulation:
    deterministic), and the roundings really are in it: the logits differ from the unrounded evaluation by about a percent."""
    import os
    import bf16_step
    g5 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v5.npz"))
    wseed, N, out_len, L = [int(v) for v in g5["bf16_cfg2_meta"]]
    cfg = synth.CONFIGS["cfg2"]
    W = synth.init_weights(cfg, seed=wseed)
    logits, loss, _ = bf16_step.step(cfg, W, g5["bf16_cfg2_ids"].astype(np.int64), g5["bf16_cfg2_target"].astype(np.int64), out_len, True)
    assert float(np.abs(logits - g5["bf16_cfg2_out"]).max()) <= 1e-5 and abs(loss - float(g5["bf16_cfg2_loss"][0])) <= 1e-6
    moved, scale = float(g5["bf16_cfg2_noise"][0][1]), float(np.abs(g5["bf16_cfg2_ref_out"]).max())
    assert 1e-4 * scale < moved < 0.05 * scale
    # ... and the evaluation ORDER is not part of the model: fp32 accumulation (another valid order) lands as far from the exact-accumulation result as
    # either lands from the unrounded reference -- at 30 layers the rounding errors of two orders are independent draws (oracle/bf16_step.py)
    bf16_step.ACCUMULATE = "f32"
    try:
        logits32, _, _ = bf16_step.step(cfg, W, g5["bf16_cfg2_ids"].astype(np.int64), g5["bf16_cfg2_target"].astype(np.int64), out_len, True)
    finally:
        bf16_step.ACCUMULATE = "exact"
    between = float(np.linalg.norm(logits32 - logits))
    assert 0.4 * float(g5["bf16_cfg2_noise"][0][0]) < between < 2.0 * float(g5["bf16_cfg2_noise"][0][0])



def test_config5_fixture_inputs_are_regenerated_bit_for_bit():
    """golden_v6.npz (BASELINE configs[4] at its own size) stores no inputs: the GPU tests and bench.py regenerate them from RandomState seeds.  The CRCs
    the generator recorded next to the reference's results prove they are the arrays the reference saw."""
    import zlib
    sys.path.insert(0, os.path.join(ROOT, "pytorch-wavenet_amd"))
    from mi355_wavenet import synth
    z = np.load(os.path.join(ROOT, "tests", "golden", "golden_v6.npz"))
    wseed, dseed, N, L, out_len, crc_ids, crc_tgt = [int(v) for v in z["cfg5_meta"]]
    assert (N, L) == (32, 16000) and out_len == L - synth.receptive_field(synth.CONFIGS["cfg3"]) + 1 == 10885
    rs = np.random.RandomState(dseed)
    ids = rs.randint(0, 256, (N, L))
    target = rs.randint(0, 256, (N, out_len))
    assert zlib.crc32(ids.astype(np.int16).tobytes()) == crc_ids and zlib.crc32(target.astype(np.int16).tobytes()) == crc_tgt
    assert abs(float(z["cfg5_n32_loss"][0]) - np.log(256.0)) > 0.3    # informative weights: not the ln 256 of near-zero logits
    assert z["cfg5_n32_logits"].shape == (32, len(z["cfg5_logit_rows"]), 256) and z["cfg5_n32_bf16_noise"].shape[1] == 5


@needs_ref
@pytest.mark.reference
def test_config5_fixture_is_reproducible_from_the_live_reference():
    """Two clips of golden_v6.npz at the full clip length (16 000 samples, output_length 10 885) re-run through the LIVE reference (~10 s): loss, the
    stored logit rows and every gradient digest come back (same machine, same torch: to fp32 reduction-order noise of the CPU GEMMs)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import digest as dg
    import make_golden as mg
    mdl, _, _ = ref_shim.load()
    z = np.load(os.path.join(ROOT, "tests", "golden", "golden_v6.npz"))
    loss, _, samp, norms, g = mg._ref_cfg5(mdl, 2, 2)
    assert abs(loss - float(z["cfg5_n2_loss"][0])) <= 1e-6 * loss
    assert float(np.abs(samp - z["cfg5_n2_logits"]).max()) <= 2e-5
    assert float(np.abs(norms / z["cfg5_n2_logit_norms"] - 1).max()) <= 1e-6
    head = "cfg5_n2_d_"
    dg.compare({k[len(head):]: z[k] for k in z.files if k.startswith(head)}, dg.digest(g), 2e-5)
