"""Generates tests/golden/golden_v1.npz by running the REAL reference (/root/reference, through
oracle/ref_shim.py -- the 3 documented import-time patches, nothing else) in the authoring container.

    python tests/golden/make_golden.py          # golden_v1.npz
    python tests/golden/make_golden.py --v2     # golden_v2.npz (cfg2 / cfg3, a few minutes of reference CPU time)
    python tests/golden/make_golden.py --v3     # golden_v3.npz (forward() + parameter gradients of the reference: cfg2, the cfg3 stack)
    python tests/golden/make_golden.py --v4     # golden_v4.npz (the same for clips SHORTER than receptive_field + output_length - 1: the
                                                #   reference left-pads the layers' activations with zeros there, wavenet_modules.py:24-27)
    python tests/golden/make_golden.py --v6     # golden_v6.npz (BASELINE configs[4] AT ITS OWN SIZE: 32 one-second clips of 16 000 samples, output_length 10 885,
                                                #   the 10 x 5 / 128 / 128 / 512 / 256 stack -- the real reference's loss, logit samples and gradient digests, run
                                                #   four clips at a time (~5 min of CPU), plus the bf16 step's oracle and its noise rows at that size (~40 min))
    python tests/golden/make_golden.py --v5     # golden_v5.npz (the BF16 training step's oracle: oracle/bf16_step.py -- the reference's step with
                                                #   operands rounded where the product rounds them -- after that restatement, roundings off, has been
                                                #   checked against the imported reference on the same cases)

The fixtures pin the oracle (oracle/restated.py, oracle/wn_oracle.c) and, through it, the HIP path.
Weights are NOT stored: they are regenerated from mi355_wavenet.synth.init_weights(cfg, seed) which is
bit-stable (numpy RandomState).  Everything stored here was produced by the reference's own code:

  queue_*      DilatedQueue known answers -- the bodies of /root/reference/tests/test_tensor_queue.py:13-50
  dilate_*     dilate() known answers     -- /root/reference/tests/test_modules.py:8-29
  gen_<case>_* generate_fast(): float64 audio returned by the reference for a seeded run (sampled branch,
               np.random.seed), the given first_samples, and the per-step logits obtained by driving the
               reference's own wavenet(input, queue_dilate) with the same index sequence
  fwd_<case>   forward() output of the reference on a seeded one-hot batch
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "pytorch-wavenet_amd"))

import ref_shim  # noqa: E402
from mi355_wavenet import synth  # noqa: E402

# case -> (config name, weight seed, n_given, num_samples, temperature, regularize, np seed)
GEN_CASES = {
    "tiny": ("tiny", 11, 20, 300, 1.0, 0.0, 101),
    "tiny_bias": ("tiny_bias", 12, 33, 300, 0.8, 0.002, 102),
    "cfg1": ("cfg1", 13, 70, 400, 1.0, 0.0, 103),
    "cfg1_seed128": ("cfg1", 14, 1, 200, 1.0, 0.0, 104),  # default first_samples=None -> [128]
}
# golden_v2.npz: the 10-layer multi-block BASELINE shapes, n_given > 513 so that the d=512 queues wrap while priming and
# again while generating.  Same recipe, same stored arrays, a second file so that v1 stays byte-stable.
GEN_CASES_V2 = {
    "cfg2": ("cfg2", 15, 640, 320, 1.0, 0.0, 105),
    "cfg3": ("cfg3", 16, 700, 320, 0.9, 0.0, 106),
}
FWD_CASES = {"tiny": ("tiny", 11, 2, 5), "tiny_bias": ("tiny_bias", 12, 3, 4), "cfg1": ("cfg1", 13, 1, 3)}


def build_ref_model(mdl, cfg, seed, output_length=8):
    m = mdl.WaveNetModel(output_length=output_length, **cfg)
    W = synth.init_weights(cfg, seed=seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    return m


def indices_from_audio(audio, classes=256):
    """invert audio_data.py:156-158 (monotone) to recover the integer indices the reference sampled."""
    cand = np.arange(classes)
    o = (cand / classes) * 2. - 1
    table = np.sign(o) * (np.exp(np.abs(o) * np.log(classes + 1)) - 1) / classes
    idx = np.array([int(np.argmin(np.abs(table - a))) for a in audio])
    assert np.array_equal(table[idx], audio)
    return idx


def gen_cases(mdl, cases, out):
    """generate_fast on the sampled branch (the only branch the unmodified reference can run)"""
    for case, (cname, wseed, n_given, n, temp, regz, npseed) in cases.items():
        cfg = synth.CONFIGS[cname]
        m = build_ref_model(mdl, cfg, wseed)
        fs = None if n_given == 1 else torch.from_numpy(np.random.RandomState(wseed).randint(0, 256, n_given))
        np.random.seed(npseed)
        audio = m.generate_fast(n, first_samples=fs, temperature=temp, regularize=regz)
        idx = indices_from_audio(audio)
        # per-step logits from the reference's own wavenet(): replay the same index sequence
        for qq in m.dilated_queues:
            qq.reset()
        given = [128] if fs is None else fs.tolist()
        seq = given + idx.tolist()
        logits = []
        for t in range(len(seq) - 1):
            inp = torch.zeros(1, 256, 1)
            inp[0, seq[t], 0] = 1.
            y = m.wavenet(inp, dilation_func=m.queue_dilate).squeeze()
            if t >= len(given) - 1:
                logits.append(y.detach().numpy().copy())
        logits = np.stack(logits)
        assert logits.shape == (n, 256)
        out["gen_%s_audio" % case] = audio
        out["gen_%s_idx" % case] = idx.astype(np.int16)
        out["gen_%s_first" % case] = np.asarray(given, dtype=np.int16)
        out["gen_%s_logits" % case] = logits[:: max(1, n // 16)].astype(np.float32)  # 16-ish rows is enough
        out["gen_%s_logit_rows" % case] = np.arange(n)[:: max(1, n // 16)].astype(np.int32)
        out["gen_%s_meta" % case] = np.array([wseed, n_given, n, npseed], dtype=np.int64)
        out["gen_%s_tr" % case] = np.array([temp, regz], dtype=np.float64)



def main():
    mdl, wm, ad = ref_shim.load()
    out = {}

    # ---- DilatedQueue known answers (reference tests/test_tensor_queue.py:13-50 bodies) ----
    q = wm.DilatedQueue(max_length=8, num_channels=3)
    e = torch.zeros(3)
    for i in range(11):
        e = e + 1
        q.enqueue(e)
    out["queue_enqueue_data"] = q.data.numpy().copy()
    assert q.data[0, 0] == 9 and q.data[0, 2] == 11 and q.data[0, 7] == 8
    q = wm.DilatedQueue(max_length=8, num_channels=1)
    e = torch.zeros(1)
    for i in range(11):
        e = e + 1
        q.enqueue(e)
    deq = []
    for i in range(9):
        d = q.dequeue(num_deq=3, dilation=2)
        deq.append(d.numpy().copy())
    assert d[0][0] == 5 and d[0][1] == 7 and d[0][2] == 9
    out["queue_dequeue_seq"] = np.stack(deq)
    q = wm.DilatedQueue(max_length=12, num_channels=1)
    e = torch.zeros(1)
    comb = []
    for i in range(30):
        e = e + 1
        q.enqueue(e)
        d = q.dequeue(num_deq=3, dilation=4)
        assert d[0][0] == max(i - 7, 0)
        comb.append(d.numpy().copy())
    out["queue_combined_seq"] = np.stack(comb)

    # ---- dilate known answers (reference tests/test_modules.py:8-29) ----
    x = torch.linspace(0, 12, steps=13).view(1, 1, 13)
    d2 = wm.dilate(x, 2)
    d4 = wm.dilate(d2, 4, init_dilation=2)
    d1 = wm.dilate(d4, 1, init_dilation=4)
    assert d2.size() == (2, 1, 7) and d2[1, 0, 2] == 4
    assert d4.size() == (4, 1, 4) and d4[3, 0, 1] == 4
    assert d1.size() == (1, 1, 16) and d1[0, 0, 7] == 4
    out["dilate_in"], out["dilate_d2"], out["dilate_d4"], out["dilate_d1"] = x.numpy(), d2.numpy(), d4.numpy(), d1.numpy()
    xm = torch.linspace(0, 35, steps=36).view(2, 3, 6)
    out["dilate_mc_in"], out["dilate_mc_d4"] = xm.numpy(), wm.dilate(xm, 4).numpy()

    gen_cases(mdl, GEN_CASES, out)

    # ---- forward() ----
    for case, (cname, wseed, N, out_len) in FWD_CASES.items():
        cfg = synth.CONFIGS[cname]
        m = build_ref_model(mdl, cfg, wseed, output_length=out_len)
        L = m.receptive_field + out_len - 1
        ids = np.random.RandomState(wseed + 1).randint(0, 256, (N, L))
        x = torch.zeros(N, 256, L)
        x.scatter_(1, torch.from_numpy(ids).view(N, 1, L), 1.)
        y = m(x).detach().numpy()
        out["fwd_%s_ids" % case] = ids.astype(np.int16)
        out["fwd_%s_out" % case] = y.astype(np.float32)
        out["fwd_%s_meta" % case] = np.array([wseed, N, out_len], dtype=np.int64)

    path = os.path.join(HERE, "golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


def main_v2():
    mdl, wm, ad = ref_shim.load()
    out = {}
    gen_cases(mdl, GEN_CASES_V2, out)
    path = os.path.join(HERE, "golden_v2.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


# golden_v3.npz: forward() AND parameter gradients of the reference (forward -> F.cross_entropy -> backward, wavenet_training.py:64-72)
# beyond cfg1: BASELINE configs[1] (cfg2) and the 10 x 5 / 128 / 128 / 512 stack of configs[2..4], N = 1, short output_length.
# case -> (config, weight seed, N, output_length)
GRAD_CASES = {"cfg2": ("cfg2", 21, 1, 6), "cfg3": ("cfg3", 22, 1, 4), "tiny_bias": ("tiny_bias", 23, 2, 5)}


# golden_v4.npz: the zero-padding regime of forward() -- clips shorter than receptive_field + output_length - 1, whose returned positions
# see the zero activations `dilate` pads the layers' inputs with on the left.  case -> (config, weight seed, N, output_length, L)
SHORT_CASES = {
    "short_cfg1": ("cfg1", 31, 2, 5, 64),            # rf 63: three samples short (shorter clips run into the skip path quirk, Appendix A item 17)
    "short_cfg1_by1": ("cfg1", 32, 1, 8, 69),        # rf + out_len - 2: one sample short
    "short_tiny_bias": ("tiny_bias", 33, 3, 4, 16),
    "short_chaconne": ("chaconne", 34, 1, 16, 2600),  # train_script.py shape (rf 3070): 485 samples short
    "short_cfg2": ("cfg2", 35, 1, 6, 2751),
}


def _grad_case(mdl, out, case, cname, wseed, N, out_len, L=None):
    import torch.nn.functional as F
    import digest as dg
    if True:
        cfg = synth.CONFIGS[cname]
        m = build_ref_model(mdl, cfg, wseed, output_length=out_len)
        L = m.receptive_field + out_len - 1 if L is None else L
        rs = np.random.RandomState(wseed + 1)
        ids = rs.randint(0, 256, (N, L))
        target = rs.randint(0, 256, (N * out_len,))
        x = torch.zeros(N, 256, L)
        x.scatter_(1, torch.from_numpy(ids).view(N, 1, L), 1.)
        y = m(x)                                                      # wavenet_model.py:186-196
        loss = F.cross_entropy(y.squeeze(), torch.from_numpy(target))  # wavenet_training.py:69
        loss.backward()
        out["grad_%s_ids" % case] = ids.astype(np.int16)
        out["grad_%s_target" % case] = target.astype(np.int16)
        out["grad_%s_out" % case] = y.detach().numpy().astype(np.float32)
        out["grad_%s_loss" % case] = np.array([float(loss)], dtype=np.float64)
        out["grad_%s_meta" % case] = np.array([wseed, N, out_len, L, m.receptive_field], dtype=np.int64)
        named = {k: (p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), dtype=np.float32)) for k, p in m.named_parameters()}
        for k, v in dg.digest(named).items():
            out["grad_%s_d_%s" % (case, k)] = v
        print(case, "L", L, "rf", m.receptive_field, "loss", float(loss), "params", len(named))


def main_v4():
    sys.path.insert(0, HERE)
    mdl, wm, ad = ref_shim.load()
    out = {}
    for case, (cname, wseed, N, out_len, L) in SHORT_CASES.items():
        _grad_case(mdl, out, case, cname, wseed, N, out_len, L)
    path = os.path.join(HERE, "golden_v4.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


def main_v3():
    import torch.nn.functional as F
    sys.path.insert(0, HERE)
    import digest as dg
    mdl, wm, ad = ref_shim.load()
    out = {}
    for case, (cname, wseed, N, out_len) in GRAD_CASES.items():
        cfg = synth.CONFIGS[cname]
        m = build_ref_model(mdl, cfg, wseed, output_length=out_len)
        L = m.receptive_field + out_len - 1
        rs = np.random.RandomState(wseed + 1)
        ids = rs.randint(0, 256, (N, L))
        target = rs.randint(0, 256, (N * out_len,))
        x = torch.zeros(N, 256, L)
        x.scatter_(1, torch.from_numpy(ids).view(N, 1, L), 1.)
        y = m(x)                                                      # wavenet_model.py:186-196
        loss = F.cross_entropy(y.squeeze(), torch.from_numpy(target))  # wavenet_training.py:69
        loss.backward()
        out["grad_%s_ids" % case] = ids.astype(np.int16)
        out["grad_%s_target" % case] = target.astype(np.int16)
        out["grad_%s_out" % case] = y.detach().numpy().astype(np.float32)
        out["grad_%s_loss" % case] = np.array([float(loss)], dtype=np.float64)
        out["grad_%s_meta" % case] = np.array([wseed, N, out_len], dtype=np.int64)
        named = {k: (p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), dtype=np.float32)) for k, p in m.named_parameters()}
        for k, v in dg.digest(named).items():
            out["grad_%s_d_%s" % (case, k)] = v
        print(case, "loss", float(loss), "params", len(named))
    path = os.path.join(HERE, "golden_v3.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


# golden_v5.npz: the oracle of the bf16 training step.  case -> (config name or dict, weight seed, N, output_length)
B64 = dict(layers=4, blocks=2, dilation_channels=64, residual_channels=64, skip_channels=128, end_channels=128, classes=256, kernel_size=2, bias=True)
BF16_CASES = {"cfg3": ("cfg3", 22, 1, 4), "cfg2": ("cfg2", 21, 1, 6), "b64": (B64, 24, 2, 5)}


def main_v5():
    import torch.nn.functional as F
    sys.path.insert(0, HERE)
    import digest as dg
    import bf16_step
    mdl, wm, ad = ref_shim.load()
    out = {}
    for case, (cname, wseed, N, out_len) in BF16_CASES.items():
        cfg = synth.CONFIGS[cname] if isinstance(cname, str) else cname
        W = synth.init_weights(cfg, seed=wseed)
        m = build_ref_model(mdl, cfg, wseed, output_length=out_len)
        L = m.receptive_field + out_len - 1
        rs = np.random.RandomState(wseed + 1)
        ids = rs.randint(0, 256, (N, L))
        target = rs.randint(0, 256, (N * out_len,))
        # (1) the REAL reference, fp32
        x = torch.zeros(N, 256, L)
        x.scatter_(1, torch.from_numpy(ids).view(N, 1, L), 1.)
        y = m(x)
        loss = F.cross_entropy(y.squeeze(), torch.from_numpy(target))
        loss.backward()
        ref_g = {k: (p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), dtype=np.float32)) for k, p in m.named_parameters()}
        ref_d = dg.digest(ref_g)
        # (2) the restatement with its roundings OFF must be the reference
        lo0, ls0, g0 = bf16_step.step(cfg, W, ids, target, out_len, round_operands=False)
        dev_out = float(np.abs(lo0 - y.detach().numpy()).max())
        assert dev_out <= 2e-5 and abs(ls0 - float(loss)) <= 1e-6 * max(1.0, abs(ls0)), (case, dev_out, ls0, float(loss))
        worst = dg.compare(ref_d, dg.digest(g0), 2e-5)
        print(case, "restatement (no rounding) vs the imported reference: logits", dev_out, "gradient digests", worst)
        # (3) ... and with the product's rounding points ON it is the bf16 step's oracle -- up to its own noise: three evaluation orders of the SAME
        # rounding model (exact / fp32 / fp32 over a permuted K axis accumulation) and how far each lands from the fp32 reference (bf16_step.py says why)
        def digest_devs(d):   # per tensor: the digest.compare() measure against the reference's gradients
            out_ = []
            for k, r in ref_d.items():
                if r[0] > 0:
                    g = d[k]
                    out_.append(max(abs(g[0] - r[0]) / r[0], abs(g[1] - r[1]) / r[1], float(np.abs(g[2:6] - r[2:6]).max()) / r[1], float(np.abs(g[6:] - r[6:]).max()) / r[0]))
            return np.array(out_)
        noise = []
        lo1 = ls1 = d1 = None
        for order in ("exact", "f32", "f32perm"):
            bf16_step.ACCUMULATE = order
            lo, ls, g = bf16_step.step(cfg, W, ids, target, out_len, round_operands=True)
            dv = digest_devs(dg.digest(g))
            noise.append([float(np.linalg.norm(lo - lo0)), float(np.abs(lo - lo0).max()), abs(ls - ls0), float(np.sqrt((dv ** 2).mean())), float(dv.max())])
            if order == "exact":
                lo1, ls1, d1 = lo, ls, dg.digest(g)
            print(case, order, "vs the fp32 reference: logits norm %.4f max %.4f (scale %.3f), loss %.5f, gradient digests rms %.4f max %.4f" % (
                tuple(noise[-1][:2]) + (float(np.abs(lo0).max()),) + tuple(noise[-1][2:])))
        bf16_step.ACCUMULATE = "exact"
        out["bf16_%s_ids" % case] = ids.astype(np.int16)
        out["bf16_%s_target" % case] = target.astype(np.int16)
        out["bf16_%s_out" % case] = lo1.astype(np.float32)                             # (exact accumulation)
        out["bf16_%s_ref_out" % case] = lo0.astype(np.float32)                         # the unrounded evaluation = the reference's fp32 logits
        out["bf16_%s_loss" % case] = np.array([ls1, ls0], dtype=np.float64)            # [with the roundings, without]
        out["bf16_%s_meta" % case] = np.array([wseed, N, out_len, L], dtype=np.int64)
        out["bf16_%s_noise" % case] = np.array(noise, dtype=np.float64)                # rows: exact, f32, f32perm; columns: logit norm, logit max, |dloss|, digest rms, digest max
        for k, v in d1.items():
            out["bf16_%s_d_%s" % (case, k)] = v
        for k, v in ref_d.items():
            out["bf16_%s_r_%s" % (case, k)] = v                                        # the REFERENCE's gradient digests (fp32), the measure's origin
    path = os.path.join(HERE, "golden_v5.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


# golden_v6.npz: BASELINE configs[4] at its own geometry.  N = 32 clips of L = 16 000 samples, output_length = L - rf + 1 = 10 885,
# cfg3's stack.  Nothing of that size was pinned before round 6 (largest case: N <= 2, output_length <= 64).  The inputs are NOT stored: they are
# regenerated from numpy RandomState seeds (bit-stable) and guarded by CRCs.  The reference runs FOUR clips per forward()/backward() call (its whole
# batch at once would need ~130 GB of autograd state) with the chunk losses averaged -- F.cross_entropy's mean over N * output_length rows of equal-sized
# chunks -- and gradients accumulated in .grad by loss_c / n_chunks: the reference's own code on every clip, fp32 summation order across chunks aside.
CFG5 = dict(cname="cfg3", wseed=41, dseed=42, N=32, L=16000, chunk=4)


def cfg5_inputs(N=CFG5["N"], L=CFG5["L"], dseed=CFG5["dseed"]):
    """(ids (N, L), target (N * out_len,)) of the config-5 fixture: shared by the generator, the GPU tests and bench.py's `verified` leg."""
    cfg = synth.CONFIGS[CFG5["cname"]]
    out_len = L - synth.receptive_field(cfg) + 1
    rs = np.random.RandomState(dseed)
    ids = rs.randint(0, 256, (CFG5["N"], L))
    target = rs.randint(0, 256, (CFG5["N"], out_len))
    return ids[:N], target[:N].reshape(-1), out_len


def _ref_cfg5(mdl, clips, chunk, double=False, ignore=None, record_tau=None):
    """The real reference on clips [0, clips): returns (loss, chunk losses, sampled logits, per-clip logit norms, gradients[, ambiguous rows]).
    double: the reference's own code in float64 (model.double()): the truth its fp32 arithmetic approximates.  ignore: bool (clips, out_len) -- those rows
    get F.cross_entropy's ignore_index (-100) as their target.  record_tau: also return the rows (clip, position) at which a ReLU input of the head
    (wavenet_model.py:167-168: the skip sum, end_conv_1's output) lies within record_tau of zero -- the rows whose ReLU mask two fp32 evaluations can
    disagree on."""
    import torch.nn.functional as F
    import digest as dg
    cfg = synth.CONFIGS[CFG5["cname"]]
    ids, target, out_len = cfg5_inputs(clips)
    m = build_ref_model(mdl, cfg, CFG5["wseed"], output_length=out_len)
    if double:
        m = m.double()
    L = ids.shape[1]
    n_chunks = (clips + chunk - 1) // chunk
    assert clips % chunk == 0
    tgt = target.reshape(clips, out_len).copy()
    if ignore is not None:
        tgt[ignore] = -100
    n_valid = int((tgt >= 0).sum())
    losses, samples, norms, amb = [], [], [], []
    real_relu = F.relu
    for c in range(n_chunks):
        sl = slice(c * chunk, (c + 1) * chunk)
        x = torch.zeros(chunk, 256, L, dtype=torch.float64 if double else torch.float32)
        x.scatter_(1, torch.from_numpy(ids[sl]).view(chunk, 1, L), 1.)
        seen = []
        if record_tau is not None:
            def spy(v, *a, **k):
                seen.append((v.detach()[:, :, -out_len:].abs() < record_tau).any(dim=1))   # (n, out_len): any channel of this ReLU input near zero
                return real_relu(v, *a, **k)
            F.relu = spy
        try:
            y = m(x)                                                                                   # wavenet_model.py:186-196
        finally:
            F.relu = real_relu
        if record_tau is not None:
            assert len(seen) == 2, len(seen)   # F.relu(skip), F.relu(end_conv_1(x))
            amb.append((seen[0] | seen[1]).numpy())
        t_c = torch.from_numpy(tgt[sl].reshape(-1))
        # F.cross_entropy's mean runs over the rows that are not ignored: chunk sums / the batch's valid rows == the batch's mean
        loss = F.cross_entropy(y.squeeze(), t_c, reduction="sum") / n_valid                          # wavenet_training.py:69 (mean over the batch)
        loss.backward()
        yy = y.detach().numpy().reshape(chunk, out_len, 256)
        samples.append(yy[:, dg.logit_rows(out_len), :].astype(np.float32))
        norms.append(np.sqrt((yy.astype(np.float64) ** 2).sum(axis=(1, 2))))
        losses.append(float(loss.detach()))
        print("  reference%s chunk %d/%d loss share %.6f" % (" (float64)" if double else "", c + 1, n_chunks, losses[-1]), flush=True)
    g = {k: (p.grad.numpy().copy() if p.grad is not None else np.zeros(tuple(p.shape), dtype=np.float32)) for k, p in m.named_parameters()}
    out = (float(np.sum(losses)), np.array(losses), np.concatenate(samples), np.concatenate(norms), g)
    return out + (np.concatenate(amb),) if record_tau is not None else out


def _bf16_cfg5(clips, order, chunk=2):
    """oracle/bf16_step.py on clips [0, clips), two at a time: gradients averaged over the chunks.  Exactly the full-batch evaluation under the
    rounding model -- the chunks' dlogits differ from the batch's by the factor n_chunks, a power of two, which no rounding to bf16 sees."""
    import bf16_step
    import digest as dg
    cfg = synth.CONFIGS[CFG5["cname"]]
    W = synth.init_weights(cfg, seed=CFG5["wseed"])
    ids, target, out_len = cfg5_inputs(clips)
    n_chunks = clips // chunk
    assert n_chunks & (n_chunks - 1) == 0, "power-of-two chunk counts only (see the docstring)"
    bf16_step.ACCUMULATE = order
    G, losses, samples, norms = None, [], [], []
    for c in range(n_chunks):
        sl = slice(c * chunk, (c + 1) * chunk)
        lo, ls, g = bf16_step.step(cfg, W, ids[sl], target.reshape(clips, out_len)[sl].reshape(-1), out_len, round_operands=True)
        yy = lo.reshape(chunk, out_len, 256)
        samples.append(yy[:, dg.logit_rows(out_len), :].astype(np.float32))
        norms.append(np.sqrt((yy.astype(np.float64) ** 2).sum(axis=(1, 2))))
        losses.append(ls)
        G = {k: v.astype(np.float64) / n_chunks for k, v in g.items()} if G is None else {k: G[k] + v.astype(np.float64) / n_chunks for k, v in g.items()}
        print("  bf16 oracle (%s) chunk %d/%d loss %.6f" % (order, c + 1, n_chunks, ls), flush=True)
    bf16_step.ACCUMULATE = "exact"
    return float(np.mean(losses)), np.concatenate(samples), np.concatenate(norms), {k: v.astype(np.float32) for k, v in G.items()}


def main_v6():
    import zlib
    sys.path.insert(0, HERE)
    import digest as dg
    mdl, wm, ad = ref_shim.load()
    path = os.path.join(HERE, "golden_v6.npz")
    out = dict(np.load(path)) if ("--resume" in sys.argv and os.path.exists(path)) else {}
    ids, target, out_len = cfg5_inputs()
    out["cfg5_meta"] = np.array([CFG5["wseed"], CFG5["dseed"], CFG5["N"], CFG5["L"], out_len, zlib.crc32(ids.astype(np.int16).tobytes()), zlib.crc32(target.astype(np.int16).tobytes())], dtype=np.int64)
    out["cfg5_logit_rows"] = dg.logit_rows(out_len).astype(np.int64)

    def devs(ref_d, d):
        o = []
        for k, r in ref_d.items():
            if r[0] > 0:
                g = d[k]
                o.append(max(abs(g[0] - r[0]) / r[0], abs(g[1] - r[1]) / r[1], float(np.abs(g[2:6] - r[2:6]).max()) / r[1], float(np.abs(g[6:] - r[6:]).max()) / r[0]))
        return np.array(o)

    for tag, clips, chunk, orders in (("n2", 2, 2, ("exact", "f32", "f32perm")), ("n32", CFG5["N"], CFG5["chunk"], ("exact", "f32"))):
        if "cfg5_%s_loss" % tag not in out:
            print("cfg5 %s: the real reference, %d clips" % (tag, clips), flush=True)
            loss, losses, samp, norms, g = _ref_cfg5(mdl, clips, chunk)
            out["cfg5_%s_loss" % tag] = np.array([loss], dtype=np.float64)
            out["cfg5_%s_chunk_losses" % tag] = losses
            out["cfg5_%s_logits" % tag] = samp                       # (clips, 8, 256): rows dg.logit_rows(out_len) of every clip
            out["cfg5_%s_logit_norms" % tag] = norms                 # (clips,) float64
            for k, v in dg.digest(g).items():
                out["cfg5_%s_d_%s" % (tag, k)] = v
            np.savez_compressed(path, **out)
            print("cfg5 %s: loss %.6f (ln 256 = %.4f)" % (tag, loss, np.log(256.0)), flush=True)
        if "--no-bf16" in sys.argv or "cfg5_%s_bf16_noise" % tag in out:
            continue
        ref_d = {k[len("cfg5_%s_d_" % tag):]: out[k] for k in out if k.startswith("cfg5_%s_d_" % tag)}
        noise = []
        for order in orders:
            ls, samp, norms, g = _bf16_cfg5(clips, order)
            d = dg.digest(g)
            dv = devs(ref_d, d)
            dl = samp.astype(np.float64) - out["cfg5_%s_logits" % tag]
            noise.append([float(np.sqrt((dl ** 2).mean())), float(np.abs(dl).max()), abs(ls - float(out["cfg5_%s_loss" % tag][0])), float(np.sqrt((dv ** 2).mean())), float(dv.max())])
            print("cfg5 %s bf16 oracle (%s) vs the fp32 reference: sampled logits rms %.4f max %.4f, |dloss| %.5f, gradient digests rms %.4f max %.4f" % ((tag, order) + tuple(noise[-1])), flush=True)
            if order == "exact":
                out["cfg5_%s_bf16_loss" % tag] = np.array([ls], dtype=np.float64)
                out["cfg5_%s_bf16_logits" % tag] = samp
                for k, v in d.items():
                    out["cfg5_%s_bf16_d_%s" % (tag, k)] = v
        out["cfg5_%s_bf16_noise" % tag] = np.array(noise, dtype=np.float64)   # rows: the orders; columns: sampled-logit rms, max, |dloss|, digest rms, digest max (all vs the fp32 reference)
        np.savez_compressed(path, **out)
    # ---- what fp32 can and cannot pin at this size.  The head has two ReLUs (wavenet_model.py:167-168) on 32 x 10 885 rows x (512 + 256) channels = 267 M
    # pre-activations; where one lies within fp32 noise of zero, two correct fp32 evaluations disagree on its mask, and the weight gradients change by that
    # row's whole contribution -- ~1e-3 of a tensor's largest element per flip.  Measured: the reference's OWN fp32 gradients differ from its float64
    # evaluation by 7e-3 from output_length ~3000 up (5e-6 below ~600), and so does any other fp32 implementation (tools/debug_grad_scaling.py).
    # (a) "n32r": the rows with a ReLU input within TAU of zero (either ReLU, any channel: a few per cent) get ignore_index as their target -- no gradient
    #     flows through an ambiguous mask, and every gradient is pinned at 2e-5 again;  (b) "n32 f64": the reference's code in float64 on the plain targets --
    #     the truth -- and how far its fp32 evaluation lands from it: the bar for the plain step.
    TAU = 2e-4
    if "cfg5_n32r_loss" not in out and "--no-robust" not in sys.argv:
        print("cfg5 n32: ambiguous ReLU rows of the real reference (tau %g)" % TAU, flush=True)
        amb = _ref_cfg5(mdl, CFG5["N"], CFG5["chunk"], record_tau=TAU)[5]
        print("cfg5 n32r: %d of %d rows ignored" % (int(amb.sum()), amb.size), flush=True)
        loss, losses, samp, norms, g = _ref_cfg5(mdl, CFG5["N"], CFG5["chunk"], ignore=amb)
        out["cfg5_n32r_ignore"] = np.packbits(amb.reshape(-1))
        out["cfg5_n32r_tau"] = np.array([TAU])
        out["cfg5_n32r_loss"] = np.array([loss], dtype=np.float64)
        for k, v in dg.digest(g).items():
            out["cfg5_n32r_d_%s" % k] = v
        np.savez_compressed(path, **out)
        print("cfg5 n32r: loss %.6f" % loss, flush=True)
    if "cfg5_n32_fp32_noise" not in out and "--no-f64" not in sys.argv:
        print("cfg5 n32: the reference in float64", flush=True)
        loss, losses, samp, norms, g = _ref_cfg5(mdl, CFG5["N"], 2, double=True)
        d64 = dg.digest(g)
        for k, v in d64.items():
            out["cfg5_n32_f64_d_%s" % k] = v
        out["cfg5_n32_f64_loss"] = np.array([loss], dtype=np.float64)
        ref_d = {k[len("cfg5_n32_d_"):]: out[k] for k in out if k.startswith("cfg5_n32_d_")}
        dv = devs(d64, ref_d)   # the fp32 reference measured against the float64 truth
        out["cfg5_n32_fp32_noise"] = np.array([float(np.sqrt((dv ** 2).mean())), float(dv.max()), abs(loss - float(out["cfg5_n32_loss"][0]))])
        print("cfg5 n32: the reference's fp32 step vs its float64 evaluation: gradient digests rms %.2e max %.2e, |dloss| %.1e" % tuple(out["cfg5_n32_fp32_noise"]), flush=True)
        np.savez_compressed(path, **out)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    if "--v6" in sys.argv:
        main_v6()
    elif "--v5" in sys.argv:
        main_v5()
    elif "--v4" in sys.argv:
        main_v4()
    elif "--v3" in sys.argv:
        main_v3()
    elif "--v2" in sys.argv:
        main_v2()
    elif "--all" in sys.argv:
        main()
        main_v2()
        main_v3()
        main_v4()
    else:
        main()
