"""-m gpu: the matrix-core forward (wn_forward, csrc/wn_forward.h) against the reference's forward() semantics:
the real reference's golden outputs (tests/golden) and the facade's torch forward (bit-equal to the reference,
tests/test_facade.py) run on the CPU in fp32.  Tolerance 1e-4 (fp32 GEMMs, different summation order)."""
import numpy as np
import pytest
import torch

import wavenet_model
from mi355_wavenet import _abi, engine, synth

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _model(cfg, seed, out_len):
    W = synth.init_weights(cfg, seed=seed)
    m = wavenet_model.WaveNetModel(output_length=out_len, **cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    return m, W


def _onehot(ids):
    N, L = ids.shape
    return torch.zeros(N, 256, L).scatter_(1, torch.from_numpy(ids).view(N, 1, L), 1.)


def test_forward_reproduces_reference_golden(golden):
    wseed, N, out_len = [int(v) for v in golden["fwd_cfg1_meta"]]
    cfg = synth.CONFIGS["cfg1"]
    eng = engine.Engine(cfg, synth.init_weights(cfg, seed=wseed))
    ids = golden["fwd_cfg1_ids"].astype(np.int64)
    y = eng.forward_indices(ids, out_len).cpu().numpy()
    ref = golden["fwd_cfg1_out"]
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() <= TOL


@pytest.mark.parametrize("case", ["cfg2", "cfg3"])
def test_forward_reproduces_reference_golden_beyond_cfg1(golden, case):
    """golden_v3.npz: forward() of the real reference for BASELINE configs[1] and the 10 x 5 / 128 / 128 / 512 stack (N = 1)."""
    wseed, N, out_len = [int(v) for v in golden["grad_%s_meta" % case]]
    cfg = synth.CONFIGS[case]
    eng = engine.Engine(cfg, synth.init_weights(cfg, seed=wseed))
    y = eng.forward_indices(golden["grad_%s_ids" % case].astype(np.int64), out_len).cpu().numpy()
    ref = golden["grad_%s_out" % case]
    assert y.shape == ref.shape
    dev = float(np.abs(y - ref).max())
    print(case, "max |dlogit| vs the reference", dev, "scale", float(np.abs(ref).max()))
    assert dev <= TOL
    eng.close()


CASES = [("cfg1", synth.CONFIGS["cfg1"], 3, 7, 0), ("cfg1_bias", dict(synth.CONFIGS["cfg1"], bias=True), 2, 5, 0),
         ("cfg2", synth.CONFIGS["cfg2"], 2, 9, 0), ("cfg2_long", synth.CONFIGS["cfg2"], 1, 33, 211),
         ("cfg3", synth.CONFIGS["cfg3"], 2, 16, 0)]


@pytest.mark.parametrize("label,cfg,N,out_len,extra", CASES, ids=[c[0] for c in CASES])
def test_forward_matches_torch_reference_path(label, cfg, N, out_len, extra):
    m, W = _model(cfg, 91, out_len)
    L = m.receptive_field + out_len - 1 + extra
    ids = np.random.RandomState(91).randint(0, 256, (N, L))
    with torch.no_grad():
        ref = m(_onehot(ids)).numpy()  # the reference's algorithm with torch ops, CPU fp32
    eng = engine.Engine(cfg, W)
    y = eng.forward_indices(ids, out_len).cpu().numpy()
    assert y.shape == (N * out_len, 256)
    dev = float(np.abs(y - ref).max())
    print(label, "max |dlogit|", dev, "scale", float(np.abs(ref).max()))
    assert dev <= TOL


def test_forward_refuses_lengths_the_reference_has_no_result_for_and_the_facade_stays_native():
    cfg = synth.CONFIGS["cfg1"]
    m, W = _model(cfg, 92, 4)
    eng = engine.Engine(cfg, W)
    with pytest.raises(_abi.WnError) as ei:
        eng.forward_indices(np.zeros((1, 41), dtype=np.int64), 4)  # the skip path's un-dilation quirk
    assert ei.value.code == _abi.WN_E_UNSUPPORTED and "un-dilation" in str(ei.value)
    with pytest.raises(_abi.WnError) as ei:
        eng.forward_indices(np.zeros((1, m.receptive_field + 1), dtype=np.int64), 40)  # rf + 1 samples leave 32 output positions
    assert ei.value.code == _abi.WN_E_UNSUPPORTED and "output positions" in str(ei.value)
    # the facade: CUDA one-hot input without autograd -> native kernel; result equals its own torch path -- for a full-length clip
    # and for one in the reference's zero-padding regime (rf + 1 < rf + output_length - 1)
    mg = None
    for k, L in enumerate((m.receptive_field + 3, m.receptive_field + 1)):
        ids = np.random.RandomState(92 + k).randint(0, 256, (2, L))
        x = _onehot(ids)
        with torch.no_grad():
            ref = m.cpu()(x).numpy()
        mg = m.cuda()
        with torch.no_grad():
            y = mg(x.cuda())
        assert mg._wn_forward_calls == k + 1  # served by wn_forward
        assert np.abs(y.cpu().numpy() - ref).max() <= TOL
    y2 = mg(x.cuda())  # autograd on: the native training forward (wn_train_forward, its backward behind loss.backward()), not wn_forward
    assert mg._wn_forward_calls == 2 and y2.requires_grad and mg._wn_train_calls >= 1


SHORT = ["short_cfg1", "short_cfg1_by1", "short_chaconne", "short_cfg2"]


@pytest.mark.parametrize("case", SHORT)
def test_forward_of_short_clips_reproduces_the_reference_golden(golden, case):
    """golden_v4.npz: forward() of the REAL reference on clips shorter than receptive_field + output_length - 1 -- the layers' inputs are
    left-padded with zero activations there (wavenet_modules.py:24-27) and returned positions see them.  The native kernels read the
    tap x(t - d) as zero on exactly those rows (wn_forward_geometry): same logits, through the engine and through the facade."""
    cfgname = {"short_cfg1": "cfg1", "short_cfg1_by1": "cfg1", "short_chaconne": "chaconne", "short_cfg2": "cfg2"}[case]
    wseed, N, out_len, L, rf = [int(v) for v in golden["grad_%s_meta" % case]]
    assert L < rf + out_len - 1
    cfg = synth.CONFIGS[cfgname]
    m, W = _model(cfg, wseed, out_len)
    ids = golden["grad_%s_ids" % case].astype(np.int64)
    ref = golden["grad_%s_out" % case]
    eng = engine.Engine(cfg, W)
    y = eng.forward_indices(ids, out_len).cpu().numpy()
    eng.close()
    assert y.shape == ref.shape
    dev = float(np.abs(y - ref).max())
    print(case, "L", L, "rf", rf, "max |dlogit| vs the reference", dev, "scale", float(np.abs(ref).max()))
    assert dev <= TOL
    mg = m.cuda()
    with torch.no_grad():
        yf = mg(_onehot(ids).cuda())
    assert mg._wn_forward_calls == 1 and float(np.abs(yf.cpu().numpy() - ref).max()) <= TOL


@pytest.mark.parametrize("cfgname,N,out_len", [("cfg2", 2, 40), ("cfg3", 2, 64)])
def test_forward_bf16_operands_close_to_fp32(cfgname, N, out_len):
    """Opt-in bf16 MFMA operands (fp32 accumulation, fp32 residual stream): the usual bf16 trade, not a parity path --
    logits within 3e-2 of the logit scale of the fp32 kernel, argmax agreement on clearly separated rows."""
    cfg = synth.CONFIGS[cfgname]
    W = synth.init_weights(cfg, seed=93)
    eng = engine.Engine(cfg, W)
    L = synth.receptive_field(cfg) + out_len - 1
    ids = np.random.RandomState(93).randint(0, 256, (N, L))
    y32 = eng.forward_indices(ids, out_len).cpu().numpy()
    eng.set_forward_precision(True)
    y16 = eng.forward_indices(ids, out_len).cpu().numpy()
    eng.set_forward_precision(False)
    again = eng.forward_indices(ids, out_len).cpu().numpy()
    assert np.array_equal(again, y32)
    scale = float(np.abs(y32).max())
    dev = float(np.abs(y16 - y32).max())
    print(cfgname, "bf16 vs fp32 max |dlogit|", dev, "scale", scale)
    assert 0 < dev <= 3e-2 * scale
    top2 = np.sort(y32, axis=1)
    clear = (top2[:, -1] - top2[:, -2]) > 4 * dev
    assert np.array_equal(y16.argmax(1)[clear], y32.argmax(1)[clear])


@pytest.mark.parametrize("L_extra", [63, -40])
def test_forward_bf16_one_launch_per_layer_equals_the_two_launch_form(monkeypatch, L_extra):
    """bf16 operands at the 128 / 128 shape: a layer is ONE launch (wn_fwd_layer_bf16: z handed from the gate epilogue to the residual
    product through LDS, operand reads of x from its bf16 shadow).  Every value is rounded to bf16 once, where it becomes an operand, in
    both forms, and the products accumulate in the same order: the logits are the two-launch form's BIT FOR BIT -- on a clip longer
    than the receptive field and on one in the reference's left-zero-pad regime (row windows in the fused kernel's loader)."""
    monkeypatch.setenv("WN_TESTING", "1")
    cfg = synth.CONFIGS["cfg3"]
    W = synth.init_weights(cfg, seed=7)
    eng = engine.Engine(cfg, W)
    out_len = 64
    L = synth.receptive_field(cfg) + out_len - 1 + L_extra
    eng.set_forward_precision(True)
    if L_extra < 0:   # a short length the reference has a result for (its padding rule refuses most of them: the library says which)
        full = L - L_extra
        for cand in range(L, full):
            try:
                eng.forward_indices(np.zeros((1, cand), dtype=np.int64), out_len)
                L = cand
                break
            except (ValueError, _abi.WnError):
                continue
        else:
            pytest.skip("no short length with a result below %d" % full)
    ids = np.random.RandomState(5).randint(0, 256, (3, L))
    fused = eng.forward_indices(ids, out_len).cpu().numpy()
    monkeypatch.setenv("WN_NO_FUSED_LAYER", "1")
    two = eng.forward_indices(ids, out_len).cpu().numpy()
    monkeypatch.delenv("WN_NO_FUSED_LAYER")
    again = eng.forward_indices(ids, out_len).cpu().numpy()
    assert np.isfinite(fused).all() and float(np.abs(fused).max()) > 0
    assert np.array_equal(fused, two) and np.array_equal(again, fused)
    eng.set_forward_precision(False)
    y32 = eng.forward_indices(ids, out_len).cpu().numpy()
    assert 0 < float(np.abs(fused - y32).max()) <= 3e-2 * float(np.abs(y32).max())


def test_forward_bf16_refused_for_small_channel_counts():
    cfg = synth.CONFIGS["cfg1"]
    eng = engine.Engine(cfg, synth.init_weights(cfg, seed=1))
    with pytest.raises(_abi.WnError) as ei:
        eng.set_forward_precision(True)
    assert ei.value.code == _abi.WN_E_UNSUPPORTED


def test_forward_indices_extension():
    cfg = synth.CONFIGS["cfg1"]
    m, W = _model(cfg, 94, 6)
    ids = np.random.RandomState(94).randint(0, 256, (2, m.receptive_field + 5))
    with torch.no_grad():
        ref = m(_onehot(ids)).numpy()
    y = m.forward_indices(torch.from_numpy(ids)).cpu().numpy()
    assert np.abs(y - ref).max() <= TOL


@pytest.mark.parametrize("cfgname", ["cfg1", "cfg2", "cfg3"])
def test_forward_equals_queue_path_logits_after_priming(cfgname):
    """forward(window)[-1] == the logits the generation path computes for the sample that follows a
    window it was primed on (1.5e-8 in the reference itself).  Ties the two native paths -- GEMM forward and the
    persistent chain with its dilation queues -- to each other, with chain priming and with batched priming."""
    cfg = synth.CONFIGS[cfgname]
    W = synth.init_weights(cfg, seed=44)
    rf = synth.receptive_field(cfg)
    rs = np.random.RandomState(45)
    window = rs.randint(0, 256, (2, rf + 9))  # a little more history than the receptive field
    eng = engine.Engine(cfg, W, n_streams=2)
    fwd = eng.forward_indices(window, 1).cpu().numpy()  # (2, 256): logits for the sample after each window
    scale = max(1.0, float(np.abs(fwd).max()))
    for batched in (False, True):
        _, logits = eng.generate(1, window, temperature=0.0, want_logits=True, batched_prime=batched)
        assert np.abs(logits[:, 0, :] - fwd).max() <= 1e-5 * scale, (cfgname, batched, float(np.abs(logits[:, 0, :] - fwd).max()))
    eng.close()


def test_forward_of_a_zero_padded_channel_shape():
    """Channel counts that are not multiples of 32 (48 / 48 / 300 / 200, bias): the handle runs the model zero-padded into a compiled
    shape (wn_pad_config), so forward() has a NATIVE path for it too -- same logits as the torch graph of the unpadded model, through the
    engine and through the facade (no autograd)."""
    cfg = dict(layers=3, blocks=2, dilation_channels=48, residual_channels=48, skip_channels=300, end_channels=200, classes=256,
               kernel_size=2, bias=True)
    m, W = _model(cfg, 95, 6)
    ids = np.random.RandomState(95).randint(0, 256, (2, m.receptive_field + 6 - 1 + 4))
    x = _onehot(ids)
    with torch.no_grad():
        ref = m(x).numpy()
    eng = engine.Engine(cfg, W)
    assert eng.info()["kernel_variant"] == 3
    y = eng.forward_indices(ids, 6).cpu().numpy()
    eng.close()
    mg = m.cuda()
    with torch.no_grad():
        yf = mg(x.cuda())
    calls = getattr(mg, "_wn_forward_calls", 0)
    dev_e, dev_f, scale = float(np.abs(y - ref).max()), float(np.abs(yf.cpu().numpy() - ref).max()), float(np.abs(ref).max())
    print("padded forward: engine dev", dev_e, "facade dev", dev_f, "scale", scale, "native facade calls", calls)
    assert y.shape == ref.shape and dev_e <= TOL * max(1.0, scale), ("engine", dev_e, scale)
    assert dev_f <= TOL * max(1.0, scale), ("facade", dev_f, scale)
    assert calls == 1, "the facade did not take the native path"
    tiny = wavenet_model.WaveNetModel(layers=2, blocks=2, dilation_channels=8, residual_channels=8, skip_channels=16, end_channels=16,
                                      classes=256, output_length=4, kernel_size=2, bias=False).cuda()
    xt = _onehot(np.random.RandomState(96).randint(0, 256, (1, tiny.receptive_field + 3))).cuda()
    with torch.no_grad():
        a = tiny(xt)     # 8 channels pad into the 16-channel kernel, which has no GEMM banks: the torch graph answers, once and for all
        b = tiny(xt)
    assert torch.equal(a, b) and getattr(tiny, "_wn_forward_calls", 0) == 0 and tiny._wn_forward_unsupported


def test_the_torch_path_is_never_silent():
    """forward() keeps the reference's algorithm in torch ops for what has no native form.  On a CUDA tensor every such call is COUNTED per
    reason (model.wn_stats()) and announced once per reason (RuntimeWarning): kernel_size 3, a soft (non-one-hot) input, WN_TORCH_BACKWARD=1.
    The native path counts too, and a CPU tensor -- the reference's own path -- is not a fallback."""
    import os
    import warnings
    k3 = wavenet_model.WaveNetModel(layers=3, blocks=2, dilation_channels=32, residual_channels=32, skip_channels=64, end_channels=64,
                                    classes=256, output_length=4, kernel_size=3, bias=True)
    x = _onehot(np.random.RandomState(97).randint(0, 256, (2, k3.receptive_field + 3)))
    with torch.no_grad():
        ref = k3(x)
    assert k3.wn_stats() == {"native_forward": 0, "native_train_forward": 0, "torch_fallbacks": {}}   # CPU: nothing to report
    k3 = k3.cuda()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        with torch.no_grad():
            a = k3(x.cuda())
            b = k3(x.cuda())
    assert torch.allclose(a.cpu(), ref, atol=1e-4) and torch.equal(a, b)
    said = [str(i.message) for i in w if issubclass(i.category, RuntimeWarning)]
    assert len(said) == 1 and "kernel_size 3" in said[0], said                   # once, with the reason
    st = k3.wn_stats()
    assert st["native_forward"] == 0 and list(st["torch_fallbacks"].values()) == [2], st
    # a model the kernels DO serve: native calls are counted, a soft input and the A/B switch are fallbacks with their own reasons
    m = wavenet_model.WaveNetModel(layers=3, blocks=2, dilation_channels=32, residual_channels=32, skip_channels=64, end_channels=64,
                                   classes=256, output_length=4, kernel_size=2, bias=False).cuda()
    xg = _onehot(np.random.RandomState(98).randint(0, 256, (2, m.receptive_field + 3))).cuda()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        with torch.no_grad():
            m(xg)
            m(xg * 0.5 + 0.25 / 128)           # not one-hot
        m(xg).sum().backward()                  # native training forward + backward
        os.environ["WN_TORCH_BACKWARD"] = "1"
        try:
            m(xg).sum().backward()
        finally:
            os.environ.pop("WN_TORCH_BACKWARD", None)
    st = m.wn_stats()
    assert st["native_forward"] == 1 and st["native_train_forward"] == 1, st
    assert sorted(st["torch_fallbacks"].values()) == [1, 1] and len([i for i in w if issubclass(i.category, RuntimeWarning)]) == 2, (st, [str(i.message) for i in w])
