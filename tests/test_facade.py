"""The drop-in Python surface (pytorch-wavenet_amd/wavenet_model.py, wavenet_modules.py, audio_data.py)
against the reference's own known answers and golden outputs.  generate_fast() is exercised here on the
host-memory test double of the C ABI (tests/double, explicitly injected); on the GPU it is covered by tests/test_gpu_facade.py."""
import io
import os
import pickle
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

import c_oracle
import wavenet_model
import wavenet_modules
from double_lib import double_backend, double_library
from mi355_wavenet import engine, synth


# ---------------------------------------------------------------- wavenet_modules known answers
def test_dilated_queue_reference_known_answers(golden):
    # /root/reference/tests/test_tensor_queue.py:13-50
    q = wavenet_modules.DilatedQueue(max_length=8, num_channels=3)
    e = torch.zeros(3)
    for _ in range(11):
        e = e + 1
        q.enqueue(e)
    assert q.data[0, 0] == 9 and q.data[0, 2] == 11 and q.data[0, 7] == 8
    assert np.array_equal(q.data.numpy(), golden["queue_enqueue_data"])
    q = wavenet_modules.DilatedQueue(max_length=8, num_channels=1)
    e = torch.zeros(1)
    for _ in range(11):
        e = e + 1
        q.enqueue(e)
    seq = [q.dequeue(num_deq=3, dilation=2).numpy().copy() for _ in range(9)]
    assert list(seq[-1][0]) == [5, 7, 9]
    assert np.array_equal(np.stack(seq), golden["queue_dequeue_seq"])
    q = wavenet_modules.DilatedQueue(max_length=12, num_channels=1)
    e = torch.zeros(1)
    seq = []
    for i in range(30):
        e = e + 1
        q.enqueue(e)
        d = q.dequeue(num_deq=3, dilation=4)
        assert d[0][0] == max(i - 7, 0)
        seq.append(d.numpy().copy())
    assert np.array_equal(np.stack(seq), golden["queue_combined_seq"])
    q.reset()
    assert q.in_pos == 0 and q.out_pos == 0 and float(q.data.abs().sum()) == 0


def test_dilate_reference_known_answers(golden):
    # /root/reference/tests/test_modules.py:8-29
    x = torch.linspace(0, 12, steps=13).view(1, 1, 13)
    assert wavenet_modules.dilate(x, 1) is x
    d2 = wavenet_modules.dilate(x, 2)
    assert d2.size() == (2, 1, 7) and d2[1, 0, 2] == 4
    d4 = wavenet_modules.dilate(d2, 4, init_dilation=2)
    assert d4.size() == (4, 1, 4) and d4[3, 0, 1] == 4
    d1 = wavenet_modules.dilate(d4, 1, init_dilation=4)
    assert d1.size() == (1, 1, 16) and d1[0, 0, 7] == 4
    for name, t in (("dilate_d2", d2), ("dilate_d4", d4), ("dilate_d1", d1)):
        assert np.array_equal(t.numpy(), golden[name])
    xm = torch.from_numpy(golden["dilate_mc_in"])
    assert np.array_equal(wavenet_modules.dilate(xm, 4).numpy(), golden["dilate_mc_d4"])


def test_constant_pad_1d_forward_and_grad():
    x = torch.ones(2, 3, requires_grad=True)
    y = wavenet_modules.constant_pad_1d(x, 5, dimension=1, pad_start=True)
    assert y.shape == (2, 5) and float(y[:, :2].detach().sum()) == 0 and float(y[:, 2:].detach().sum()) == 6
    y.sum().backward()
    assert torch.equal(x.grad, torch.ones(2, 3))
    with pytest.raises(AssertionError):
        wavenet_modules.constant_pad_1d(x, 2, dimension=1)


# ---------------------------------------------------------------- model surface
def _model(cname, seed, **kw):
    cfg = synth.CONFIGS[cname]
    W = synth.init_weights(cfg, seed=seed)
    m = wavenet_model.WaveNetModel(**dict(cfg, **kw))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    return m, cfg, W


def test_constructor_surface_matches_reference():
    m = wavenet_model.WaveNetModel(layers=10, blocks=3, dilation_channels=32, residual_channels=32,
                                   skip_channels=1024, end_channels=512, output_length=16, bias=True)  # train_script.py:17-25
    assert m.receptive_field == 3070 and m.output_length == 16 and m.classes == 256
    assert m.parameter_count() == synth.parameter_count(synth.CONFIGS["chaconne"]) == 1834592
    assert list(m.state_dict().keys()) == list(synth.param_shapes(synth.CONFIGS["chaconne"]).keys())
    assert m.dilations[0] == (1, 1) and m.dilations[1] == (2, 1) and m.dilations[10] == (1, 512)
    assert [q.max_length for q in m.dilated_queues[:4]] == [2, 3, 5, 9]
    assert wavenet_model.WaveNetModel().receptive_field == 4093  # defaults 10x4


@pytest.mark.parametrize("case", ["tiny", "tiny_bias", "cfg1"])
def test_forward_matches_reference_golden(golden, case):
    wseed, N, out_len = [int(v) for v in golden["fwd_%s_meta" % case]]
    m, cfg, W = _model(case, wseed, output_length=out_len)
    ids = torch.from_numpy(golden["fwd_%s_ids" % case].astype(np.int64))
    L = ids.shape[1]
    x = torch.zeros(N, 256, L).scatter_(1, ids.view(N, 1, L), 1.)
    y = m(x)
    assert y.shape == (N * out_len, 256)
    assert np.array_equal(y.detach().numpy(), golden["fwd_%s_out" % case])  # same ATen ops, same order
    loss = F_cross_entropy(y, ids[:, -out_len:].reshape(-1))
    loss.backward()
    last = "residual_convs.%d." % (cfg["layers"] * cfg["blocks"] - 1)  # its output is never consumed (also upstream)
    for name, p in m.named_parameters():
        if name.startswith(last):
            assert p.grad is None
        else:
            assert p.grad is not None and torch.isfinite(p.grad).all(), name


def F_cross_entropy(y, t):
    return torch.nn.functional.cross_entropy(y, t)


def test_forward_equals_queue_path():
    """forward(window)[-1] == logits of the queue path primed with the same window."""
    m, cfg, W = _model("tiny", 61, output_length=1)
    rs = np.random.RandomState(61)
    ids = rs.randint(0, 256, m.receptive_field)
    x = torch.zeros(1, 256, len(ids))
    x[0, ids, np.arange(len(ids))] = 1.
    y = m(x)[-1].detach().numpy()
    _, logits = c_oracle.generate(cfg, W, 1, ids, 0.0, 0.0)
    assert np.abs(y - logits[0]).max() < 1e-5


_REAL_ENGINE = engine.Engine


def _inject_double(m, monkeypatch):
    real = _REAL_ENGINE

    def make(cfg, weights, n_streams=1, device_index=0, **kw):
        return real(cfg, weights, n_streams=n_streams, device_index=device_index, **double_backend(), **kw)

    monkeypatch.setattr(engine, "Engine", make)


def test_generate_fast_contract_on_the_double(golden, monkeypatch):
    """Audio equals what the REAL reference returned for the same seed (tests/golden): same RNG consumption,
    same de-quantisation + mu-law expansion, float64 (num_samples,)."""
    for case, cname in (("tiny", "tiny"), ("tiny_bias", "tiny_bias"), ("cfg1_seed128", "cfg1")):
        wseed, n_given, n, npseed = [int(v) for v in golden["gen_%s_meta" % case]]
        temp, regz = [float(v) for v in golden["gen_%s_tr" % case]]
        m, cfg, W = _model(cname, wseed)
        _inject_double(m, monkeypatch)
        first = None if n_given == 1 else torch.from_numpy(golden["gen_%s_first" % case].astype(np.int64))
        np.random.seed(npseed)
        buf = io.StringIO()
        with redirect_stdout(buf):
            audio = m.generate_fast(n, first_samples=first, temperature=temp, regularize=regz)
        assert audio.dtype == np.float64 and audio.shape == (n,)
        assert np.array_equal(audio, golden["gen_%s_audio" % case])
        assert "one generating step does take approximately" in buf.getvalue()
        assert m.training  # generate_fast leaves the module in train() (:313)
        after = np.random.random_sample()
        np.random.seed(npseed)
        np.random.random_sample(n)
        assert after == np.random.random_sample()  # exactly n uniforms were consumed


def test_generate_fast_progress_callbacks_match_reference_cadence(monkeypatch):
    m, cfg, W = _model("tiny", 62)
    _inject_double(m, monkeypatch)
    first = torch.from_numpy(np.random.RandomState(62).randint(0, 256, 23))
    calls = []
    np.random.seed(9)
    a = m.generate_fast(57, first_samples=first, temperature=1.0, progress_callback=lambda s, t: calls.append((s, t)),
                        progress_interval=10)
    expect = [(i, 80) for i in range(22) if i % 10 == 0] + [(i + 23, 80) for i in range(57) if (i + 23) % 10 == 0]
    assert calls == expect
    np.random.seed(9)
    b = m.generate_fast(57, first_samples=first, temperature=1.0)
    assert np.array_equal(a, b)  # cutting the job at callbacks does not change the audio
    g1 = m.generate_fast(30, first_samples=first, temperature=0)
    idx, _ = c_oracle.generate(cfg, W, 30, first.numpy(), 0.0, 0.0)
    assert np.array_equal(g1, c_oracle.expand(idx))


def test_priming_callbacks_fire_once_each_after_batched_priming(monkeypatch):
    """generate_fast(0, first_samples=<102 samples>): 101 priming evaluations (batched: >= 64), callbacks at i % 50 == 0 -> 0, 50, 100,
    each exactly once (wavenet_model.py:259-269) -- the last priming evaluation (100) is also where the job ends."""
    m, cfg, W = _model("tiny", 68)
    _inject_double(m, monkeypatch)
    first = torch.from_numpy(np.random.RandomState(68).randint(0, 256, 102))
    calls = []
    out = m.generate_fast(0, first_samples=first, temperature=0, progress_callback=lambda s, t: calls.append((s, t)), progress_interval=50)
    assert out.shape == (0,)
    assert m._wn_last_prime_batched
    assert calls == [(0, 102), (50, 102), (100, 102)]


def test_generate_fast_multi_stream_extension(monkeypatch):
    m, cfg, W = _model("tiny_bias", 63)
    _inject_double(m, monkeypatch)
    first = torch.from_numpy(np.random.RandomState(63).randint(0, 256, (3, 9)))
    out = m.generate_fast(20, first_samples=first, temperature=0)
    assert out.shape == (3, 20)
    for s in range(3):
        idx, _ = c_oracle.generate(cfg, W, 20, first[s].numpy(), 0.0, 0.0)
        assert np.array_equal(out[s], c_oracle.expand(idx))


def test_weights_update_reaches_engine(monkeypatch):
    m, cfg, W = _model("tiny", 64)
    _inject_double(m, monkeypatch)
    a = m.generate_fast(20, temperature=0)
    with torch.no_grad():
        m.end_conv_2.bias.add_(torch.linspace(-1, 1, 256))
    b = m.generate_fast(20, temperature=0)
    W2 = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    idx, _ = c_oracle.generate(cfg, W2, 20, None, 0.0, 0.0)
    assert np.array_equal(b, c_oracle.expand(idx)) and not np.array_equal(a, b)


def test_pickle_snapshot_roundtrip(tmp_path, monkeypatch):
    """torch.save(model) / load_latest_model_from, the reference's checkpoint flow (wavenet_training.py:84-88)."""
    m, cfg, W = _model("tiny", 65)
    _inject_double(m, monkeypatch)
    m.generate_fast(5, temperature=0)  # engine exists now; must not break pickling
    torch.save(m, str(tmp_path / "snap_2026"))
    m2 = wavenet_model.load_latest_model_from(str(tmp_path), use_cuda=False)
    assert isinstance(m2, wavenet_model.WaveNetModel)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    assert pickle.loads(pickle.dumps(m))._wn_engine is None


def test_snapshot_pickled_by_the_reference_class_loads_and_generates(monkeypatch):
    """torch.save(model) by the REFERENCE's WaveNetModel (its only checkpoint format, wavenet_training.py:84-88) has no
    end_channels / bias / engine attributes in its __dict__ (wavenet_model.py:42-56) and its DilatedQueue objects carry
    plain data / in_pos / out_pos: unpickling into this class must still generate."""
    m, cfg, W = _model("tiny", 66)
    state = m.__getstate__()
    for k in ("end_channels", "bias", "_wn_engine", "_wn_engine_key", "_wn_forward_calls", "_wn_train_runner",
              "_wn_train_calls", "matrix_precision"):
        state.pop(k)
    ref_queues = []
    for q in state["dilated_queues"]:
        rq = wavenet_modules.DilatedQueue.__new__(wavenet_modules.DilatedQueue)
        rq.__setstate__(dict(in_pos=0, out_pos=0, num_deq=1, num_channels=q.num_channels, dilation=q.dilation,
                             max_length=q.max_length, data=torch.zeros(q.num_channels, q.max_length), dtype=torch.FloatTensor))
        ref_queues.append(rq)
    state["dilated_queues"] = ref_queues
    # ... and a snapshot of an OLD torch (the reference pins 0.3) lacks the hook / buffer bookkeeping newer nn.Modules expect:
    # nn.Module.__setstate__ back-fills it, so our __setstate__ must go through it (the first forward() dies otherwise)
    stripped = [k for k in ("_backward_pre_hooks", "_forward_hooks_with_kwargs", "_forward_hooks_always_called", "_forward_pre_hooks_with_kwargs",
                            "_state_dict_pre_hooks", "_load_state_dict_post_hooks", "_non_persistent_buffers_set") if k in state]
    assert stripped
    for k in stripped:
        state.pop(k)
    m2 = wavenet_model.WaveNetModel.__new__(wavenet_model.WaveNetModel)
    m2.__setstate__(state)
    assert m2.end_channels == cfg["end_channels"] and m2.bias is False and m2._wn_engine is None
    assert all(hasattr(m2, k) for k in stripped)
    x = torch.zeros(1, 256, m2.receptive_field + m2.output_length - 1)
    x[0, 128, :] = 1.0
    assert m2(x).shape == (m2.output_length, 256)   # forward() through nn.Module.__call__ (hooks machinery) works
    _inject_double(m2, monkeypatch)
    a = m2.generate_fast(20, temperature=0)
    idx, _ = c_oracle.generate(cfg, W, 20, None, 0.0, 0.0)
    assert np.array_equal(a, c_oracle.expand(idx))
    m3 = pickle.loads(pickle.dumps(m2))   # and our own pickles still round-trip, queues included
    assert m3.dilated_queues[1].data.shape == (cfg["residual_channels"], 3)


def test_dilated_queues_hold_the_final_state_after_generate_fast(monkeypatch):
    """wavenet_model.py:177-184: generation leaves model.dilated_queues in their final state.  Here the state is read back
    from the engine lazily (first attribute access)."""
    import restated
    m, cfg, W = _model("tiny", 67)
    _inject_double(m, monkeypatch)
    first = torch.from_numpy(np.random.RandomState(67).randint(0, 256, 12))
    m.generate_fast(30, first_samples=first, temperature=0)
    assert all(q._lazy is not None for q in m.dilated_queues)  # nothing downloaded yet
    r = restated.RestatedWaveNet(cfg, W)
    r.generate_fast(30, first_samples=first.numpy(), temperature=0.0, return_details=True)
    for q, rq in zip(m.dilated_queues, r.queues):
        assert (q.in_pos, q.out_pos) == (rq.in_pos, rq.out_pos)
        assert np.allclose(q.data.numpy(), rq.data.numpy(), rtol=0, atol=2e-6)
    m.dilated_queues[0].enqueue(torch.ones(cfg["residual_channels"]))  # still a working DilatedQueue
    assert float(m.dilated_queues[0].data[0, r.queues[0].in_pos]) == 1.0
    m.generate_fast(3, temperature=0)  # the next call resets (wavenet_model.py:250-251): 3 evaluations into a 2-slot ring
    assert [q.in_pos for q in m.dilated_queues] == [3 % q.max_length for q in m.dilated_queues]


def test_generate_raises_like_dead_code():
    with pytest.raises(NotImplementedError):
        wavenet_model.WaveNetModel(layers=2, blocks=1).generate(3)


def test_mu_law_helpers():
    import audio_data
    x = np.linspace(-1, 1, 1001)
    assert np.allclose(audio_data.mu_law_expansion(audio_data.mu_law_encoding(x, 256), 256), x, atol=1e-12)
    q = audio_data.quantize_data(x, 256)
    assert q.min() == 0 and q.max() == 255


@pytest.mark.parametrize("case", ["tiny_bias", "cfg2", "cfg3"])
def test_forward_loss_and_gradients_match_the_reference_golden(golden, case):
    """golden_v3.npz: the REAL reference's forward() -> F.cross_entropy -> backward() (wavenet_model.py:186-196, wavenet_training.py:64-72)
    on a seeded batch, for BASELINE configs[1] and the 10 x 5 / 128 / 128 / 512 stack: the facade's torch path (what every GPU
    forward / gradient test uses as its checker) reproduces logits, loss and every parameter gradient."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import digest as dg
    wseed, N, out_len = [int(v) for v in golden["grad_%s_meta" % case]]
    cfg = synth.CONFIGS[case]
    m = wavenet_model.WaveNetModel(output_length=out_len, **cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.init_weights(cfg, seed=wseed).items()})
    ids = torch.from_numpy(golden["grad_%s_ids" % case].astype(np.int64))
    x = torch.zeros(N, 256, ids.shape[1]).scatter_(1, ids.view(N, 1, -1), 1.0)
    y = m(x)
    ref = golden["grad_%s_out" % case]
    assert y.shape == ref.shape and float((y.detach() - torch.from_numpy(ref)).abs().max()) <= 1e-6 * max(1.0, float(np.abs(ref).max()))
    loss = torch.nn.functional.cross_entropy(y, torch.from_numpy(golden["grad_%s_target" % case].astype(np.int64)))
    assert abs(float(loss.detach()) - float(golden["grad_%s_loss" % case][0])) <= 1e-6
    loss.backward()
    got = dg.digest({k: (p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)) for k, p in m.named_parameters()})
    want = {k: golden["grad_%s_d_%s" % (case, k)] for k in got}
    worst = dg.compare(want, got, 1e-5)
    print(case, "worst gradient digest deviation", worst)
