"""-m gpu: WaveNetModel.generate_fast() drop-in on the real engine (CPU-resident module, like every reference
caller: generate_script.py:6, train_script.py:48) reproduces the REAL reference's golden audio."""
import io
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

import c_oracle
import wavenet_model
from mi355_wavenet import synth

pytestmark = pytest.mark.gpu


def _model(cname, seed, **kw):
    cfg = synth.CONFIGS[cname]
    W = synth.init_weights(cfg, seed=seed)
    m = wavenet_model.WaveNetModel(**dict(cfg, **kw))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    return m, cfg, W


def test_generate_fast_reproduces_reference_golden(golden):
    for case, cname in (("tiny", "tiny"), ("tiny_bias", "tiny_bias"), ("cfg1", "cfg1"), ("cfg1_seed128", "cfg1"),
                        ("cfg2", "cfg2"), ("cfg3", "cfg3")):
        wseed, n_given, n, npseed = [int(v) for v in golden["gen_%s_meta" % case]]
        temp, regz = [float(v) for v in golden["gen_%s_tr" % case]]
        m, cfg, W = _model(cname, wseed)
        first = None if n_given == 1 else torch.from_numpy(golden["gen_%s_first" % case].astype(np.int64))
        np.random.seed(npseed)
        with redirect_stdout(io.StringIO()):
            audio = m.generate_fast(n, first_samples=first, temperature=temp, regularize=regz)
        assert audio.dtype == np.float64 and audio.shape == (n,)
        assert np.array_equal(audio, golden["gen_%s_audio" % case]), case


def test_generate_fast_callbacks_and_cuda_module():
    m, cfg, W = _model("cfg1", 71)
    m = m.cuda()
    first = torch.from_numpy(np.random.RandomState(71).randint(0, 256, 64))
    calls = []
    np.random.seed(4)
    with redirect_stdout(io.StringIO()):
        a = m.generate_fast(350, first_samples=first, temperature=1.0, progress_callback=lambda s, t: calls.append((s, t)),
                            progress_interval=100)
    assert calls == [(0, 414)] + [(i + 64, 414) for i in range(350) if (i + 64) % 100 == 0]
    np.random.seed(4)
    idx, _ = c_oracle.generate(cfg, W, 350, first.numpy(), 1.0, 0.0, np.random.random_sample(350))
    assert np.array_equal(a, c_oracle.expand(idx))


def test_generate_script_shape_batched_priming_behind_the_drop_in():
    """/root/reference/generate_script.py:19-33: first_samples = one dataset item (receptive_field + output_length - 1 =
    5116 samples at cfg3), a progress callback every 1000 steps.  The 5115 priming evaluations run as ONE batched pass
    (wn_prime) behind generate_fast(); audio equals the oracle's, the callback list equals the reference's, the final
    queue state is readable from model.dilated_queues like upstream."""
    import time
    m, cfg, W = _model("cfg3", 72)
    rs = np.random.RandomState(72)
    first = torch.from_numpy(rs.randint(0, 256, 5116))
    N = 900
    calls = []
    np.random.seed(5)
    with redirect_stdout(io.StringIO()):
        m.generate_fast(1, first_samples=first[:100], temperature=1.0)  # builds the engine (weight upload) outside the stopwatch
    np.random.seed(5)
    t0 = time.perf_counter()
    with redirect_stdout(io.StringIO()):
        a = m.generate_fast(N, first_samples=first, temperature=1.0, progress_callback=lambda s, t: calls.append((s, t)),
                            progress_interval=1000)
    wall = time.perf_counter() - t0
    assert m._wn_last_prime_batched
    total = 5116 + N
    assert calls == [(i, total) for i in range(5115) if i % 1000 == 0] + [(i + 5116, total) for i in range(N) if (i + 5116) % 1000 == 0]
    np.random.seed(5)
    idx, _ = c_oracle.generate(cfg, W, N, first.numpy(), 1.0, 0.0, np.random.random_sample(N))
    assert np.array_equal(a, c_oracle.expand(idx))
    # priming cost: the engine alone, stopwatch around wn_prime
    eng = m._engine(1)
    eng.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    assert eng.prime_host(first.numpy()[None, :5115])
    torch.cuda.synchronize()
    prime_ms = (time.perf_counter() - t0) * 1e3
    print("generate_script shape: %d given + %d generated in %.1f ms wall; batched priming of 5115 samples %.2f ms" % (5116, N, wall * 1e3, prime_ms))
    assert prime_ms <= 10.0
    # queues after generation (wavenet_model.py:177-184): layer 9 (d = 512) holds the last 513 inputs of that layer
    np.random.seed(5)
    m.generate_fast(N, first_samples=first, temperature=1.0)
    q = m.dilated_queues[9]
    assert q.data.shape == (128, 513) and q.in_pos == q.out_pos == (5115 + N) % 513
    assert float(q.data.abs().max()) > 0


def test_queue_state_after_generation_matches_the_reference_queues():
    import restated
    m, cfg, W = _model("tiny", 73)
    first = torch.from_numpy(np.random.RandomState(73).randint(0, 256, 12))
    m.generate_fast(40, first_samples=first, temperature=0)
    r = restated.RestatedWaveNet(cfg, W)
    r.generate_fast(40, first_samples=first.numpy(), temperature=0.0, return_details=True)
    for q, rq in zip(m.dilated_queues, r.queues):
        assert (q.in_pos, q.out_pos) == (rq.in_pos, rq.out_pos)
        assert np.allclose(q.data.numpy(), rq.data.numpy(), rtol=0, atol=2e-6)


def test_forward_on_gpu_matches_golden(golden):
    wseed, N, out_len = [int(v) for v in golden["fwd_cfg1_meta"]]
    m, cfg, W = _model("cfg1", wseed, output_length=out_len)
    m = m.cuda()
    ids = torch.from_numpy(golden["fwd_cfg1_ids"].astype(np.int64))
    L = ids.shape[1]
    x = torch.zeros(N, 256, L).scatter_(1, ids.view(N, 1, L), 1.).cuda()
    y = m(x).cpu().detach().numpy()
    ref = golden["fwd_cfg1_out"]
    assert np.abs(y - ref).max() <= 1e-4  # fp32 GPU convs vs the reference's CPU fp32


def test_generate_fast_with_128_classes():
    """The drop-in on a 128-class model (the reference's older checkpoints): default seed classes // 2 = 64, indices in [0, 128), the
    de-quantisation o = idx / 128 * 2 - 1 and mu_law_expansion with mu = 128 (wavenet_model.py:246,296,314) -- audio equals the oracle's."""
    cfg = dict(synth.CONFIGS["cfg1"], classes=128)
    W = synth.init_weights(cfg, seed=73)
    m = wavenet_model.WaveNetModel(**cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    for first in (None, torch.from_numpy(np.random.RandomState(73).randint(0, 128, 90))):
        np.random.seed(6)
        with redirect_stdout(io.StringIO()):
            a = m.generate_fast(300, first_samples=first, temperature=0.9)
        np.random.seed(6)
        f = np.array([64]) if first is None else first.numpy()
        idx, _ = c_oracle.generate(cfg, W, 300, f, 0.9, 0.0, np.random.random_sample(300))
        assert idx.max() < 128 and np.array_equal(a, c_oracle.expand(idx, 128))
        assert np.abs(a).max() <= 1.0
