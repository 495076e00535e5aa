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
    for case, cname in (("tiny", "tiny"), ("tiny_bias", "tiny_bias"), ("cfg1", "cfg1"), ("cfg1_seed128", "cfg1")):
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
