"""CPU-side (-m "not gpu") parity of the engine's HOST logic and KERNEL ARITHMETIC: the same csrc/ sources
compiled with -DWN_EMU (tests/emu/build_emu.py) against the C oracle.  Covers planner, weight packer,
C-ABI argument handling, chain split (P, PA), multi-stream indexing, priming, continuation, queue export.
What it cannot cover (memory model, LDS limits, real concurrency) is covered by tests/test_gpu_parity.py."""
import numpy as np
import pytest

import c_oracle
import restated
from emu_lib import emu_library
from mi355_wavenet import engine, synth
from parity_common import check_engine, make_case

ODD = dict(layers=4, blocks=2, dilation_channels=10, residual_channels=7, skip_channels=13, end_channels=9,
           classes=256, kernel_size=2, bias=True)
K3 = dict(synth.CONFIGS["tiny_bias"], kernel_size=3)
C64 = dict(layers=3, blocks=1, dilation_channels=8, residual_channels=8, skip_channels=16, end_channels=16,
           classes=64, kernel_size=2, bias=False)
C300 = dict(layers=2, blocks=2, dilation_channels=6, residual_channels=6, skip_channels=8, end_channels=8,
            classes=300, kernel_size=2, bias=True)

CASES = [
    ("tiny", "tiny", dict(), 1), ("tiny_bias", "tiny_bias", dict(), 2),
    ("tiny_bias_split", "tiny_bias", dict(layer_split=3, head_split=5), 1),
    ("cfg1", "cfg1", dict(), 1), ("cfg1_split", "cfg1", dict(layer_split=2, head_split=8), 2),
    ("odd", ODD, dict(layer_split=3, head_split=2), 2), ("k3", K3, dict(layer_split=2, head_split=2), 1),
    ("c64", C64, dict(), 1), ("c300", C300, dict(head_split=2), 1),
]


@pytest.mark.parametrize("label,cfgname,kw,ns", CASES, ids=[c[0] for c in CASES])
def test_emulated_chain_matches_oracle(label, cfgname, kw, ns):
    N, n_given = 120, 25
    cfg, W, first, uniforms = make_case(cfgname, 41, ns, n_given, N)
    eng = engine.Engine(cfg, W, n_streams=ns, lib=emu_library(), **kw)
    check_engine(eng, cfg, W, N, first, 0.0, 0.0, None, label + " greedy")
    check_engine(eng, cfg, W, N, first, 0.85, 0.0015, uniforms, label + " sampled")
    eng.close()


def test_default_first_sample_and_no_priming():
    cfg, W, _, uniforms = make_case("tiny", 42, 1, 1, 80)
    eng = engine.Engine(cfg, W, lib=emu_library())
    idx = eng.generate(80, None, temperature=1.0, uniforms=uniforms)
    o_idx, _ = c_oracle.generate(cfg, W, 80, None, 1.0, 0.0, uniforms[0])  # first_samples=None -> [classes//2]
    assert np.array_equal(idx[0], o_idx)


def test_continuation_equals_one_shot():
    """generate(N) == generate(a) then generate(N-a, first=[last], reset=False): how the facade implements
    progress callbacks (wavenet_model.py:308-311) with one launch per interval."""
    cfg, W, first, uniforms = make_case("tiny_bias", 43, 2, 9, 90)
    eng = engine.Engine(cfg, W, n_streams=2, lib=emu_library(), layer_split=2)
    full = eng.generate(90, first, temperature=1.0, uniforms=uniforms)
    a = eng.generate(37, first, temperature=1.0, uniforms=uniforms[:, :37])
    b = eng.generate(53, a[:, -1:], temperature=1.0, uniforms=uniforms[:, 37:], reset=False)
    assert np.array_equal(np.concatenate([a, b], axis=1), full)
    assert eng.info()["evals_done"] == 9 - 1 + 90


def test_export_queue_matches_reference_queue_layout():
    """wn_export_queue reproduces DilatedQueue.data / in_pos / out_pos (wavenet_modules.py:43-57) after the
    same pushes, checked against the torch restatement of the reference queue."""
    cfg, W, first, _ = make_case("tiny", 44, 1, 12, 30)
    eng = engine.Engine(cfg, W, lib=emu_library())
    idx = eng.generate(30, first, temperature=0.0)
    r = restated.RestatedWaveNet(cfg, W)
    _, ridx, _ = r.generate_fast(30, first_samples=first[0], temperature=0.0, return_details=True)
    assert np.array_equal(idx[0], ridx)
    for layer in range(cfg["layers"] * cfg["blocks"]):
        data, ip, op = eng.export_queue(layer)
        q = r.queues[layer]
        assert (ip, op) == (q.in_pos, q.out_pos)
        assert np.allclose(data, q.data.numpy(), rtol=0, atol=2e-6)


def test_zero_samples_and_prime_only():
    cfg, W, first, _ = make_case("tiny", 45, 1, 6, 1)
    eng = engine.Engine(cfg, W, lib=emu_library())
    idx = eng.generate(0, first, temperature=0.0)
    assert idx.shape == (1, 0)
    assert eng.info()["evals_done"] == 5
