"""CPU-side (-m "not gpu") tests of the HOST logic between the Python facade and the C ABI -- mi355_wavenet/engine.py:
argument shaping, default first sample, continuation without reset, zero-sample jobs, batched priming hand-over, queue
export -- against the C oracle, with the host-memory test double of include/wn_abi.h (tests/double) standing in for the
device.  The double IS the oracle behind the ABI, so nothing here says anything about the HIP kernels: those are checked on
the GPU (tests/test_gpu_parity.py)."""
import os
import sys

import numpy as np

import c_oracle
import restated
from double_lib import double_backend, double_library
from mi355_wavenet import engine
from parity_common import check_engine, make_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_engine_wrapper_round_trip_multi_stream():
    cfg, W, first, uniforms = make_case("tiny_bias", 41, 3, 25, 60)
    eng = engine.Engine(cfg, W, n_streams=3, **double_backend())
    check_engine(eng, cfg, W, 60, first, 0.0, 0.0, None, "double greedy")
    check_engine(eng, cfg, W, 60, first, 0.85, 0.0015, uniforms, "double sampled")
    eng.close()


def test_default_first_sample_and_no_priming():
    cfg, W, _, uniforms = make_case("tiny", 42, 1, 1, 80)
    eng = engine.Engine(cfg, W, **double_backend())
    idx = eng.generate(80, None, temperature=1.0, uniforms=uniforms)
    o_idx, _ = c_oracle.generate(cfg, W, 80, None, 1.0, 0.0, uniforms[0])  # first_samples=None -> [classes//2]
    assert np.array_equal(idx[0], o_idx)


def test_continuation_equals_one_shot():
    """generate(N) == generate(a) then generate(N-a, first=[last], reset=False): how the facade implements
    progress callbacks (wavenet_model.py:308-311) with one launch per interval."""
    cfg, W, first, uniforms = make_case("tiny_bias", 43, 2, 9, 90)
    eng = engine.Engine(cfg, W, n_streams=2, **double_backend())
    full = eng.generate(90, first, temperature=1.0, uniforms=uniforms)
    a = eng.generate(37, first, temperature=1.0, uniforms=uniforms[:, :37])
    b = eng.generate(53, a[:, -1:], temperature=1.0, uniforms=uniforms[:, 37:], reset=False)
    assert np.array_equal(np.concatenate([a, b], axis=1), full)
    assert eng.info()["evals_done"] == 9 - 1 + 90


def test_batched_priming_hand_over():
    """Engine.generate primes long given windows through wn_prime and continues with n_given = 1 from the last given
    sample: same indices as per-sample priming."""
    cfg, W, first, uniforms = make_case("tiny", 46, 2, engine.Engine.PRIME_BATCH_MIN + 10, 30)
    eng = engine.Engine(cfg, W, n_streams=2, **double_backend())
    a = eng.generate(30, first, temperature=1.0, uniforms=uniforms, batched_prime=True)
    b = eng.generate(30, first, temperature=1.0, uniforms=uniforms, batched_prime=False)
    assert np.array_equal(a, b)
    o_idx, _ = c_oracle.generate(cfg, W, 30, first[1], 1.0, 0.0, uniforms[1])
    assert np.array_equal(a[1], o_idx)


def test_export_queue_matches_reference_queue_layout():
    """wn_export_queue hands out DilatedQueue.data / in_pos / out_pos (wavenet_modules.py:43-57) after the same pushes,
    checked against the torch restatement of the reference queue."""
    cfg, W, first, _ = make_case("tiny", 44, 1, 12, 30)
    eng = engine.Engine(cfg, W, **double_backend())
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
    eng = engine.Engine(cfg, W, **double_backend())
    idx = eng.generate(0, first, temperature=0.0)
    assert idx.shape == (1, 0)
    assert eng.info()["evals_done"] == 5


def test_engine_passes_the_no_padding_flag_and_the_abi_rejects_other_reserved_bits():
    """Engine(pad_channels=False) sets wn_config.reserved[0] = WN_CFG_NO_PADDING (a handle that serves wn_train_* keeps the model's own
    channel shape: include/wn_abi.h); every other reserved bit is an argument error -- the same rule in the double as in the product."""
    import ctypes
    from mi355_wavenet import _abi, synth
    cfg = synth.CONFIGS["tiny"]
    W = synth.init_weights(cfg, seed=3)
    seen = []
    be = double_backend()
    lib = be["lib"]
    real = lib.dll.wn_create

    class Spy:
        def __init__(self, dll):
            self._dll = dll

        def __getattr__(self, name):
            if name == "wn_create":
                def create(cfg_ref, out):
                    seen.append(int(ctypes.cast(cfg_ref, ctypes.POINTER(_abi.wn_config)).contents.reserved[0]))
                    return real(cfg_ref, out)
                return create
            return getattr(self._dll, name)

    dll = lib.dll
    lib.dll = Spy(dll)
    try:
        engine.Engine(cfg, W, **be).close()
        engine.Engine(cfg, W, pad_channels=False, **be).close()
    finally:
        lib.dll = dll
    assert seen == [0, 1]
    h = ctypes.c_void_p()
    for bad in ((2, 0, 0), (0, 1, 0), (1, 0, 7)):
        c = _abi.wn_config(cfg["layers"], cfg["blocks"], cfg["dilation_channels"], cfg["residual_channels"], cfg["skip_channels"],
                           cfg["end_channels"], 256, 2, 0, 1, 0, 0, 0)
        c.reserved[0], c.reserved[1], c.reserved[2] = bad
        assert dll.wn_create(ctypes.byref(c), ctypes.byref(h)) == _abi.WN_E_BADARG


def test_bench_reducer_on_fake_eight_rank_stats():
    """bench.py's reduction of an N-rank run (the first multi-GPU line must explain itself).  Fed fake 8-rank records: weak scaling =
    64 streams on every rank, strong = BASELINE configs[3]'s 512 streams sharded; `value` is the whole job against the SLOWEST rank's wall clock,
    `value_per_gpu` the metric's per-GPU figure, and every rank reports its own rate (a straggler is visible)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import json
    assert bench.BASELINE_METRIC == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    samples, steps = 16000, 5
    ms = [930.0, 931.0, 929.5, 930.2, 990.0, 930.1, 930.3, 929.9]   # rank 4 is a straggler
    per_rank = [{"rank": r, "kernel_ms": ms[r] - 6.0, "gather_ms": 1.0, "facade_ms": ms[r]} for r in (3, 0, 7, 1, 2, 6, 5, 4)]   # (any order)
    fac_wall = max(ms) * steps * 1e-3
    eng_wall = (max(ms) - 5.0) * steps * 1e-3
    weak = bench.reduce_ranks(per_rank, bench.streams_by_rank("weak", 64, 8), samples, steps, fac_wall, eng_wall)
    assert weak["total_streams"] == 512
    assert abs(weak["value"] - 512 * samples / 0.990) < 1e-6 * weak["value"]
    assert abs(weak["value_per_gpu"] * 8 - weak["value"]) < 1e-9 * weak["value"]
    assert [r["rank"] for r in weak["per_rank"]] == list(range(8)) and all(r["streams"] == 64 for r in weak["per_rank"])
    assert abs(weak["per_rank"][0]["samples_per_s"] - 64 * samples / 0.930) < 1.0
    assert weak["per_rank"][4]["samples_per_s"] < 0.95 * weak["per_rank"][0]["samples_per_s"]      # the straggler shows
    assert sum(r["samples_per_s"] for r in weak["per_rank"]) > weak["value"]                       # ... and is what bounds the job
    strong = bench.reduce_ranks(per_rank, bench.streams_by_rank("strong", 64, 8), samples, steps, fac_wall, eng_wall)
    assert strong["total_streams"] == 512 and [r["streams"] for r in strong["per_rank"]] == [64] * 8
    assert bench.streams_by_rank("strong", 64, 3) == [171, 171, 170] or sum(bench.streams_by_rank("strong", 64, 3)) == 512
    one = bench.reduce_ranks([{"rank": 0, "kernel_ms": 922.0, "gather_ms": 0.0, "facade_ms": 928.0}], [64], samples, 20, 0.928 * 20, 0.922 * 20)
    assert abs(one["value"] - one["value_per_gpu"]) < 1e-9 and abs(one["value"] - 64 * samples / 0.928) < 1e-3


def test_bench_reads_the_pmc_counters_of_the_one_generation_kernel(tmp_path):
    """bench.py's live roofline.traffic (round 6): the mean counter value of the ONE generation kernel in a rocprofv3 results database, scaled to the
    launch's timesteps with the calibrated FETCH_SIZE correction -- on a hand-made table; two generation kernels (a rounds job) or none are refused."""
    import sqlite3
    import bench
    db = str(tmp_path / "r.db")
    con = sqlite3.connect(db)
    con.execute("create table counters_collection (kernel_name text, counter_name text, value real)")
    k = "void wn_generate_kernel_v3m<128, 32, 512, 32, 4, 2, 0>(WnPlan, WnRun)"
    con.executemany("insert into counters_collection values (?, ?, ?)",
                    [(k, "FETCH_SIZE", 72.0e6), (k, "FETCH_SIZE", 72.4e6), ("__amd_rocclr_fillBufferAligned", "FETCH_SIZE", 11.7), (k, "WRITE_SIZE", 142.5e6)])
    con.commit()
    name, fetch, n = bench.pmc_counter_from_db(db, "FETCH_SIZE")
    assert name == k and n == 2 and abs(fetch - 72.2e6) < 1.0
    _, write, _ = bench.pmc_counter_from_db(db, "WRITE_SIZE")
    assert bench.pmc_traffic_bytes(fetch, write, 16000, 2000) == int((2.0 * 72.2e6 + 142.5e6) * 1024 * 8)
    assert bench.pmc_counter_from_db(db, "TCC_HIT_sum") is None
    con.execute("insert into counters_collection values (?, ?, ?)", ("void wn_generate_kernel_v4<64, 64, 256, 64, 3>(WnPlan, WnRun)", "WRITE_SIZE", 5.0))
    con.commit()
    assert bench.pmc_counter_from_db(db, "WRITE_SIZE") is None
