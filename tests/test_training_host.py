"""Host side of the training rows on CPU: the dataset item path against golden items
produced by the real reference, device batches, the trainer loop against a hand-rolled copy of the reference's loop
on the reference's own model, the logger cadence, per-stream temperatures through engine.py (test double), and the
data-parallel gradient exchange over gloo with world_size 2."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))

import audio_data  # noqa: E402
import model_logging  # noqa: E402
import wavenet_model  # noqa: E402
import wavenet_training  # noqa: E402
from make_golden_dataset import SETTINGS, files  # noqa: E402


@pytest.fixture()
def dataset_file(tmp_path):
    path = str(tmp_path / "ds.npz")
    np.savez(path, *files())
    return path


def test_dataset_items_equal_the_references(dataset_file):
    g = np.load(os.path.join(HERE, "golden", "golden_dataset_v1.npz"))
    for k, (il, tl, stride) in enumerate(SETTINGS):
        for train in (True, False):
            tag = "s%d_%s" % (k, "train" if train else "test")
            ds = audio_data.WavenetDataset(dataset_file, item_length=il, target_length=tl, train=train, test_stride=stride)
            assert len(ds) == int(g[tag + "_len"]), tag
            for row, i in enumerate(g[tag + "_ids"]):
                one_hot, target = ds[int(i)]
                assert one_hot.dtype == torch.float32 and one_hot.shape == (256, il)
                assert target.dtype == torch.int64 and target.shape == (1, tl)
                assert np.array_equal(one_hot.argmax(0).numpy(), g[tag + "_x"][row]), (tag, i)
                assert float(one_hot.sum()) == il
                assert np.array_equal(target.numpy().reshape(-1), g[tag + "_t"][row]), (tag, i)


@pytest.mark.reference
def test_dataset_equals_reference_class_everywhere(dataset_file):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_shim
    if not ref_shim.available():
        pytest.skip("reference tree not present")
    _, _, ad = ref_shim.load()
    for il, tl, stride in SETTINGS:
        for train in (True, False):
            ref = ad.WavenetDataset(dataset_file, item_length=il, target_length=tl, train=train, test_stride=stride)
            ds = audio_data.WavenetDataset(dataset_file, item_length=il, target_length=tl, train=train, test_stride=stride)
            assert len(ds) == len(ref) and ds.start_samples == ref.start_samples
            for i in range(len(ref)):
                a, b = ref[i], ds[i]
                assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    x = np.random.RandomState(0).uniform(-1, 1, 1000)
    assert np.array_equal(ad.quantize_data(x, 256), audio_data.quantize_data(x, 256))
    assert np.array_equal(ad.mu_law_expansion(x, 256), audio_data.mu_law_expansion(x, 256))


def test_device_batches_equal_dataset_items(dataset_file):
    ds = audio_data.WavenetDataset(dataset_file, item_length=40, target_length=8, test_stride=3)
    db = audio_data.DeviceBatches(ds, "cpu")
    ids = [0, 5, len(ds) - 1, 17]
    idx, target = db.batch(ids)
    assert idx.dtype == torch.int32 and idx.shape == (4, 40) and target.shape == (32,)
    for row, i in enumerate(ids):
        one_hot, t = ds[i]
        assert torch.equal(idx[row].long(), one_hot.argmax(0))
        assert torch.equal(target[row * 8:(row + 1) * 8], t.view(-1))
    seen = sum(b[0].shape[0] for b in db.epoch(7, shuffle=True, generator=torch.Generator().manual_seed(0)))
    assert seen == len(ds)
    with pytest.raises(IndexError):
        ds.item_indices(len(ds) + 10_000)
    # data parallel: the ranks' shards of one epoch are disjoint, equally long and together one permutation (minus the remainder)
    shards = []
    for rank in range(3):
        g = torch.Generator().manual_seed(7)
        shards.append(torch.cat([b[0][:, 0] * 0 + 1 for b in db.epoch(5, shuffle=True, generator=g, rank=rank, world=3)]).numel())
    assert len(set(shards)) == 1 and sum(shards) == len(ds) - len(ds) % 3
    picks = []
    for rank in range(3):
        g = torch.Generator().manual_seed(7)
        order = torch.randperm(len(ds), generator=g)
        picks.append(set(order[:len(ds) - len(ds) % 3][rank::3].tolist()))
    assert not (picks[0] & picks[1]) and not (picks[1] & picks[2]) and len(picks[0] | picks[1] | picks[2]) == len(ds) - len(ds) % 3


def test_dataset_from_wav_files(tmp_path):
    from scipy.io import wavfile
    rs = np.random.RandomState(3)
    wav = (rs.uniform(-0.9, 0.9, 4000) * 32767).astype(np.int16)
    os.makedirs(tmp_path / "audio")
    wavfile.write(str(tmp_path / "audio" / "a.wav"), 16000, wav)
    ds = audio_data.WavenetDataset(str(tmp_path / "new.npz"), item_length=50, target_length=10, file_location=str(tmp_path / "audio"))
    assert os.path.isfile(str(tmp_path / "new.npz")) and len(ds) > 0
    expect = audio_data.quantize_data(wav.astype(np.float32) / 32768.0, 256).astype(np.uint8)
    assert np.array_equal(ds.data["arr_0"], expect)
    try:
        import librosa
        have_librosa = hasattr(librosa, "load")  # oracle/ref_shim.py may have planted an empty stub module
    except ImportError:
        have_librosa = False
    if not have_librosa:
        wavfile.write(str(tmp_path / "audio" / "b.wav"), 8000, wav)
        with pytest.raises(RuntimeError, match="librosa"):
            audio_data.WavenetDataset(str(tmp_path / "new2.npz"), item_length=50, target_length=10, file_location=str(tmp_path / "audio"))


def _tiny_model(seed=0):
    torch.manual_seed(seed)
    return wavenet_model.WaveNetModel(layers=3, blocks=2, dilation_channels=8, residual_channels=8, skip_channels=16, end_channels=16,
                                      classes=256, output_length=8, kernel_size=2, bias=True)


def test_logger_cadence():
    events = []

    class T:
        def validate(self):
            events.append("validate")
            return 1.0, 0.5

    lg = model_logging.Logger(log_interval=2, validation_interval=4, generate_interval=3, trainer=T(),
                              generate_function=lambda step: events.append(("gen", step)))
    for step in range(1, 7):
        lg.log(step, 1.0)
        if lg.generate_function is not None and lg.generate_thread.is_alive():
            lg.generate_thread.join()
    assert events.count("validate") == 1 and ("gen", 3) in events and ("gen", 6) in events
    with pytest.raises(NotImplementedError):
        model_logging.TensorboardLogger()


@pytest.mark.reference
def test_trainer_steps_equal_the_reference_loop(dataset_file):
    """Three optimiser steps of WavenetTrainer.train against the reference's loop body (wavenet_training.py:64-77, its two
    legacy `.data[0]` reads dropped) run on the reference's own WaveNetModel with the same initial weights and the same
    shuffled DataLoader order."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_shim
    if not ref_shim.available():
        pytest.skip("reference tree not present")
    mdl, _, ad = ref_shim.load()
    m = _tiny_model()
    ref = mdl.WaveNetModel(layers=3, blocks=2, dilation_channels=8, residual_channels=8, skip_channels=16, end_channels=16,
                           classes=256, output_length=8, kernel_size=2, bias=True)
    ref.load_state_dict(m.state_dict())
    il = m.receptive_field + m.output_length - 1
    ds = audio_data.WavenetDataset(dataset_file, item_length=il, target_length=8, test_stride=5)
    rds = ad.WavenetDataset(dataset_file, item_length=il, target_length=8, test_stride=5)

    class Stop(Exception):
        pass

    class StopLogger(model_logging.Logger):
        def log(self, step, loss):
            losses.append(loss)
            if step == 3:
                raise Stop()

    losses = []
    tr = wavenet_training.WavenetTrainer(m, ds, lr=0.01, gradient_clipping=1.0, logger=StopLogger(), num_workers=0)
    torch.manual_seed(11)
    with pytest.raises(Stop):
        tr.train(batch_size=4, epochs=1)
    # the reference's loop
    opt = torch.optim.Adam(params=ref.parameters(), lr=0.01, weight_decay=0)
    ref.train()
    torch.manual_seed(11)
    loader = torch.utils.data.DataLoader(rds, batch_size=4, shuffle=True, num_workers=0, pin_memory=False)
    ref_losses = []
    for step, (x, target) in enumerate(iter(loader), 1):
        x = x.type(torch.FloatTensor)
        target = target.view(-1).type(torch.LongTensor)
        output = ref(x)
        loss = torch.nn.functional.cross_entropy(output.squeeze(), target.squeeze())
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        opt.step()
        ref_losses.append(loss.item())
        if step == 3:
            break
    assert losses == ref_losses
    for (k, a), (_, b) in zip(m.state_dict().items(), ref.state_dict().items()):
        assert torch.equal(a, b), k
    avg_loss, acc = tr.validate()
    assert np.isfinite(avg_loss) and 0.0 <= acc <= 1.0 and ds.train is True


def test_per_stream_temperatures_through_the_engine_wrapper():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    from double_lib import double_backend, double_library
    from mi355_wavenet import engine, synth
    cfg = synth.CONFIGS["tiny_bias"]
    W = synth.init_weights(cfg, seed=5)
    temps = [0.0, 1.0, 0.6, -1.0]
    rs = np.random.RandomState(9)
    first = rs.randint(0, 256, (4, 3))
    u = rs.random_sample((4, 30))
    eng = engine.Engine(cfg, W, n_streams=4, **double_backend())
    out = eng.generate(30, first, temperature=np.asarray(temps, dtype=np.float32), uniforms=u)
    eng.close()
    for s, t in enumerate(temps):
        idx, _ = c_oracle.generate(cfg, W, 30, first[s], t if t > 0 else 0.0, 0.0, u[s] if t > 0 else None)
        assert np.array_equal(out[s], idx), (s, t)


def _dp_worker(rank, world, port, q):
    for p in (os.path.join(ROOT, "pytorch-wavenet_amd"), HERE):
        sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    m = _tiny_model(seed=4)
    tr = wavenet_training.WavenetTrainer(m, dataset=None, optimizer=torch.optim.SGD, lr=0.05, gradient_clipping=0.5, num_workers=0)
    g = torch.Generator().manual_seed(21)
    L = m.receptive_field + m.output_length - 1
    for _ in range(2):
        idx = torch.randint(0, 256, (4, L), generator=g)
        target = torch.randint(0, 256, (4, m.output_length), generator=g)
        mine = slice(2 * rank, 2 * rank + 2)  # every rank owns half of the global batch
        x = torch.zeros(2, 256, L).scatter_(1, idx[mine].unsqueeze(1), 1.0)
        tr.train_step("onehot", x, target[mine].reshape(-1))
    q.put((rank, {k: v.numpy().copy() for k, v in m.state_dict().items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_world2_equals_single_process_on_the_global_batch():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 1000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=180) for _ in range(2))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    m = _tiny_model(seed=4)
    tr = wavenet_training.WavenetTrainer(m, dataset=None, optimizer=torch.optim.SGD, lr=0.05, gradient_clipping=0.5, num_workers=0)
    g = torch.Generator().manual_seed(21)
    L = m.receptive_field + m.output_length - 1
    for _ in range(2):
        idx = torch.randint(0, 256, (4, L), generator=g)
        target = torch.randint(0, 256, (4, m.output_length), generator=g)
        x = torch.zeros(4, 256, L).scatter_(1, idx.unsqueeze(1), 1.0)
        tr.train_step("onehot", x, target.reshape(-1))
    for k, v in m.state_dict().items():
        assert np.array_equal(got[0][k], got[1][k]), k          # the ranks stay in lock step
        assert np.allclose(got[0][k], v.numpy(), rtol=1e-5, atol=1e-7), k  # and follow the single-process run on the whole batch


def _dp_validate_worker(rank, world, port, q, ds_path):
    for p in (os.path.join(ROOT, "pytorch-wavenet_amd"), HERE):
        sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    m = _tiny_model(seed=4)
    il = m.receptive_field + m.output_length - 1
    ds = audio_data.WavenetDataset(ds_path, item_length=il, target_length=8, test_stride=5)
    seen = []

    class Stop(Exception):
        pass

    class Rec(model_logging.Logger):
        def log(self, step, loss):
            if step == 4:
                raise Stop()

    tr = wavenet_training.WavenetTrainer(m, ds, optimizer=torch.optim.SGD, lr=0.0, logger=Rec(), num_workers=0)
    orders = []
    for epoch_run in range(2):  # the sampler's permutation must change from epoch to epoch (set_epoch)
        tr.dataloader, tr._sampler = tr._loader(4, train=True)
        tr._sampler.set_epoch(epoch_run)
        orders.append(list(iter(tr._sampler))[:6])
    try:
        tr.train(batch_size=4, epochs=1)
    except Stop:
        pass
    res = tr.validate()
    q.put((rank, res, orders, len(ds), ds.train))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_validate_scores_the_test_split_once(dataset_file):
    """with a DistributedSampler the old validate() walked a train-sized, wrapped-around index list and divided
    by the full test length.  Now: a sampler over the TEST split, padding not scored, sums all-reduced -- both ranks return
    the figures of a single-process validate() on the same weights (lr = 0: the weights never move)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 1000
    procs = [ctx.Process(target=_dp_validate_worker, args=(r, 2, port, q, dataset_file)) for r in range(2)]
    [p.start() for p in procs]
    got = dict((r, rest) for r, *rest in (q.get(timeout=240) for _ in range(2)))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    m = _tiny_model(seed=4)
    il = m.receptive_field + m.output_length - 1
    ds = audio_data.WavenetDataset(dataset_file, item_length=il, target_length=8, test_stride=5)
    tr = wavenet_training.WavenetTrainer(m, ds, optimizer=torch.optim.SGD, lr=0.0, num_workers=0)
    tr.dataloader, tr._sampler = tr._loader(4, train=True)
    loss1, acc1 = tr.validate()
    for r in (0, 1):
        (loss, acc), orders, n_train, train_flag = got[r]
        assert train_flag is True and n_train == len(ds)
        assert 0.0 <= acc <= 1.0 and abs(acc - acc1) < 1e-12          # same correct-count / same target count
        assert orders[0] != orders[1]                                   # a new permutation per epoch
    assert got[0][0] == got[1][0]                                       # all-reduced: identical on both ranks
    # the per-batch mean of batch-mean losses depends on how the items fall into batches; bound it by the spread
    assert abs(got[0][0][0] - loss1) < 0.05 * max(1.0, abs(loss1))


def test_fused_adam_refuses_what_it_cannot_step_natively():
    """mi355_wavenet.optim.FusedAdam is the engine's optimiser kernel or nothing: CPU parameters raise (no silent torch fallback); its constructor
    and state layout are torch.optim.Adam's."""
    import torch
    from mi355_wavenet.optim import FusedAdam
    p = torch.nn.Parameter(torch.randn(7, 3))
    opt = FusedAdam([p], lr=1e-3, betas=(0.8, 0.9), weight_decay=0.1)
    assert opt.param_groups[0]["betas"] == (0.8, 0.9) and opt.param_groups[0]["weight_decay"] == 0.1
    opt.step()                                   # no gradient: nothing to do, like torch's optimisers
    p.grad = torch.ones_like(p)
    with pytest.raises(TypeError, match="MI355X"):
        opt.step()
    with pytest.raises(ValueError):
        FusedAdam([p], betas=(1.0, 0.9))
