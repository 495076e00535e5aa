"""Drop-in for the reference's ``wavenet_training`` (/root/reference/wavenet_training.py): ``WavenetTrainer`` and
``generate_audio``.

The loop is the reference's (:50-88): batch -> ``model(x)`` -> ``F.cross_entropy`` -> backward -> optional clip -> step ->
snapshot -> logger.  On an MI355X ``model(x)`` and ``loss.backward()`` run through the native matrix-core stack
(wavenet_model.WaveNetModel._native_forward, mi355_wavenet/training.py).  Extensions, all off by default:
  * ``device_batches=True``: batches are cut on the GPU from the resident class-index stream (audio_data.DeviceBatches)
    and enter the model as indices -- no host one-hot, no 256x inflated H2D copy;
  * ``optimizer=mi355_wavenet.optim.FusedAdam``: Adam's step and the gradient clipping in front of it as the engine's native kernels
    (the default stays ``optim.Adam``, like upstream: any torch optimiser class works);
  * ``process_group``: data-parallel training, one process per GPU: every rank steps on the average of all ranks'
    gradients (ONE flat all-reduce per step over RCCL; SURVEY.md section 8e "Training (cfg5): plain data parallel").
"""
import os
import time

import numpy as np
import torch
import torch.nn.functional as F
import torch.optim as optim
import torch.utils.data

from model_logging import Logger


def print_last_loss(opt):
    print("loss: ", opt.losses[-1])


def print_last_validation_result(opt):
    print("validation loss: ", opt.validation_results[-1])


def average_gradients(parameters, process_group=None, always=False):
    """Data-parallel gradient exchange: ONE all-reduce over the flattened gradients of ``parameters`` (30 MB at BASELINE
    config 5: latency-bound on xGMI, so a single bucket), then every .grad <- mean over ranks.  No-op without a group and
    in a group of one rank -- unless ``always`` (the collective then runs all the same: how a 1-GPU box exercises the RCCL
    call of the N-GPU path, tests/test_gpu_multi.py)."""
    import torch.distributed as dist
    if process_group is None and not (dist.is_available() and dist.is_initialized()):
        return
    world = dist.get_world_size(process_group)
    if world == 1 and not always:
        return
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=process_group)
    flat.div_(world)
    pos = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[pos:pos + n].view_as(g))
        pos += n


class WavenetTrainer:
    def __init__(self, model, dataset, optimizer=optim.Adam, lr=0.001, weight_decay=0, gradient_clipping=None,
                 logger=None, snapshot_path=None, snapshot_name='snapshot', snapshot_interval=1000,
                 dtype=torch.FloatTensor, ltype=torch.LongTensor, device_batches=False, process_group=None, num_workers=8):
        self.model = model
        self.dataset = dataset
        self.dataloader = None
        self.lr = lr
        self.weight_decay = weight_decay
        self.clip = gradient_clipping
        self.optimizer_type = optimizer
        self.optimizer = self.optimizer_type(params=self.model.parameters(), lr=self.lr, weight_decay=self.weight_decay)
        self.logger = logger if logger is not None else Logger()  # upstream shares ONE default Logger between trainers (:27)
        self.logger.trainer = self
        self.snapshot_path = snapshot_path
        self.snapshot_name = snapshot_name
        self.snapshot_interval = snapshot_interval
        self.dtype = dtype
        self.ltype = ltype
        self.device_batches = device_batches
        self.process_group = process_group
        self.num_workers = num_workers
        self._batches = None

    def _device(self):
        return next(self.model.parameters()).device

    def _to_model(self, x, target):
        dev = self._device()
        if isinstance(self.dtype, torch.dtype):
            x = x.to(dev, self.dtype)
        else:
            x = x.type(self.dtype).to(dev)
        target = target.view(-1).type(self.ltype).to(dev)
        return x, target

    def _rank_world(self):
        import torch.distributed as dist
        if self.process_group is None and not (dist.is_available() and dist.is_initialized()):
            return 0, 1
        return dist.get_rank(self.process_group), dist.get_world_size(self.process_group)

    def _epoch(self, batch_size, shuffle):
        """Yields (kind, x, target): kind "indices" (device batches) or "onehot" (the reference's DataLoader items).
        Data parallel: every rank sees a disjoint shard of the epoch (same permutation seed on all ranks)."""
        rank, world = self._rank_world()
        if self.device_batches:
            if self._batches is None or self._batches.device != self._device():
                from audio_data import DeviceBatches
                if getattr(self.dataset, "classes", 0) > getattr(self.model, "classes", 1 << 30):
                    raise ValueError("dataset quantised to %d classes, model has %d" % (self.dataset.classes, self.model.classes))
                self._batches = DeviceBatches(self.dataset, self._device())
            gen = None
            if world > 1:
                self._epoch_count = getattr(self, "_epoch_count", 0) + 1
                gen = torch.Generator().manual_seed(1234 + self._epoch_count)
            for idx, target in self._batches.epoch(batch_size, shuffle=shuffle, generator=gen, rank=rank, world=world):
                yield "indices", idx, target
        else:
            for x, target in iter(self.dataloader):
                x, target = self._to_model(x, target)
                yield "onehot", x, target

    def _forward(self, kind, x):
        if kind == "indices":  # cut from the dataset's own quantised stream (DeviceBatches): in range by construction
            return self.model.train_forward_indices(x, check=False) if torch.is_grad_enabled() else self.model.forward_indices(x, check=False)
        return self.model(x)

    def _loss(self, output, target):
        """F.cross_entropy(output.squeeze(), target.squeeze()) (wavenet_training.py:69) -- on the engine when the logits came from it
        (fp32, 256 classes, on the GPU): loss and dLoss/dlogits in one pass (wn_train_loss).  WN_TORCH_LOSS=1 pins torch's."""
        out2, tgt = output.squeeze(), target.squeeze()
        runner = getattr(self.model, "_wn_train_runner", None)
        if (runner is not None and out2.is_cuda and out2.dim() == 2 and out2.size(1) == 256 and out2.dtype == torch.float32
                and tgt.dim() == 1 and os.environ.get("WN_TORCH_LOSS") != "1"):
            from mi355_wavenet import training
            return training.cross_entropy(runner, out2, tgt)
        return F.cross_entropy(out2, tgt)

    def train_step(self, kind, x, target):
        """One optimiser step (wavenet_training.py:68-77); returns the loss as a float."""
        output = self._forward(kind, x)
        loss = self._loss(output, target)
        self.optimizer.zero_grad()
        loss.backward()
        average_gradients(self.model.parameters(), self.process_group)
        if self._fused_optimizer():   # the engine's optimiser kernels: clip_grad_norm folded into the step (mi355_wavenet/optim.py)
            self.optimizer.step(max_grad_norm=self.clip)
        else:
            if self.clip is not None:
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip)
            self.optimizer.step()
        return loss.item()

    def _fused_optimizer(self):
        from mi355_wavenet.optim import FusedAdam
        return isinstance(self.optimizer, FusedAdam)

    def _loader(self, batch_size, train):
        """DataLoader over the split selected by ``dataset.train``; data parallel: a DistributedSampler over THAT split
        (the dataset's length depends on the split, so the sampler is built per split and re-seeded every epoch)."""
        rank, world = self._rank_world()
        sampler = None
        if world > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(self.dataset, num_replicas=world, rank=rank, shuffle=train,
                                                                      drop_last=False)
        loader = torch.utils.data.DataLoader(self.dataset, batch_size=batch_size, shuffle=train and sampler is None, sampler=sampler,
                                             num_workers=self.num_workers if train else 0, pin_memory=False)
        return loader, sampler

    def train(self, batch_size=32, epochs=10, continue_training_at_step=0):
        self.model.train()
        self.dataloader, self._sampler = self._loader(batch_size, train=True)
        step = continue_training_at_step
        for current_epoch in range(epochs):
            print("epoch", current_epoch)
            if self._sampler is not None:
                self._sampler.set_epoch(current_epoch)  # a new permutation every epoch, the same one on every rank
            tic = time.time()
            for kind, x, target in self._epoch(batch_size, shuffle=True):
                loss = self.train_step(kind, x, target)
                step += 1
                if step == 100:
                    toc = time.time()
                    print("one training step does take approximately " + str((toc - tic) * 0.01) + " seconds)")
                if step % self.snapshot_interval == 0:
                    if self.snapshot_path is None:
                        continue
                    time_string = time.strftime("%Y-%m-%d_%H-%M-%S", time.gmtime())
                    torch.save(self.model, self.snapshot_path + '/' + self.snapshot_name + '_' + time_string)
                self.logger.log(step, loss)

    def validate(self):  # :89-112
        """(average loss per batch, accuracy over the test split).  Data parallel: every rank scores a disjoint shard of the
        test split (device batches: DeviceBatches.epoch(rank, world); DataLoader path: a sampler over the TEST split, whose
        wrap-around padding is not counted) and the sums are all-reduced, so every rank returns the global figures."""
        self.model.eval()
        self.dataset.train = False
        rank, world = self._rank_world()
        train_loader = self.dataloader
        batch_size = train_loader.batch_size if train_loader is not None else 32
        n_items = len(self.dataset)
        budget = None
        if not self.device_batches:
            self.dataloader, _ = self._loader(batch_size, train=False)
            if world > 1:  # the sampler pads every rank to ceil(n/world) items by wrapping around: score only the real ones
                budget = len(range(rank, n_items, world))
        total_loss = 0.0
        accurate_classifications = 0
        n_batches = 0
        n_targets = 0
        with torch.no_grad():
            for kind, x, target in self._epoch(batch_size, shuffle=False):
                if budget is not None:
                    keep = min(budget, x.size(0))
                    if keep == 0:
                        break
                    budget -= keep
                    x, target = x[:keep], target.view(x.size(0), -1)[:keep].reshape(-1)
                output = self._forward(kind, x)
                loss = F.cross_entropy(output.squeeze(), target.squeeze())
                total_loss += loss.item()
                predictions = torch.max(output, 1)[1].view(-1)
                accurate_classifications += torch.sum(torch.eq(target.view(-1), predictions)).item()
                n_batches += 1
                n_targets += target.numel()
        if world > 1:
            import torch.distributed as dist
            dev = self._device() if dist.get_backend(self.process_group) == "nccl" else torch.device("cpu")
            sums = torch.tensor([total_loss, float(n_batches), float(accurate_classifications), float(n_targets)], dtype=torch.float64, device=dev)
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.process_group)
            total_loss, n_batches, accurate_classifications, n_targets = (float(v) for v in sums.tolist())
        avg_loss = total_loss / max(n_batches, 1)
        # upstream divides by len(dataset) * target_length (:109); that is n_targets whenever the whole split was scored
        avg_accuracy = accurate_classifications / max(n_targets, 1)
        self.dataset.train = True
        self.dataloader = train_loader
        self.model.train()
        return avg_loss, avg_accuracy


def generate_audio(model, length=8000, temperatures=[0., 1.]):
    """:115-124.  The reference generates once per temperature, one after the other; the temperatures are independent
    streams, and on the MI355X engine all of them run in ONE persistent-kernel job (model.generate_fast_streams)."""
    if hasattr(model, "generate_fast_streams"):
        return model.generate_fast_streams(length, temperatures=list(temperatures))
    samples = [model.generate_fast(length, temperature=temp) for temp in temperatures]
    return np.stack(samples, axis=0)
