"""TEST INFRASTRUCTURE ONLY -- torch (CPU) restatement of the reference's generation path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file; the product never does (it fails loudly when the HIP library is missing).

What it restates, op for op (same ATen ops in the same order, so its CPU timing is the
reference's CPU timing and its numerics are the reference's numerics):

  * ``WaveNetModel.wavenet()`` with ``queue_dilate``     /root/reference/wavenet_model.py:125-184
  * ``DilatedQueue.enqueue/dequeue/reset``               /root/reference/wavenet_modules.py:42-77
  * ``WaveNetModel.generate_fast()``                     /root/reference/wavenet_model.py:237-315
  * ``mu_law_expansion``                                 /root/reference/audio_data.py:156-158

Differences from the reference, all deliberate:
  * the greedy branch (wavenet_model.py:292 ``torch.max(x, 0)[1][0]``) raises IndexError on
    torch>=0.4; it is restated as ``torch.max(x, 0)[1].reshape(1)`` (first index of the maximum);
  * ``DilatedQueue.enqueue`` flattens its (R,1) input (same patch as oracle/ref_shim.py);
  * per-step logits and integer indices can be returned for differential tests.

Pinning: tests/test_oracle_pinning.py proves this file == the shimmed reference itself
(identical float64 audio for identical seeds on the sampled branch, bit-equal logits) in the
authoring container, and == the committed fixtures in tests/golden/ everywhere.
"""
import time

import numpy as np
import torch
import torch.nn.functional as F


class Queue:
    """wavenet_modules.py:42-77 ring buffer (R, max_length); push one column, pop k taps spaced d."""

    def __init__(self, channels, max_length):
        self.channels, self.max_length = channels, max_length
        self.reset()

    def reset(self):  # wavenet_modules.py:74-77
        self.data = torch.zeros(self.channels, self.max_length)
        self.in_pos = 0
        self.out_pos = 0

    def enqueue(self, col):  # wavenet_modules.py:55-57
        self.data[:, self.in_pos] = col.reshape(-1)
        self.in_pos = (self.in_pos + 1) % self.max_length

    def dequeue(self, num_deq, dilation):  # wavenet_modules.py:59-72
        start = self.out_pos - (num_deq - 1) * dilation
        if start < 0:
            head = self.data[:, start::dilation]
            tail = self.data[:, self.out_pos % dilation:self.out_pos + 1:dilation]
            taps = torch.cat((head, tail), 1)
        else:
            taps = self.data[:, start:self.out_pos + 1:dilation]
        self.out_pos = (self.out_pos + 1) % self.max_length
        return taps


def mu_law_expansion(data, mu):  # audio_data.py:156-158 (note: mu = classes = 256, not 255)
    return np.sign(data) * (np.exp(np.abs(data) * np.log(mu + 1)) - 1) / mu


class RestatedWaveNet:
    def __init__(self, cfg, weights, dtype=torch.float32):
        """cfg: ctor kwargs of the reference; weights: name -> ndarray in reference layout."""
        self.cfg = dict(cfg)
        self.layers, self.blocks = cfg["layers"], cfg["blocks"]
        self.k = cfg.get("kernel_size", 2)
        self.classes = cfg.get("classes", 256)
        self.R = cfg["residual_channels"]
        self.bias = cfg.get("bias", False)
        self.dtype = dtype
        self.w = {n: torch.from_numpy(np.asarray(a)).to(dtype) for n, a in weights.items()}
        self.nl = self.layers * self.blocks
        # wavenet_model.py:70-110
        self.dilations = [2 ** i for _ in range(self.blocks) for i in range(self.layers)]
        self.queues = [Queue(self.R, (self.k - 1) * d + 1) for d in self.dilations]
        for q in self.queues:
            q.data = q.data.to(dtype)
        self.receptive_field = 1 + self.blocks * (self.k - 1) * (2 ** self.layers - 1)

    def _b(self, name):
        return self.w.get(name) if self.bias or name.startswith("end_conv") else None

    def reset(self):
        for q in self.queues:
            q.reset()
            q.data = q.data.to(self.dtype)

    def wavenet_step(self, inp):
        """One timestep of wavenet() with queue_dilate; inp is the (1, classes, 1) one-hot."""
        w = self.w
        x = F.conv1d(inp, w["start_conv.weight"], self._b("start_conv.bias"))  # :127
        skip = 0
        for i in range(self.nl):  # :131
            q = self.queues[i]
            q.enqueue(x.data[0])  # :179
            residual = q.dequeue(self.k, self.dilations[i]).unsqueeze(0)  # :180-182
            f = torch.tanh(F.conv1d(residual, w["filter_convs.%d.weight" % i], self._b("filter_convs.%d.bias" % i)))
            g = torch.sigmoid(F.conv1d(residual, w["gate_convs.%d.weight" % i], self._b("gate_convs.%d.bias" % i)))
            x = f * g  # :147-151
            s = F.conv1d(x, w["skip_convs.%d.weight" % i], self._b("skip_convs.%d.bias" % i))  # :157
            skip = s + skip  # :158-162 (length-1 case)
            x = F.conv1d(x, w["residual_convs.%d.weight" % i], self._b("residual_convs.%d.bias" % i))
            x = x + residual[:, :, (self.k - 1):]  # :164-165 newest tap
        x = F.relu(skip)  # :167
        x = F.relu(F.conv1d(x, w["end_conv_1.weight"], w["end_conv_1.bias"]))
        x = F.conv1d(x, w["end_conv_2.weight"], w["end_conv_2.bias"])
        return x

    def _onehot(self, idx):
        inp = torch.zeros(1, self.classes, 1, dtype=self.dtype)
        inp[0, int(idx), 0] = 1.0
        return inp

    def generate_fast(self, num_samples, first_samples=None, temperature=1., regularize=0.,
                      progress_callback=None, progress_interval=100, return_details=False,
                      forced=None):
        """wavenet_model.py:237-315.  ``forced``: optional int array (num_samples,) of indices fed
        back instead of the sampled ones (teacher forcing for logit comparisons; the sampled index is
        still computed and reported)."""
        if first_samples is None:
            first_samples = torch.zeros(1, dtype=torch.long) + (self.classes // 2)  # :245-247
        first_samples = torch.as_tensor(first_samples, dtype=torch.long)
        self.reset()  # :250-251
        n_given = first_samples.numel()
        total = n_given + num_samples
        inp = self._onehot(first_samples[0])
        for i in range(n_given - 1):  # :260-269
            self.wavenet_step(inp)
            inp = self._onehot(first_samples[i + 1])
            if i % progress_interval == 0 and progress_callback is not None:
                progress_callback(i, total)
        generated = np.array([])
        indices = np.zeros(num_samples, dtype=np.int64)
        logits = np.zeros((num_samples, self.classes), dtype=np.float64 if self.dtype == torch.float64 else np.float32)
        # :273-274 -- arange is integral, the subtraction promotes to the default float dtype
        regularizer = torch.pow(torch.arange(self.classes) - self.classes / 2., 2)
        regularizer = (regularizer.squeeze() * regularize).to(self.dtype)
        tic = time.time()
        for i in range(num_samples):  # :276-311
            x = self.wavenet_step(inp).squeeze()
            logits[i] = x.numpy()
            x = x - regularizer  # :280
            if temperature > 0:
                x = x / temperature  # :284
                prob = F.softmax(x, dim=0)
                np_prob = prob.numpy()
                if self.dtype == torch.float64:
                    np_prob = np_prob / np_prob.sum()
                idx = np.random.choice(self.classes, p=np_prob)  # :288 global numpy RNG
                xa = np.array([idx])
            else:
                xa = torch.max(x, 0)[1].reshape(1).numpy()  # :292 restated
            indices[i] = int(xa[0])
            o = (xa / self.classes) * 2. - 1  # :296
            generated = np.append(generated, o)
            nxt = int(xa[0]) if forced is None else int(forced[i])
            inp = self._onehot(nxt)  # :300-302
            if (i + 1) == 100 and return_details is False:
                toc = time.time()
                print("one generating step does take approximately " + str((toc - tic) * 0.01) + " seconds)")
            if (i + n_given) % progress_interval == 0 and progress_callback is not None:
                progress_callback(i + n_given, total)
        audio = mu_law_expansion(generated, self.classes)  # :314
        if return_details:
            return audio, indices, logits
        return audio
