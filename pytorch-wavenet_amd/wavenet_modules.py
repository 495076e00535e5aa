"""Drop-in for the reference's ``wavenet_modules`` (same public names and behaviour):
``dilate``, ``DilatedQueue``, ``constant_pad_1d`` / ``ConstantPad1d``.

These are the host-side tensor helpers used by ``WaveNetModel.forward()`` (training) and kept for API
compatibility (``model.dilated_queues``); the generation hot path does not use them -- its queues live on
the GPU inside the HIP engine (csrc/wn_kernel.h).  Behaviour follows /root/reference/wavenet_modules.py:
dilate :10-39, DilatedQueue :42-77, ConstantPad1d/constant_pad_1d :80-127.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import Parameter  # noqa: F401  (re-exported like the reference's star-import surface)
from torch.autograd import Variable, Function  # noqa: F401


def constant_pad_1d(input, target_size, dimension=0, value=0, pad_start=False):
    """Pad ``input`` along ``dimension`` with ``value`` up to ``target_size`` (at the start if ``pad_start``).
    Same contract as wavenet_modules.py:121-127; differentiable (F.pad), so gradients crop like the
    reference's hand-written backward (:106-118)."""
    num_pad = target_size - input.size(dimension)
    assert num_pad >= 0, 'target size has to be greater than input size'
    if num_pad == 0:
        return input
    spec = [0, 0] * input.dim()
    spec[2 * (input.dim() - 1 - dimension) + (0 if pad_start else 1)] = num_pad
    return F.pad(input, spec, mode="constant", value=value)


class ConstantPad1d(nn.Module):
    """Name-compatible wrapper (the reference's is a legacy autograd Function, unusable on torch >= 1.3)."""

    def __init__(self, target_size, dimension=0, value=0, pad_start=False):
        super().__init__()
        self.target_size, self.dimension, self.value, self.pad_start = target_size, dimension, value, pad_start

    def forward(self, input):
        return constant_pad_1d(input, self.target_size, self.dimension, self.value, self.pad_start)


def dilate(x, dilation, init_dilation=1, pad_start=True):
    """(N, C, L) at dilation ``init_dilation`` -> (N*f, C, L/f) at ``dilation``, f = dilation/init_dilation
    (wavenet_modules.py:10-39).

    An (n, c, l) tensor at dilation n is one sequence of l*n timesteps, time index j*n + i; re-dilating is
    re-folding that sequence with a different row count.  If l is not a multiple of f each row is zero padded
    (at the start by default), i.e. the sequence gains pad*n leading zeros -- the reference's quirk that
    forward() must reproduce."""
    n, c, l = x.size()
    factor = dilation / init_dilation
    if factor == 1:
        return x
    padded_l = int(np.ceil(l / factor) * factor)
    if padded_l != l:
        x = constant_pad_1d(x, padded_l, dimension=2, pad_start=pad_start)
        l = padded_l
    rows = math.ceil(n * dilation / init_dilation)
    cols = math.ceil(l * init_dilation / dilation)
    seq = x.permute(1, 2, 0).contiguous().view(c, cols, rows)  # (c, time) refolded as (c, cols, rows)
    return seq.permute(2, 0, 1).contiguous()


def _zeros(dtype, *shape):
    """``dtype`` may be a legacy tensor type object (torch.FloatTensor, torch.cuda.FloatTensor -- the
    reference's convention, wavenet_modules.py:53) or a torch.dtype."""
    if isinstance(dtype, torch.dtype):
        return torch.zeros(*shape, dtype=dtype)
    return dtype(*shape).zero_()


class DilatedQueue:
    """Ring buffer (num_channels, max_length): ``enqueue`` writes one column at ``in_pos``; ``dequeue`` returns
    ``num_deq`` columns spaced ``dilation`` ending at ``out_pos``, oldest first (wavenet_modules.py:42-77).

    ``data`` / ``in_pos`` / ``out_pos`` are the reference's attributes.  After ``WaveNetModel.generate_fast`` the queues
    of the run live on the GPU; the facade then only leaves a loader here (``_defer``) and the first access to one of the
    three attributes downloads that layer's ring (C ABI wn_export_queue) -- the reference leaves its queues in their final
    state (wavenet_model.py:177-184), and callers that never look pay nothing."""

    def __init__(self, max_length, data=None, dilation=1, num_deq=1, num_channels=1, dtype=torch.FloatTensor):
        self._lazy = None
        self._in_pos = 0
        self._out_pos = 0
        self.num_deq = num_deq
        self.num_channels = num_channels
        self.dilation = dilation
        self.max_length = max_length
        self._data = data
        self.dtype = dtype
        if data is None:
            self._data = _zeros(dtype, num_channels, max_length)

    # -- lazily materialised state
    def _defer(self, loader):
        """loader() -> (float32 ndarray (num_channels, max_length), in_pos, out_pos), called at most once."""
        self._lazy = loader

    def _sync(self):
        loader, self._lazy = self._lazy, None
        if loader is not None:
            data, ip, op = loader()
            t = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32))
            self._data = t.type(self.dtype) if not isinstance(self.dtype, torch.dtype) else t.to(self.dtype)
            self._in_pos, self._out_pos = int(ip), int(op)

    def _get(name):  # noqa: N805
        def getter(self):
            if self._lazy is not None:
                self._sync()
            return getattr(self, name)

        def setter(self, value):
            if self._lazy is not None:
                self._sync()
            setattr(self, name, value)
        return property(getter, setter)

    data = _get("_data")
    in_pos = _get("_in_pos")
    out_pos = _get("_out_pos")
    del _get

    def __getstate__(self):  # pickles carry the reference's attribute names (and no loader)
        if self._lazy is not None:
            self._sync()
        st = {k: v for k, v in self.__dict__.items() if k not in ("_lazy", "_data", "_in_pos", "_out_pos")}
        st.update(data=self._data, in_pos=self._in_pos, out_pos=self._out_pos)
        return st

    def __setstate__(self, st):  # also accepts a queue pickled by the reference's own class
        st = dict(st)
        self._lazy = None
        self._data = st.pop("data", st.pop("_data", None))
        self._in_pos = st.pop("in_pos", st.pop("_in_pos", 0))
        self._out_pos = st.pop("out_pos", st.pop("_out_pos", 0))
        st.pop("_lazy", None)
        self.__dict__.update(st)

    def enqueue(self, input):
        self.data[:, self.in_pos] = input.reshape(-1)
        self.in_pos = (self.in_pos + 1) % self.max_length

    def dequeue(self, num_deq=1, dilation=1):
        first = self.out_pos - (num_deq - 1) * dilation
        if first >= 0:
            cols = list(range(first, self.out_pos + 1, dilation))
        else:
            # the reference concatenates data[:, first::dilation] and data[:, out_pos % dilation : out_pos+1 : dilation]
            cols = list(range(self.max_length + first, self.max_length, dilation))
            cols += list(range(self.out_pos % dilation, self.out_pos + 1, dilation))
        t = self.data.index_select(1, torch.as_tensor(cols, dtype=torch.long, device=self.data.device))
        self.out_pos = (self.out_pos + 1) % self.max_length
        return t

    def reset(self):
        # rebinds ``data`` to fresh zeros like the reference (:75) -- on first access: generate_fast() resets all queues per call
        # (wavenet_model.py:250-251) and then leaves a loader for the GPU state here; zero-filling 50 rings was 3.6 ms per call
        nc, ml = self.num_channels, self.max_length
        self._lazy = lambda: (np.zeros((nc, ml), dtype=np.float32), 0, 0)
        self._data = None
        self._in_pos = 0
        self._out_pos = 0
