"""Host-side driver of the MI355X generation engine (C ABI: include/wn_abi.h).

``Engine`` owns one ``wn_handle``: a planned chain of persistent workgroups with the weights of one
WaveNet packed for LDS residency.  It mirrors what ``WaveNetModel.generate_fast`` needs
(/root/reference/wavenet_model.py:237-315) for one or many independent streams.  PyTorch is used only
for device memory and the current HIP stream.
"""
import ctypes

import numpy as np

from . import _abi


def stack_state(cfg, weights):
    """name -> array (reference Conv1d layout) -> dict of the 14 stacked fp32 host arrays of wn_weight_ptrs."""
    nl = cfg["layers"] * cfg["blocks"]
    bias = bool(cfg.get("bias", False))

    def arr(x):
        if hasattr(x, "detach"):
            x = x.detach().cpu().numpy()
        return np.ascontiguousarray(np.asarray(x), dtype=np.float32)

    def cat(fmt):
        return np.ascontiguousarray(np.stack([arr(weights[fmt % i]) for i in range(nl)]))

    return {
        "start_w": arr(weights["start_conv.weight"]), "start_b": arr(weights["start_conv.bias"]) if bias else None,
        "filter_w": cat("filter_convs.%d.weight"), "filter_b": cat("filter_convs.%d.bias") if bias else None,
        "gate_w": cat("gate_convs.%d.weight"), "gate_b": cat("gate_convs.%d.bias") if bias else None,
        "res_w": cat("residual_convs.%d.weight"), "res_b": cat("residual_convs.%d.bias") if bias else None,
        "skip_w": cat("skip_convs.%d.weight"), "skip_b": cat("skip_convs.%d.bias") if bias else None,
        "end1_w": arr(weights["end_conv_1.weight"]), "end1_b": arr(weights["end_conv_1.bias"]),
        "end2_w": arr(weights["end_conv_2.weight"]), "end2_b": arr(weights["end_conv_2.bias"]),
    }


def regularizer_array(classes, regularize):
    """(c - classes/2)^2 * regularize evaluated like the reference does (wavenet_model.py:273-274):
    a float32 tensor times a python scalar."""
    import torch
    r = torch.pow(torch.arange(classes) - classes / 2., 2)
    return (r.squeeze() * regularize).numpy().astype(np.float32)


class _TorchMem:
    """Device memory, streams and transfers through PyTorch-ROCm (the memory provider of every product Engine)."""
    def __init__(self, device_index):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", device_index)

    def upload(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def empty(self, shape, dtype):
        td = {np.int32: self.torch.int32, np.float32: self.torch.float32, np.float64: self.torch.float64}[dtype]
        return self.torch.empty(shape, dtype=td, device=self.device)

    def ptr(self, a):
        return a.data_ptr() if a is not None else None

    def download(self, a):
        return a.cpu().numpy()

    def stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream


class _Resident:
    """uniforms already uploaded (Engine.upload_uniforms)"""
    def __init__(self, dev, shape):
        self.dev, self.shape = dev, tuple(shape)


class Engine:
    def __init__(self, cfg, weights, n_streams=1, device_index=0, layer_split=0, head_split=0, lib=None, mem=None, pad_channels=True):
        """lib / mem: the loaded C-ABI library and the memory provider (upload / empty / ptr / download / stream); the defaults are
        the HIP library and torch device memory.  (Dependency injection for the host-logic tests, which pass a test double of the C
        ABI together with ITS memory provider -- tests/double_lib.py; nothing in this package knows about it.)
        pad_channels=False: no zero padding of a channel shape the fast kernel is not compiled for (include/wn_abi.h: wn_create)."""
        self.lib = lib if lib is not None else _abi.load_product_library()
        self.cfg = dict(cfg)
        self.n_streams = int(n_streams)
        self.classes = int(cfg.get("classes", 256))
        if mem is not None:
            self.mem = mem
        else:
            import torch
            if not torch.cuda.is_available():
                raise RuntimeError("mi355_wavenet: no HIP device visible; the generation engine needs an MI355X "
                                   "(there is no CPU fallback)")
            self.mem = _TorchMem(device_index)
        c = _abi.wn_config(cfg["layers"], cfg["blocks"], cfg["dilation_channels"], cfg["residual_channels"],
                           cfg["skip_channels"], cfg["end_channels"], self.classes, cfg.get("kernel_size", 2),
                           int(bool(cfg.get("bias", False))), self.n_streams, device_index, layer_split, head_split)
        if not pad_channels:  # WN_CFG_NO_PADDING: plan the model's own channel shape (a handle that serves wn_train_* must)
            c.reserved[0] = 1
        self._h = ctypes.c_void_p()
        self.lib.check(self.lib.dll.wn_create(ctypes.byref(c), ctypes.byref(self._h)))
        self.load_weights(weights)

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.dll.wn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- ABI calls
    def load_weights(self, weights):
        st = stack_state(self.cfg, weights)
        w = _abi.wn_weight_ptrs(*[st[n].ctypes.data if st[n] is not None else None for n in _abi.WEIGHT_FIELDS])
        self.lib.check(self.lib.dll.wn_load_weights(self._h, ctypes.byref(w)))

    def info(self):
        i = _abi.wn_info()
        self.lib.check(self.lib.dll.wn_get_info(self._h, ctypes.byref(i)))
        return {n: getattr(i, n) for n, _ in i._fields_}

    def reset(self):
        """DilatedQueue.reset() of every layer and stream (wavenet_model.py:250-251)."""
        self.lib.check(self.lib.dll.wn_reset(self._h, self.mem.stream()))

    def export_queue(self, layer, stream=0):
        """(data (R, (k-1)d+1) float32, in_pos, out_pos) like the reference's DilatedQueue fields."""
        k = self.cfg.get("kernel_size", 2)
        d = 2 ** (layer % self.cfg["layers"])
        R = self.cfg["residual_channels"]
        data = np.zeros((R, (k - 1) * d + 1), dtype=np.float32)
        ip, op = ctypes.c_int32(), ctypes.c_int32()
        self.lib.check(self.lib.dll.wn_export_queue(self._h, layer, stream, data.ctypes.data, ctypes.byref(ip), ctypes.byref(op)))
        return data, ip.value, op.value

    def set_forward_precision(self, bf16):
        """bf16 operands (fp32 accumulation) for forward_indices; fp32 is the parity default."""
        self.lib.check(self.lib.dll.wn_set_forward_precision(self._h, 1 if bf16 else 0))

    def forward_indices(self, indices, output_length):
        """WaveNetModel.forward() on class indices: int tensor/array (N, L) -> float32 (N*output_length, classes) on the
        engine's device (a torch tensor; numpy for the emulator is not available: GPU only).  Asynchronous."""
        import torch
        idx = torch.as_tensor(indices).to(self.mem.device, torch.int32).contiguous()
        N, L = idx.shape
        out = torch.empty(N * output_length, self.classes, dtype=torch.float32, device=self.mem.device)
        self.lib.check(self.lib.dll.wn_forward(self._h, idx.data_ptr(), N, L, int(output_length), out.data_ptr(), self.mem.stream()))
        return out

    def profile_next(self, n_items):
        self.lib.check(self.lib.dll.wn_profile_next(self._h, int(n_items)))

    def profile_read(self, n_items):
        """int64 (n_workgroups, n_items, 8) wall-clock stamps (100 MHz ticks) of the last profiled job."""
        info = self.info()
        n_wg = info["n_workgroups"] // max(1, info["n_chains"])  # stamps of the first chain
        out = np.zeros((n_wg, n_items, 8), dtype=np.int64)
        self.lib.check(self.lib.dll.wn_profile_read(self._h, out.ctypes.data, out.size))
        return out

    def launch(self, first_dev, n_given, num_samples, temperature, reg_dev, uni_dev, out_dev, logits_dev, timeout_ms=0,
               temps_dev=None):
        """Enqueue one job on the current stream (asynchronous).  temps_dev: optional fp32 (n_streams,) per-stream temperatures."""
        a = _abi.wn_generate_args(self.mem.ptr(first_dev), n_given, num_samples, float(temperature), 0,
                                  self.mem.ptr(reg_dev), self.mem.ptr(uni_dev), self.mem.ptr(out_dev),
                                  self.mem.ptr(logits_dev), self.mem.stream(), int(timeout_ms), 0, self.mem.ptr(temps_dev))
        self.lib.check(self.lib.dll.wn_generate(self._h, ctypes.byref(a)))

    def wait(self):
        self.lib.check(self.lib.dll.wn_wait(self._h))

    def prime(self, first_dev, n_prime, row_stride):
        """Batched teacher-forced priming of freshly reset queues (wn_prime).  Returns False when the engine cannot
        (shape limits / emulator): the caller then primes through wn_generate, one chain pass per sample."""
        rc = self.lib.dll.wn_prime(self._h, self.mem.ptr(first_dev), int(n_prime), int(row_stride), self.mem.stream())
        if rc == _abi.WN_E_UNSUPPORTED:
            return False
        self.lib.check(rc)
        return True

    def prime_host(self, first_samples):
        """prime() on a host array (n_streams, n_prime) of class indices: all of them are teacher-forced inputs."""
        fs = np.ascontiguousarray(np.asarray(first_samples), dtype=np.int32)
        if fs.ndim != 2 or fs.shape[0] != self.n_streams:
            raise ValueError("first_samples must be (n_streams, n_prime)")
        if fs.size and (fs.min() < 0 or fs.max() >= self.classes):
            raise ValueError("first_samples outside [0, classes)")
        if fs.shape[1] == 0:
            return True
        return self.prime(self.mem.upload(fs), fs.shape[1], fs.shape[1])

    def upload_uniforms(self, u):
        """float64 (n_streams, n) uniforms -> a resident handle ``generate(uniforms=...)`` accepts"""
        u = np.ascontiguousarray(np.asarray(u, dtype=np.float64))
        return _Resident(self.mem.upload(u), u.shape)

    # -- convenience: one synchronous generate_fast-shaped job
    PRIME_BATCH_MIN = 64  # given samples from which the GEMM priming path beats the per-sample chain passes

    def generate(self, num_samples, first_samples=None, temperature=1.0, regularize=0.0, uniforms=None,
                 want_logits=False, reset=True, timeout_ms=0, batched_prime=True, while_running=None):
        """first_samples: (n_streams, n_given) or (n_given,) ints (broadcast to every stream) or None -> classes//2.
        uniforms: float64 (n_streams, num_samples) (np.random.random_sample draws) or None -> greedy.
        temperature: a number, or one per stream (a stream with temperature <= 0 is greedy and ignores its uniforms row).
        uniforms may also be what ``upload_uniforms`` returned (already resident: drawn and uploaded while an earlier job ran).
        while_running: optional callable, invoked after the job has been enqueued and before this call waits for it -- host work
        that overlaps the kernel (the facade draws and uploads the NEXT piece's uniforms there).
        Returns indices int32 (n_streams, num_samples) [, logits float32 (n_streams, num_samples, classes)]."""
        ns, C = self.n_streams, self.classes
        if first_samples is None:
            first_samples = np.full((ns, 1), C // 2, dtype=np.int32)  # wavenet_model.py:245-247
        fs = np.asarray(first_samples)
        if fs.ndim == 1:
            fs = np.broadcast_to(fs[None, :], (ns, fs.shape[0]))
        if fs.shape[0] != ns or fs.shape[1] < 1:
            raise ValueError("first_samples must be (n_streams, n_given>=1)")
        if fs.min() < 0 or fs.max() >= C:
            raise ValueError("first_samples outside [0, classes)")
        fs = np.ascontiguousarray(fs, dtype=np.int32)
        temps_dev = None
        if np.ndim(temperature) > 0:
            temps = np.ascontiguousarray(np.asarray(temperature, dtype=np.float32).reshape(ns))
            temps_dev = self.mem.upload(temps)
            temperature = float(temps.max())
        greedy = not (temperature > 0) or uniforms is None
        uni_dev = None
        if not greedy:
            if isinstance(uniforms, _Resident):
                if uniforms.shape != (ns, num_samples):
                    raise ValueError("resident uniforms have shape %r, the job needs %r" % (uniforms.shape, (ns, num_samples)))
                uni_dev = uniforms.dev
            else:
                uni_dev = self.mem.upload(np.ascontiguousarray(np.asarray(uniforms, dtype=np.float64).reshape(ns, num_samples)))
        reg_dev = self.mem.upload(regularizer_array(C, regularize)) if regularize else None
        first_dev = self.mem.upload(fs)
        out_dev = self.mem.empty((ns, max(num_samples, 1)), np.int32)
        logits_dev = self.mem.empty((ns, max(num_samples, 1), C), np.float32) if want_logits else None
        n_given = fs.shape[1]
        if reset:
            self.reset()
            if batched_prime and n_given - 1 >= self.PRIME_BATCH_MIN and self.prime(first_dev, n_given - 1, n_given):
                first_dev = self.mem.upload(np.ascontiguousarray(fs[:, -1:]))  # the last given sample is the next input
                n_given = 1
        self.launch(first_dev, n_given, num_samples, temperature, reg_dev, uni_dev, out_dev, logits_dev, timeout_ms, temps_dev)
        if while_running is not None:
            while_running()
        self.wait()
        idx = self.mem.download(out_dev)[:, :num_samples]
        if want_logits:
            return idx, self.mem.download(logits_dev)[:, :num_samples]
        return idx
