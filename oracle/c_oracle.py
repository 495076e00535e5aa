"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/libwn_oracle.so (the plain-C oracle).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Cfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("layers", "blocks", "dilation_channels", "residual_channels",
                                               "skip_channels", "end_channels", "classes", "kernel_size", "bias")]


_WNAMES = ["start_w", "start_b", "filter_w", "filter_b", "gate_w", "gate_b", "res_w", "res_b", "skip_w", "skip_b",
           "end1_w", "end1_b", "end2_w", "end2_b"]


class _W(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in _WNAMES]


def build(force=False):
    so = os.path.join(_HERE, "libwn_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("wn_oracle.c", "wn_oracle_impl.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libwn_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        for fn in ("wno_generate_f32", "wno_generate_f64"):
            getattr(_LIB, fn).restype = ctypes.c_int
    return _LIB


def stack_weights(cfg, weights):
    """name->ndarray (reference layout) -> the 14 concatenated fp32 arrays of wno_weights."""
    nl = cfg["layers"] * cfg["blocks"]
    bias = cfg.get("bias", False)

    def cat(fmt):
        return np.ascontiguousarray(np.stack([np.asarray(weights[fmt % i]) for i in range(nl)]), dtype=np.float32)

    def one(name):
        return np.ascontiguousarray(np.asarray(weights[name]), dtype=np.float32)

    return {
        "start_w": one("start_conv.weight"), "start_b": one("start_conv.bias") if bias else None,
        "filter_w": cat("filter_convs.%d.weight"), "filter_b": cat("filter_convs.%d.bias") if bias else None,
        "gate_w": cat("gate_convs.%d.weight"), "gate_b": cat("gate_convs.%d.bias") if bias else None,
        "res_w": cat("residual_convs.%d.weight"), "res_b": cat("residual_convs.%d.bias") if bias else None,
        "skip_w": cat("skip_convs.%d.weight"), "skip_b": cat("skip_convs.%d.bias") if bias else None,
        "end1_w": one("end_conv_1.weight"), "end1_b": one("end_conv_1.bias"),
        "end2_w": one("end_conv_2.weight"), "end2_b": one("end_conv_2.bias"),
    }


def regularizer_array(classes, regularize):
    """wavenet_model.py:273-274 evaluated the way torch does (fp32 tensor * python scalar)."""
    import torch
    r = torch.pow(torch.arange(classes) - classes / 2., 2)
    return (r.squeeze() * regularize).numpy().astype(np.float32)


def generate(cfg, weights, num_samples, first_samples=None, temperature=1.0, regularize=0.0, uniforms=None,
             forced=None, precision="f32", want_logits=True):
    """Returns (indices int32 (num_samples,), logits (num_samples, C) or None).

    ``uniforms``: float64 (num_samples,) as drawn by np.random.random_sample (one per generated sample,
    Appendix A item 10); None or temperature<=0 -> greedy.
    """
    L = lib()
    c = _Cfg(cfg["layers"], cfg["blocks"], cfg["dilation_channels"], cfg["residual_channels"], cfg["skip_channels"],
             cfg["end_channels"], cfg.get("classes", 256), cfg.get("kernel_size", 2), int(cfg.get("bias", False)))
    st = stack_weights(cfg, weights)
    w = _W(*[st[n].ctypes.data if st[n] is not None else None for n in _WNAMES])
    C = c.classes
    if first_samples is None:
        first_samples = [C // 2]
    fs = np.ascontiguousarray(np.asarray(first_samples).reshape(-1), dtype=np.int32)
    out_idx = np.zeros(num_samples, dtype=np.int32)
    rdt = np.float32 if precision == "f32" else np.float64
    logits = np.zeros((num_samples, C), dtype=rdt) if want_logits else None
    reg = regularizer_array(C, regularize) if regularize else None
    u = np.ascontiguousarray(uniforms, dtype=np.float64) if (uniforms is not None and temperature > 0) else None
    fz = np.ascontiguousarray(forced, dtype=np.int32) if forced is not None else None
    fn = getattr(L, "wno_generate_" + precision)
    rc = fn(ctypes.byref(c), ctypes.byref(w), fs.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(fs.size),
            ctypes.c_int64(num_samples), ctypes.c_double(temperature),
            reg.ctypes.data_as(ctypes.c_void_p) if reg is not None else None,
            u.ctypes.data_as(ctypes.c_void_p) if u is not None else None,
            fz.ctypes.data_as(ctypes.c_void_p) if fz is not None else None,
            out_idx.ctypes.data_as(ctypes.c_void_p),
            logits.ctypes.data_as(ctypes.c_void_p) if logits is not None else None)
    if rc != 0:
        raise RuntimeError("wno_generate failed: %d" % rc)
    return out_idx, logits


def expand(indices, classes=256):
    """float64 audio exactly as wavenet_model.py:296,314 + audio_data.py:156-158 compute it (numpy)."""
    x = np.asarray(indices).astype(np.int64)
    o = (x / classes) * 2. - 1
    return np.sign(o) * (np.exp(np.abs(o) * np.log(classes + 1)) - 1) / classes
