"""ctypes binding of include/wn_abi.h (libwn_mi355.so).

The product loads exactly one library: ``libwn_mi355.so`` next to this file (built in-tree by
``pytorch-wavenet_amd/build.py`` with hipcc for gfx950).  If it is missing this module raises --
there is no CPU or torch fallback for the generation path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB = os.path.join(_HERE, "libwn_mi355.so")

WN_OK, WN_E_BADARG, WN_E_UNSUPPORTED, WN_E_HIP, WN_E_NOMEM, WN_E_TIMEOUT, WN_E_STATE, WN_E_BUSY = 0, -1, -2, -3, -4, -5, -6, -7
ERROR_NAMES = {0: "WN_OK", -1: "WN_E_BADARG", -2: "WN_E_UNSUPPORTED", -3: "WN_E_HIP", -4: "WN_E_NOMEM",
               -5: "WN_E_TIMEOUT", -6: "WN_E_STATE", -7: "WN_E_BUSY"}


class WnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERROR_NAMES.get(code, "?"), code, msg))
        self.code = code


class wn_config(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "layers", "blocks", "dilation_channels", "residual_channels", "skip_channels", "end_channels", "classes",
        "kernel_size", "bias", "n_streams", "device_id", "layer_split", "head_split")] + [("reserved", ctypes.c_int32 * 3)]


WEIGHT_FIELDS = ["start_w", "start_b", "filter_w", "filter_b", "gate_w", "gate_b", "res_w", "res_b", "skip_w", "skip_b",
                 "end1_w", "end1_b", "end2_w", "end2_b"]


class wn_weight_ptrs(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in WEIGHT_FIELDS]


class wn_generate_args(ctypes.Structure):
    _fields_ = [("first_samples", ctypes.c_void_p), ("n_given", ctypes.c_int64), ("num_samples", ctypes.c_int64),
                ("temperature", ctypes.c_float), ("flags", ctypes.c_int32), ("regularizer", ctypes.c_void_p),
                ("uniforms", ctypes.c_void_p), ("out_idx", ctypes.c_void_p), ("dbg_logits", ctypes.c_void_p),
                ("hip_stream", ctypes.c_void_p), ("timeout_ms", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("stream_temperatures", ctypes.c_void_p)]


ABI_VERSION = 5  # include/wn_abi.h: WN_ABI_VERSION


class wn_info(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("abi_version", "n_layers", "layer_split", "head_split", "n_workgroups",
                                               "lds_bytes", "n_compute_units", "receptive_field")] + \
               [(n, ctypes.c_int64) for n in ("weight_bytes", "queue_bytes", "handoff_bytes", "evals_done")] + \
               [(n, ctypes.c_int32) for n in ("kernel_variant", "n_chains", "streams_per_item", "head_replicas", "n_samplers", "dev_overrides",
                                               "layers_per_workgroup", "gate_shared", "gate_waited_ms", "gate_need_per_xcd",
                                               "forward_native", "workgroups_per_cu", "resident_timeout_ms", "skip_lane_slots")]


class wn_adam_args(ctypes.Structure):
    _fields_ = [("n_tensors", ctypes.c_int32), ("device_id", ctypes.c_int32), ("sizes", ctypes.c_void_p), ("params", ctypes.c_void_p),
                ("grads", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p),
                ("lr", ctypes.c_double), ("beta1", ctypes.c_double), ("beta2", ctypes.c_double), ("eps", ctypes.c_double), ("weight_decay", ctypes.c_double),
                ("max_grad_norm", ctypes.c_double), ("step", ctypes.c_int64), ("total_norm", ctypes.c_void_p), ("scratch", ctypes.c_void_p),
                ("hip_stream", ctypes.c_void_p), ("flags", ctypes.c_int64)]


WN_ADAM_NORM_ONLY, WN_ADAM_NORM_KEEP, WN_ADAM_NORM_GIVEN = 1, 2, 4

TRAIN_TENSOR_ARRAYS = ("filter_w", "gate_w", "res_w", "skip_w", "filter_b", "gate_b", "res_b", "skip_b")
TRAIN_TENSOR_SINGLES = ("start_w", "start_b", "end1_w", "end1_b", "end2_w", "end2_b")


class wn_train_tensors(ctypes.Structure):
    _fields_ = [("n_layers", ctypes.c_int32), ("reserved", ctypes.c_int32)] + [(n, ctypes.c_void_p) for n in TRAIN_TENSOR_ARRAYS + TRAIN_TENSOR_SINGLES]


EXPORTS = ["wn_abi_version", "wn_create", "wn_destroy", "wn_load_weights", "wn_reset", "wn_generate", "wn_wait",
           "wn_get_info", "wn_export_queue", "wn_forward", "wn_set_forward_precision", "wn_prime", "wn_train_get_layout", "wn_train_export_params", "wn_train_forward", "wn_train_backward", "wn_train_loss", "wn_train_pack", "wn_train_unpack_grads", "wn_train_set_deterministic", "wn_adam_step", "wn_profile_next", "wn_profile_read", "wn_last_error"]


TRAIN_SECTIONS = ("fg", "bfg", "res", "bres", "skip", "bskip", "bskip_total", "w1", "b1", "w2", "b2", "start_t", "start_b")


class wn_train_layout(ctypes.Structure):
    _fields_ = [("total", ctypes.c_int64)] + [(n, ctypes.c_int64) for n in TRAIN_SECTIONS]


class Library:
    """A loaded library that exports include/wn_abi.h."""

    def __init__(self, path):
        self.path = path
        # One HIP runtime per process: torch bundles its own libamdhip64.so.7; importing torch FIRST makes the loader
        # resolve our DT_NEEDED libamdhip64.so.7 to that already-loaded copy, so torch's device pointers and streams
        # are valid in our launches (two runtimes in one process cannot even both open the device).
        import torch  # noqa: F401
        self.dll = ctypes.CDLL(path)
        d = self.dll
        for name in EXPORTS:
            if not hasattr(d, name):
                raise RuntimeError("%s does not export %s" % (path, name))
        d.wn_abi_version.restype = ctypes.c_int
        d.wn_last_error.restype = ctypes.c_char_p
        d.wn_create.argtypes = [ctypes.POINTER(wn_config), ctypes.POINTER(ctypes.c_void_p)]
        d.wn_destroy.argtypes = [ctypes.c_void_p]
        d.wn_destroy.restype = None
        d.wn_load_weights.argtypes = [ctypes.c_void_p, ctypes.POINTER(wn_weight_ptrs)]
        d.wn_reset.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        d.wn_generate.argtypes = [ctypes.c_void_p, ctypes.POINTER(wn_generate_args)]
        d.wn_wait.argtypes = [ctypes.c_void_p]
        d.wn_get_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(wn_info)]
        d.wn_export_queue.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                      ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
        d.wn_forward.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        d.wn_set_forward_precision.argtypes = [ctypes.c_void_p, ctypes.c_int32]
        d.wn_prime.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
        d.wn_train_get_layout.argtypes = [ctypes.c_void_p, ctypes.POINTER(wn_train_layout)]
        d.wn_train_export_params.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        d.wn_train_forward.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                       ctypes.c_void_p, ctypes.c_void_p]
        d.wn_train_backward.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        d.wn_train_loss.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        d.wn_train_pack.argtypes = [ctypes.c_void_p, ctypes.POINTER(wn_train_tensors), ctypes.c_void_p, ctypes.c_void_p]
        d.wn_train_unpack_grads.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(wn_train_tensors), ctypes.c_void_p]
        d.wn_train_set_deterministic.argtypes = [ctypes.c_void_p, ctypes.c_int32]
        d.wn_adam_step.argtypes = [ctypes.POINTER(wn_adam_args)]
        d.wn_profile_next.argtypes = [ctypes.c_void_p, ctypes.c_int32]
        d.wn_profile_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
        for name in EXPORTS:
            if name not in ("wn_destroy", "wn_last_error"):
                getattr(d, name).restype = ctypes.c_int
        if d.wn_abi_version() != ABI_VERSION:
            raise RuntimeError("%s: ABI version %d, expected %d" % (path, d.wn_abi_version(), ABI_VERSION))

    def last_error(self):
        return (self.dll.wn_last_error() or b"").decode("utf-8", "replace")

    def check(self, rc):
        if rc != 0:
            raise WnError(rc, self.last_error())


_product = None


def load_product_library():
    """The HIP engine.  Fails loudly when it has not been built -- there is no fallback."""
    global _product
    if _product is None:
        if not os.path.exists(PRODUCT_LIB):
            raise RuntimeError(
                "mi355_wavenet: %s not found. Build it with `python pytorch-wavenet_amd/build.py` (needs hipcc, "
                "targets gfx950). The generation path has no CPU/torch fallback." % PRODUCT_LIB)
        _product = Library(PRODUCT_LIB)
    return _product
