"""TEST INFRASTRUCTURE ONLY -- loader for the read-only upstream reference.

Imports vincentherrmann/pytorch-wavenet from /root/reference *unmodified* and applies the
three import-time patches SURVEY.md section 8(c) / Appendix C documents as necessary to run
the reference's own ``forward()`` and sampled ``generate_fast()`` on torch 2.x:

  1. stub ``librosa``            (audio_data.py:8 -- only used by create_dataset)
  2. ``constant_pad_1d``         (wavenet_modules.py:80-127 is a legacy autograd Function that
                                  modern torch refuses to run; semantics == F.pad)
  3. ``DilatedQueue.enqueue``    (wavenet_modules.py:55-57 receives an (R,1) tensor and assigns
                                  it to an (R,) column; modern torch refuses the broadcast)

Nothing under /root/reference is written.  This module exists only in the authoring
container (the GPU box has no /root/reference): it is used by tests/golden/make_golden.py to
produce committed fixtures and by the CPU-only tests that pin oracle/restated.py and
oracle/wn_oracle.c against the real reference.  The product never imports it.
"""
import os
import sys
import types

REFERENCE_DIR = os.environ.get("WN_REFERENCE_DIR", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REFERENCE_DIR, "wavenet_model.py"))


_loaded = None


def load():
    """Return (wavenet_model, wavenet_modules, audio_data) modules of the *reference*.

    The reference modules are imported under private names (``_ref_wavenet_model`` ...) so they
    never collide with this repository's drop-in modules of the same public names.
    """
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_DIR)
    import importlib.util
    import torch.nn.functional as F

    sys.dont_write_bytecode = True  # reference dir is read-only
    sys.modules.setdefault("librosa", types.ModuleType("librosa"))

    saved = {k: sys.modules.get(k) for k in ("wavenet_modules", "audio_data", "wavenet_model")}
    mods = {}
    try:
        for name in ("wavenet_modules", "audio_data", "wavenet_model"):
            # the reference uses ``from wavenet_modules import *`` -- the plain names must
            # resolve to the reference's files while we import it.
            spec = importlib.util.spec_from_file_location(name, os.path.join(REFERENCE_DIR, name + ".py"))
            m = importlib.util.module_from_spec(spec)
            sys.modules[name] = m
            spec.loader.exec_module(m)
            mods[name] = m
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    for name, m in mods.items():
        sys.modules["_ref_" + name] = m

    wm, ad, mdl = mods["wavenet_modules"], mods["audio_data"], mods["wavenet_model"]

    def enqueue(self, input):  # wavenet_modules.py:55-57
        self.data[:, self.in_pos] = input.reshape(-1)
        self.in_pos = (self.in_pos + 1) % self.max_length

    wm.DilatedQueue.enqueue = enqueue

    def constant_pad_1d(input, target_size, dimension=0, value=0, pad_start=False):  # wavenet_modules.py:80-127
        num_pad = target_size - input.size(dimension)
        assert num_pad >= 0, "target size has to be greater than input size"
        pad = [0, 0] * input.dim()
        pad[2 * (input.dim() - 1 - dimension) + (0 if pad_start else 1)] = num_pad
        return F.pad(input, pad, value=value)

    wm.constant_pad_1d = constant_pad_1d
    mdl.constant_pad_1d = constant_pad_1d
    _loaded = (mdl, wm, ad)
    return _loaded
