import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pytorch-wavenet_amd"), os.path.join(ROOT, "oracle"), ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


os.environ.setdefault("WN_TESTING", "1")  # lets the tests pin kernels / forms (WN_KERNEL, WN_V3_MODE, ...): ignored by the library otherwise


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (authoring container only)")


def pytest_collection_modifyitems(config, items):
    import torch
    has_gpu = torch.cuda.is_available()
    skip_gpu = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords and not has_gpu:
            item.add_marker(skip_gpu)


@pytest.fixture(scope="session")
def golden():
    """golden_v1.npz (tiny / cfg1, queue and dilate known answers, forward) + golden_v2.npz (cfg2 / cfg3 generate_fast) +
    golden_v3.npz (forward() outputs, loss and parameter-gradient digests of the reference for cfg2 and the cfg3 stack) +
    golden_v4.npz (the same for clips in the reference's zero-padding regime, shorter than receptive_field + output_length - 1):
    everything in them was produced by the real reference (tests/golden/make_golden.py)."""
    import numpy as np
    merged = {}
    for name in ("golden_v1.npz", "golden_v2.npz", "golden_v3.npz", "golden_v4.npz"):
        z = np.load(os.path.join(ROOT, "tests", "golden", name))
        merged.update({k: z[k] for k in z.files})
    return merged
