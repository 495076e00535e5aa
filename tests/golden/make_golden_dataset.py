"""Generates tests/golden/golden_dataset_v1.npz with the REAL reference's WavenetDataset (authoring container only):
a small synthetic class-index dataset (three "files") and what the reference returns for it -- lengths and items in
train and test mode for several (item_length, target_length, test_stride) settings.

    python tests/golden/make_golden_dataset.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

SETTINGS = [(40, 8, 100), (40, 8, 3), (33, 5, 1), (64, 16, 7)]  # (item_length, target_length, test_stride)
FILE_LENGTHS = (301, 157, 420)


def files():
    rs = np.random.RandomState(2024)
    return [rs.randint(0, 256, n).astype(np.uint8) for n in FILE_LENGTHS]


def main():
    import ref_shim
    _, _, ad = ref_shim.load()
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "ds.npz")
        np.savez(path, *files())
        for k, (il, tl, stride) in enumerate(SETTINGS):
            for train in (True, False):
                ds = ad.WavenetDataset(path, item_length=il, target_length=tl, train=train, test_stride=stride)
                n = len(ds)
                ids = sorted(set([0, 1, n // 2, n - 1]) | set(range(0, n, max(1, n // 6)))) if n > 0 else []
                tag = "s%d_%s" % (k, "train" if train else "test")
                out[tag + "_len"] = np.int64(n)
                out[tag + "_ids"] = np.asarray(ids, dtype=np.int64)
                xs, ts = [], []
                for i in ids:
                    one_hot, target = ds[i]
                    assert one_hot.shape == (256, il) and float(one_hot.sum()) == il
                    xs.append(one_hot.argmax(0).numpy())
                    ts.append(target.numpy().reshape(-1))
                out[tag + "_x"] = np.asarray(xs, dtype=np.int16).reshape(len(ids), il)
                out[tag + "_t"] = np.asarray(ts, dtype=np.int16).reshape(len(ids), tl)
    np.savez_compressed(os.path.join(HERE, "golden_dataset_v1.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
