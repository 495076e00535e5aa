"""Drop-in for the mu-law helpers of the reference's ``audio_data`` that sit on the generation path
(/root/reference/audio_data.py:133-158).  ``WavenetDataset`` (librosa decoding, .npz mmap items) is host-side
I/O outside this repository's scope and is not provided."""
import numpy as np


def mu_law_encoding(data, mu):  # audio_data.py:151-153
    return np.sign(data) * np.log(1 + mu * np.abs(data)) / np.log(mu + 1)


def mu_law_expansion(data, mu):  # audio_data.py:156-158 -- called with mu = classes (256, not 255)
    return np.sign(data) * (np.exp(np.abs(data) * np.log(mu + 1)) - 1) / mu


def quantize_data(data, classes):  # audio_data.py:133-137
    mu_x = mu_law_encoding(data, classes)
    bins = np.linspace(-1, 1, classes)
    return np.digitize(mu_x, bins) - 1


class WavenetDataset:
    def __init__(self, *a, **kw):
        raise NotImplementedError("WavenetDataset (librosa/.npz dataset I/O) is outside the MI355X hot-path scope; "
                                  "see SURVEY.md section 8(f)")
