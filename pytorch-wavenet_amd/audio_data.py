"""Drop-in for the reference's ``audio_data`` (/root/reference/audio_data.py): the mu-law codec that sits on the
generation path (:133-158) and the dataset item path that feeds ``WaveNetModel.forward`` (:12-130).

``WavenetDataset`` reads the same ``.npz`` files (one uint8 class-index array per audio file, ``arr_0`` ...) and returns
the same ``(one_hot, target)`` items.  Two extensions remove the 256x inflated one-hot from the training input path
(SURVEY.md section 8f rank 3): ``item_indices`` returns the item as class indices, and ``DeviceBatches`` keeps the whole
index stream resident in HBM and cuts ``(indices, target)`` batches there with one gather.
Creating a dataset from audio files needs an audio decoder: ``librosa`` when it is installed (as upstream), otherwise
16-bit/float ``.wav`` files whose sample rate already equals ``sampling_rate`` are read with scipy.
"""
import bisect
import math
import os

import numpy as np
import torch
import torch.utils.data


def mu_law_encoding(data, mu):  # audio_data.py:151-153
    return np.sign(data) * np.log(1 + mu * np.abs(data)) / np.log(mu + 1)


def mu_law_expansion(data, mu):  # audio_data.py:156-158 -- called with mu = classes (256, not 255)
    return np.sign(data) * (np.exp(np.abs(data) * np.log(mu + 1)) - 1) / mu


def quantize_data(data, classes):  # audio_data.py:133-137
    mu_x = mu_law_encoding(data, classes)
    bins = np.linspace(-1, 1, classes)
    return np.digitize(mu_x, bins) - 1


AUDIO_SUFFIXES = (".mp3", ".wav", ".aif", "aiff")


def list_all_audio_files(location):
    """Every audio file below `location`, in os.walk order (same result as the reference's helper, audio_data.py:140-148:
    the dataset builder depends on that order); prints the reference's notice when there is none."""
    found = [os.path.join(folder, name)
             for folder, _subdirs, names in os.walk(location)
             for name in names if name.endswith(AUDIO_SUFFIXES)]
    if not found:
        print("found no audio files in " + location)
    return found


def _load_audio(path, sampling_rate, mono):
    try:
        import librosa
    except ImportError:
        librosa = None
    if librosa is not None and hasattr(librosa, "load"):
        return librosa.load(path=path, sr=sampling_rate, mono=mono)[0]
    if not path.endswith(".wav"):
        raise RuntimeError("decoding %s needs librosa, which is not installed; only .wav files are read without it" % path)
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    if sampling_rate is not None and sr != sampling_rate:
        raise RuntimeError("%s is sampled at %d Hz; resampling to %d Hz needs librosa, which is not installed" % (path, sr, sampling_rate))
    if np.issubdtype(data.dtype, np.integer):
        data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    data = data.astype(np.float32)
    if data.ndim == 2:
        data = data.mean(axis=1) if mono else data.T
    return data


class WavenetDataset(torch.utils.data.Dataset):
    """Same constructor, item indexing and train/test split as audio_data.py:12-130."""

    def __init__(self, dataset_file, item_length, target_length, file_location=None, classes=256, sampling_rate=16000,
                 mono=True, normalize=False, dtype=np.uint8, train=True, test_stride=100):
        self.dataset_file = dataset_file
        self._item_length = item_length
        self._test_stride = test_stride
        self.target_length = target_length
        self.classes = classes
        if not os.path.isfile(dataset_file):
            assert file_location is not None, "no location for dataset files specified"
            self.mono = mono
            self.normalize = normalize
            self.sampling_rate = sampling_rate
            self.dtype = dtype
            self.create_dataset(file_location, dataset_file)
        else:  # parameters of a stored dataset are unknown (also upstream, :47-53)
            self.mono = None
            self.normalize = None
            self.sampling_rate = None
            self.dtype = None
        with np.load(self.dataset_file) as z:
            files = [np.asarray(z["arr_" + str(i)]) for i in range(len(z.keys()))]
        self.data = {"arr_" + str(i): a for i, a in enumerate(files)}
        # One contiguous class-index stream: an item that runs over the end of a file continues in the next one
        # (audio_data.py:105-117 concatenates the two pieces), which is a plain slice of the concatenation.
        self._stream = np.concatenate(files) if files else np.zeros(0, dtype=np.uint8)
        self.start_samples = [0]
        self._length = 0
        self.calculate_length()
        self.train = train

    def create_dataset(self, location, out_file):  # :62-77
        print("create dataset from audio files at", location)
        self.dataset_file = out_file
        files = list_all_audio_files(location)
        processed_files = []
        for i, file in enumerate(files):
            print("  processed " + str(i) + " of " + str(len(files)) + " files")
            file_data = _load_audio(file, self.sampling_rate, self.mono)
            if self.normalize:
                peak = np.abs(file_data).max()
                file_data = file_data / peak if peak > 0 else file_data  # librosa.util.normalize: divide by max |x|
            processed_files.append(quantize_data(file_data, self.classes).astype(self.dtype))
        np.savez(self.dataset_file, *processed_files)

    def calculate_length(self):  # :79-85
        self.start_samples = [0] + np.cumsum([len(self.data["arr_" + str(i)]) for i in range(len(self.data))]).tolist()
        context = self._item_length - (self.target_length - 1)  # samples in front of the first target
        self._length = math.floor((self.start_samples[-1] - context - 1) / self.target_length)

    def set_item_length(self, l):
        self._item_length = l
        self.calculate_length()

    def sample_index(self, idx):
        """Position of item ``idx`` in the concatenated sample stream (:92-97)."""
        if self._test_stride < 2:
            return idx * self.target_length
        if self.train:
            return idx * self.target_length + math.floor(idx / (self._test_stride - 1))
        return self._test_stride * (idx + 1) - 1

    def item_indices(self, idx):
        """Extension: the item_length + 1 class indices of item ``idx`` (int64) -- the window :99-117 cuts out of the files."""
        first = self.sample_index(idx)
        window = self._stream[first:first + self._item_length + 1]
        if len(window) != self._item_length + 1:
            raise IndexError("item %d (samples %d..%d) runs past the end of the dataset (%d samples)"
                             % (idx, first, first + self._item_length, len(self._stream)))
        return window.astype(np.int64)

    def __getitem__(self, idx):  # :91-123
        example = torch.from_numpy(self.item_indices(idx))
        one_hot = torch.zeros(self.classes, self._item_length)
        one_hot.scatter_(0, example[:self._item_length].unsqueeze(0), 1.)
        target = example[-self.target_length:].unsqueeze(0)
        return one_hot, target

    def __len__(self):  # :125-130
        test_length = math.floor(self._length / self._test_stride)
        if self.train:
            return self._length - test_length
        return test_length


class DeviceBatches:
    """Extension: the dataset's concatenated class-index stream resident on one device; ``batch(item_ids)`` cuts
    ``(indices (N, item_length) int32, target (N * target_length,) int64)`` there with one gather -- the inputs of
    ``WaveNetModel.forward_indices`` / ``train_forward_indices`` and of ``F.cross_entropy`` -- instead of building N
    one-hot ``(classes, item_length)`` float tensors on the host (524 MB per batch at BASELINE config 5) and copying them.
    Items equal ``dataset[i]``."""

    def __init__(self, dataset, device):
        self.dataset = dataset
        self.device = torch.device(device)
        self.stream = torch.from_numpy(dataset._stream.astype(np.uint8 if dataset.classes <= 256 else np.int32)).to(self.device)

    def __len__(self):
        return len(self.dataset)

    def batch(self, item_ids):
        ds = self.dataset
        starts = torch.tensor([ds.sample_index(int(i)) for i in item_ids], dtype=torch.int64, device=self.device)
        window = self.stream[starts.unsqueeze(1) + torch.arange(ds._item_length + 1, device=self.device).unsqueeze(0)]
        indices = window[:, :ds._item_length].to(torch.int32)
        target = window[:, -ds.target_length:].to(torch.int64).reshape(-1)
        return indices, target

    def epoch(self, batch_size, shuffle=True, generator=None, rank=0, world=1):
        """Batches of one pass over the items.  Data parallel: every rank draws the SAME permutation (same ``generator``
        seed on all ranks) and keeps items rank, rank + world, ... of it -- disjoint shards of equal size."""
        n = len(self.dataset)
        order = torch.randperm(n, generator=generator) if shuffle else torch.arange(n)
        if world > 1:
            order = order[:n - n % world][rank::world]
        for i in range(0, len(order), batch_size):
            yield self.batch(order[i:i + batch_size].tolist())
