"""Deterministic synthetic spectrograms / weights / noise and a loader with the
reference's loader contract.

The reference feeds the VAE from hdf5 files through ``SyllableDataset`` /
``get_syllable_data_loaders`` (``ava/models/vae_dataset.py:62-145``): a
``torch.utils.data.DataLoader`` yielding CPU float32 ``[B,128,128]`` tensors,
whose ``dataset`` supports ``len()`` and indexing by an int *or* an iterable of
ints (used by ``VAE.visualize``, ``ava/models/vae.py:504-507``).  There are no
data files on the benchmark machines, so this module is the synthetic
counterpart with the same contract.

All values come from a pure-integer splitmix64 hash, so
the same arrays are produced on any platform with no torch RNG involved.
Value distribution of the spectrograms: ``clip(1.4*U[0,1) - 0.4, 0, 1)`` which
mimics clipped log-spectrograms (``ava/preprocessing/utils.py:102-104``): about
29 % exact zeros, the rest uniform.
"""
import numpy as np
import torch
from torch.utils.data import Dataset, DataLoader

from .layout import param_specs, X_SHAPE

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def u01(n, salt, start=0):
    """``n`` uniform float64 in [0,1): splitmix64 finaliser of ``i + salt*GOLD``."""
    with np.errstate(over="ignore"):
        x = np.arange(start, start + n, dtype=np.uint64) + np.uint64(salt) * _GOLD
        x ^= x >> np.uint64(30)
        x *= _M1
        x ^= x >> np.uint64(27)
        x *= _M2
        x ^= x >> np.uint64(31)
    return (x >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def gauss(n, salt):
    """Box-Muller standard normals (float64) from two hash streams."""
    return np.sqrt(-2.0 * np.log(1.0 - u01(n, salt))) * np.cos(2.0 * np.pi * u01(n, salt + 7777))


def spectrograms(batch, salt=1001, start_item=0, shape=X_SHAPE):
    """``[batch,H,W]`` float32 synthetic spectrograms; item ``i`` of the stream
    ``salt`` is always the same array regardless of how it is batched."""
    hw = shape[0] * shape[1]
    u = u01(batch * hw, salt, start=start_item * hw)
    return np.clip(1.4 * u - 0.4, 0.0, 1.0).astype(np.float32).reshape(batch, *shape)


def noise(batch, z_dim, salt_w=2002, salt_d=3003):
    """(eps_W [B,1], eps_D [B,z]) float32, the two draws of
    ``LowRankMultivariateNormal.rsample`` in their reference order."""
    eps_w = gauss(batch, salt_w).astype(np.float32).reshape(batch, 1)
    eps_d = gauss(batch * z_dim, salt_d).astype(np.float32).reshape(batch, z_dim)
    return eps_w, eps_d


def fixture_parameters(z_dim, x_shape=X_SHAPE):
    """name -> float32 ndarray for all 80 parameters (Appendix E recipe):
    weights ``r/sqrt(fan)``, BN weight ``1+0.1r``, BN bias ``0.1r``, other
    biases ``0.25r`` with ``r = 2*u01(numel, index+1) - 1``."""
    out = {}
    for s in param_specs(z_dim, x_shape):
        r = 2.0 * u01(s.numel, s.index + 1) - 1.0
        if len(s.shape) >= 2:
            fan = 1
            for d in s.shape[1:]:
                fan *= d
            v = r / np.sqrt(fan)
        elif s.layer.startswith("bn"):
            v = 1.0 + 0.1 * r if s.kind == "weight" else 0.1 * r
        else:
            v = 0.25 * r
        out[s.name] = v.astype(np.float32).reshape(s.shape)
    return out


def latent_conditions(n_per=(70, 53, 91), z=32, salt=9100):
    """Synthetic latent means of several "conditions" for the MMD^2 estimators (``ava/plotting/mmd_plots.py``):
    Gaussians with different means / scales from the hash recipe.  Returns ``(latent [N,z] float64, condition [N])``."""
    lat, cond = [], []
    for c, n in enumerate(n_per):
        g = gauss(n * z, salt + c).reshape(n, z)
        shift = 0.6 * c * np.cos(np.arange(z) * (c + 1.0))
        lat.append(g * (1.0 + 0.25 * c) + shift)
        cond += [c] * n
    return np.concatenate(lat).astype(np.float64), np.array(cond)


class SyntheticSpecDataset(Dataset):
    """Synthetic stand-in for ``SyllableDataset`` (``vae_dataset.py:102-145``)."""

    def __init__(self, num_specs, salt=1001, shape=X_SHAPE):
        self.num_specs = int(num_specs)
        self.salt = salt
        self.shape = shape

    def __len__(self):
        return self.num_specs

    def _one(self, i):
        i = int(i)
        if i < 0 or i >= self.num_specs:
            raise IndexError(i)
        return torch.from_numpy(spectrograms(1, self.salt, start_item=i, shape=self.shape)[0])

    def __getitem__(self, index):
        try:
            it = iter(index)
        except TypeError:
            return self._one(index)
        return [self._one(i) for i in it]


def get_synthetic_data_loaders(num_train, num_test=0, batch_size=64, shuffle=(True, False),
                               num_workers=0, salt=1001):
    """Same return contract as ``get_syllable_data_loaders``
    (``vae_dataset.py:62-97``): ``{'train': DataLoader, 'test': DataLoader|None}``."""
    train = DataLoader(SyntheticSpecDataset(num_train, salt), batch_size=batch_size,
                       shuffle=shuffle[0], num_workers=num_workers)
    if not num_test:
        return {"train": train, "test": None}
    test = DataLoader(SyntheticSpecDataset(num_test, salt + 1), batch_size=batch_size,
                      shuffle=shuffle[1], num_workers=num_workers)
    return {"train": train, "test": test}


# ---- synthetic recordings for the shotgun-spectrogram path ----------------------------
FINCH_PARAMS = {          # examples/finch_window_mwe.py:29-49 (the numeric entries)
    'fs': 32000, 'num_freq_bins': X_SHAPE[0], 'num_time_bins': X_SHAPE[1], 'nperseg': 512, 'noverlap': 256,
    'max_dur': 1e9, 'window_length': 0.12, 'min_freq': 400, 'max_freq': 10e3, 'spec_min_val': 2.0,
    'spec_max_val': 6.5, 'mel': True, 'time_stretch': False, 'within_syll_normalize': False,
}
MOUSE_PARAMS = {          # examples/mouse_window_mwe.py:29-49
    'fs': 250000, 'num_freq_bins': X_SHAPE[0], 'num_time_bins': X_SHAPE[1], 'nperseg': 1024, 'noverlap': 512,
    'max_dur': 1e9, 'window_length': 0.20, 'min_freq': 30e3, 'max_freq': 110e3, 'spec_min_val': -6.5,
    'spec_max_val': -2.0, 'mel': False, 'time_stretch': False, 'within_syll_normalize': False,
}


def recordings(n_files=3, fs=32000, seconds=2.0, salt=4004, dtype=np.int16, amplitude=3000.0):
    """``n_files`` synthetic mono recordings (frequency sweeps with harmonics over noise, bursts separated by near
    silence) of slightly different lengths, and per file an ``[n_i, 2]`` array of vocalisation segments (onset, offset
    in seconds) as ``np.loadtxt(roi_file, ndmin=2)`` returns it.  Pure numpy, hash-seeded: identical wherever generated."""
    audio, rois = [], []
    for f in range(n_files):
        n = int(fs * seconds * (1.0 + 0.13 * f))
        t = np.arange(n) / fs
        noise = gauss(n, salt + 17 * f)
        f0 = 0.06 * fs * (1.0 + 0.5 * f / max(1, n_files))
        phase = 2.0 * np.pi * (f0 * t + 0.04 * fs * np.sin(2.0 * np.pi * 3.0 * t) / (2.0 * np.pi * 3.0))
        env = (np.sin(2.0 * np.pi * 2.5 * t + f) > -0.2).astype(np.float64)          # bursts
        x = amplitude * env * (np.sin(phase) + 0.4 * np.sin(2.1 * phase) + 0.15 * np.sin(3.3 * phase))
        x = x + 0.02 * amplitude * noise + 0.01 * amplitude                           # noise floor + a DC offset
        if np.issubdtype(np.dtype(dtype), np.integer):
            x = np.clip(np.rint(x), np.iinfo(dtype).min, np.iinfo(dtype).max)
        else:
            x = x / 32768.0
        audio.append(x.astype(dtype))
        dur = n / fs
        k = 3 + f
        edges = np.linspace(0.02, dur - 0.02, 2 * k)
        rois.append(np.stack([edges[0::2], edges[1::2]], axis=1))
    return audio, rois
