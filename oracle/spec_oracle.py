"""CPU restatement of the reference's shotgun-spectrogram path.  TEST INFRASTRUCTURE ONLY:
nothing under ``autoencoded-vocal-analysis_amd/`` may import this module; only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` legs of the benches do.

Restates
  * ``get_spec``                         ava/preprocessing/utils.py:18-110
  * ``_mel`` / ``_inv_mel``              ava/preprocessing/utils.py:113-120
  * ``FixedWindowDataset.__init__``      ava/models/window_vae_dataset.py:145-181 (weights of files and segments)
  * ``FixedWindowDataset.__getitem__``   ava/models/window_vae_dataset.py:189-256 (window selection, retry on silence)

PARITY PIN.  ``get_spec`` calls ``scipy.interpolate.interp2d`` (utils.py:77), which SciPy removed in 1.14; the image
ships SciPy 1.15.3, so the reference's own ``get_spec`` raises ``NotImplementedError`` here and no output of it can be
generated: the INTERPOLATION STEP IS PARITY-UNPINNED.  It is restated through the very FITPACK routines
``interp2d(kind='linear')`` used for a rectangular grid in SciPy <= 1.13 (``dfitpack.regrid_smth`` with kx = ky = 1,
s = 0 for the fit, ``bispev`` for the evaluation -- what ``RectBivariateSpline(kx=1, ky=1, s=0)`` calls), followed by
interp2d's own out-of-bounds rule (``x < x_min or x > x_max`` -> ``fill_value``; points ON the border are inside).
``stft`` is SciPy's (the reference's dependency, present).  The WINDOW SELECTION is pinned to the real
``FixedWindowDataset`` (tests/golden/make_golden.py: ``shotgun_case`` runs the reference class over synthetic wav
files with a recording function handed over through the reference's own ``p['get_spec']`` hook and stores the file
indices / onsets / offsets it drew, every ``get_spec`` call it made and which candidates its ``min_spec_val`` rule
rejected).
"""
import warnings

import numpy as np
from scipy.interpolate import RectBivariateSpline
from scipy.signal import stft

EPSILON = 1e-12        # utils.py:13


def _mel(a):
    """utils.py:113-115"""
    return 1127 * np.log(1 + a / 700)


def _inv_mel(a):
    """utils.py:118-120"""
    return 700 * (np.exp(a / 1127) - 1)


def target_freqs_of(p):
    """utils.py:80-88"""
    if p['mel']:
        return _inv_mel(np.linspace(_mel(p['min_freq']), _mel(p['max_freq']), p['num_freq_bins']))
    return np.linspace(p['min_freq'], p['max_freq'], p['num_freq_bins'])


def _interp2d_linear(x, y, z, xq, yq, fill_value):
    """interp2d(x, y, z, copy=False, bounds_error=False, fill_value=...)(xq, yq, assume_sorted=True), kind 'linear',
    rectangular grid: z has shape [len(y), len(x)], the result [len(yq), len(xq)]."""
    xq = np.atleast_1d(np.asarray(xq, dtype=np.float64))
    yq = np.atleast_1d(np.asarray(yq, dtype=np.float64))
    spl = RectBivariateSpline(x, y, np.ascontiguousarray(z.T), kx=1, ky=1, s=0)
    out = spl(np.clip(xq, x[0], x[-1]), np.clip(yq, y[0], y[-1]), grid=True).T       # [len(yq), len(xq)]
    out = np.array(out, dtype=np.float64)
    if fill_value is not None:
        out[:, (xq < x[0]) | (xq > x[-1])] = fill_value
        out[(yq < y[0]) | (yq > y[-1]), :] = fill_value
    return out


def get_spec(t1, t2, audio, p, fs=32000, target_freqs=None, target_times=None, fill_value=-1 / EPSILON, max_dur=None,
             remove_dc_offset=True):
    """utils.py:18-110, statement by statement."""
    if max_dur is None:
        max_dur = p['max_dur']
    if t2 - t1 > max_dur + 1e-4:
        warnings.warn("Found segment longer than max_dur: " + str(t2 - t1) + "s, max_dur = " + str(max_dur) + "s")
    s1, s2 = int(round(t1 * fs)), int(round(t2 * fs))
    assert s1 < s2, "s1: " + str(s1) + " s2: " + str(s2) + " t1: " + str(t1) + " t2: " + str(t2)
    temp = min(len(audio), s2) - max(0, s1)
    if temp < p['nperseg'] or s2 <= 0 or s1 >= len(audio):
        return np.zeros((p['num_freq_bins'], p['num_time_bins'])), True
    temp_audio = audio[max(0, s1):min(len(audio), s2)]
    if remove_dc_offset:
        temp_audio = temp_audio - np.mean(temp_audio)
    f, t, spec = stft(temp_audio, fs=fs, nperseg=p['nperseg'], noverlap=p['noverlap'])
    t += max(0, t1)
    spec = np.log(np.abs(spec) + EPSILON)
    if target_freqs is None:
        target_freqs = target_freqs_of(p)
    if target_times is None:
        duration = t2 - t1
        if p['time_stretch']:
            duration = np.sqrt(duration * max_dur)
        shoulder = 0.5 * (max_dur - duration)
        target_times = np.linspace(t1 - shoulder, t2 + shoulder, p['num_time_bins'])
    spec = _interp2d_linear(t, f, spec, target_times, target_freqs, fill_value)
    spec -= p['spec_min_val']
    spec /= (p['spec_max_val'] - p['spec_min_val'])
    spec = np.clip(spec, 0.0, 1.0)
    if p['within_syll_normalize']:
        spec -= np.quantile(spec, p['normalize_quantile'])
        spec[spec < 0.0] = 0.0
        spec /= np.max(spec) + EPSILON
    return spec, True


class FixedWindowOracle:
    """FixedWindowDataset (window_vae_dataset.py:143-256) over in-memory audio: ``audio`` a list of 1-D arrays (what
    ``wavfile.read(fn)[1]`` returns per file), ``rois`` a list of [n_i, 2] arrays (``np.loadtxt(fn, ndmin=2)``)."""

    def __init__(self, audio, fs, rois, p, dataset_length=2048, min_spec_val=None, get_spec_fn=None):
        self.audio = audio
        self.fs = fs
        self.rois = [np.asarray(r, dtype=np.float64).reshape(-1, 2) for r in rois]
        self.p = p
        self.dataset_length = dataset_length
        self.min_spec_val = min_spec_val
        self.get_spec_fn = get_spec_fn or get_spec          # the reference's p['get_spec'] hook (:220)
        self.file_weights = np.array([np.sum(np.diff(i)) for i in self.rois])          # :171-172
        self.file_weights /= np.sum(self.file_weights)
        self.roi_weights = []
        for i in range(len(self.rois)):                                                 # :173-176
            temp = np.diff(self.rois[i]).flatten()
            self.roi_weights.append(temp / np.sum(temp))

    def __len__(self):
        return self.dataset_length

    def getitem(self, index, seed=None, shoulder=0.05):
        """``__getitem__(index, seed, shoulder, return_seg_info=True)`` for a list ``index`` (:189-256)"""
        specs, file_indices, onsets, offsets = [], [], [], []
        np.random.seed(seed)
        for _ in index:
            while True:
                file_index = np.random.choice(np.arange(len(self.audio)), p=self.file_weights)
                roi_index = np.random.choice(np.arange(len(self.roi_weights[file_index])), p=self.roi_weights[file_index])
                roi = self.rois[file_index][roi_index]
                onset = roi[0] + (roi[1] - roi[0] - self.p['window_length']) * np.random.rand()
                offset = onset + self.p['window_length']
                target_times = np.linspace(onset, offset, self.p['num_time_bins'])
                spec, flag = self.get_spec_fn(max(0.0, onset - shoulder), offset + shoulder, self.audio[file_index],
                                              self.p, fs=self.fs, target_times=target_times)
                if not flag:
                    continue
                if self.min_spec_val is not None and np.max(spec) < self.min_spec_val:
                    continue
                specs.append(spec)
                file_indices.append(file_index)
                onsets.append(onset)
                offsets.append(offset)
                break
        np.random.seed(None)
        return specs, file_indices, onsets, offsets
