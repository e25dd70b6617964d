"""Shotgun spectrograms computed on the device.

Host-side mirror of the reference's spectrogram path for the shotgun VAE:

  ``get_spec``                      ava/preprocessing/utils.py:18-110       same signature, one window
  ``get_spec_batch``                the same for n windows in one call      (what the dataset below uses)
  ``DeviceWindowDataset``           ava/models/window_vae_dataset.py:143-256  ``FixedWindowDataset``
  ``get_fixed_window_data_loaders`` ava/models/window_vae_dataset.py:102-139

The audio of every file is uploaded ONCE (``DeviceAudio``; a day of 32 kHz int16 recordings is 5.5 GB of the 288 GB),
window selection stays on the host (three uniforms per window, drawn from numpy's legacy generator in the reference's
order, so that a seeded call picks the very windows the reference picks), and the spectrograms of a whole batch are
produced by three launches (``csrc/spec.hip``) straight into the fp32 ``[batch, freq, time]`` tensor the VAE step
consumes: no CPU workers, no host-to-device copy of spectrograms.

``nperseg`` may be any length in 64..2048: powers of two (every example script of the reference) run a radix-2 transform, other
lengths a direct fp64 transform of the bins the target grid can touch.  Not covered: ``nperseg`` outside 64..2048, more than 512
target times per window (``NotImplementedError`` otherwise).  There is no CPU
fallback.
"""
import warnings

import numpy as np
import torch

from . import _lib

__all__ = ["EPSILON", "DeviceAudio", "get_spec", "get_spec_batch", "target_freqs_of", "DeviceWindowDataset",
           "DeviceWindowLoader", "get_fixed_window_data_loaders"]

EPSILON = 1e-12                      # utils.py:13
_AUDIO_CODES = {np.dtype(np.int16): 0, np.dtype(np.int32): 1, np.dtype(np.float32): 2, np.dtype(np.float64): 3}


def _mel(a):
    """utils.py:113-115"""
    return 1127 * np.log(1 + a / 700)


def _inv_mel(a):
    """utils.py:118-120"""
    return 700 * (np.exp(a / 1127) - 1)


def target_freqs_of(p):
    """The interpolated frequencies get_spec defaults to (utils.py:80-88)."""
    if p['mel']:
        return _inv_mel(np.linspace(_mel(p['min_freq']), _mel(p['max_freq']), p['num_freq_bins']))
    return np.linspace(p['min_freq'], p['max_freq'], p['num_freq_bins'])


class DeviceAudio:
    """The samples of a list of recordings (1-D arrays as ``scipy.io.wavfile.read`` returns them,
    window_vae_dataset.py:167) concatenated in one device buffer, in their own integer / float dtype."""

    def __init__(self, audio, device="cuda"):
        audio = [np.ascontiguousarray(a) for a in audio]
        if not audio:
            raise ValueError("no audio")
        dt = audio[0].dtype
        if any(a.ndim != 1 for a in audio):
            raise ValueError("expected mono recordings (1-D arrays)")
        if any(a.dtype != dt for a in audio):
            dt = np.result_type(*[a.dtype for a in audio])
            audio = [a.astype(dt) for a in audio]
        if np.dtype(dt) not in _AUDIO_CODES:
            raise TypeError("unsupported audio dtype %s (int16, int32, float32, float64)" % dt)
        self.dtype = np.dtype(dt)
        self.code = _AUDIO_CODES[self.dtype]
        self.lengths = np.array([len(a) for a in audio], dtype=np.int64)
        self.offsets = np.concatenate([[0], np.cumsum(self.lengths)[:-1]]).astype(np.int64)
        self.device = torch.device(device)
        self.samples = torch.from_numpy(np.concatenate(audio)).to(self.device)
        self.file_off = torch.from_numpy(self.offsets).to(self.device)
        self.file_len = torch.from_numpy(self.lengths).to(self.device)

    def __len__(self):
        return len(self.lengths)


_CONST_CACHE = {}
_STAGING = {}       # device -> [ring of (page-locked uint8 buffer, event of the last DMA out of it)], next slot
_WORKSPACE = {}     # (device, stream) -> scratch tensor; reuse is safe in stream order


def _staging(dev, nbytes):
    """next slot of the per-device ring of page-locked parameter buffers (waits for the DMA that last read it)"""
    ring = _STAGING.setdefault(str(dev), [[], 0])
    k = ring[1] % 4
    ring[1] += 1
    if len(ring[0]) <= k:
        ring[0].append([None, None])
    slot = ring[0][k]
    if slot[0] is None or slot[0].numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8)
        cuda = torch.device(dev).type == "cuda"
        slot[0] = buf.pin_memory() if cuda else buf
        slot[1] = torch.cuda.Event() if cuda else None
    elif slot[1] is not None:
        slot[1].synchronize()
    return slot


def _workspace(dev, nbytes):
    key = (str(dev), torch.cuda.current_stream(dev).cuda_stream if torch.device(dev).type == "cuda" else 0)
    ws = _WORKSPACE.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=dev)
        _WORKSPACE[key] = ws
    return ws


def _stft_constants(nperseg, device):
    """scipy.signal.stft's window and 'spectrum' scale for ``nperseg`` (what utils.py:74 applies), on the device"""
    key = (int(nperseg), str(device))
    if key not in _CONST_CACHE:
        from scipy.signal import get_window
        win = get_window('hann', int(nperseg))
        scale = float(np.sqrt(1.0 / win.sum() ** 2))
        _CONST_CACHE[key] = (torch.from_numpy(win).to(device), scale)
    return _CONST_CACHE[key]


def get_spec_batch(audio, file_idx, t1, t2, p, fs, target_times, target_freqs=None, fill_value=-1 / EPSILON,
                   max_dur=None, remove_dc_offset=True, return_max=False):
    """``get_spec`` (utils.py:18-110) for n windows at once.

    ``audio``: a ``DeviceAudio``; ``file_idx`` [n] which recording each window is cut from; ``t1``, ``t2`` [n] onset /
    offset in seconds; ``target_times`` [n, T] the interpolated times of each window (the shotgun dataset passes
    ``linspace(onset, offset, T)``, window_vae_dataset.py:218-219).  Returns the fp32 device tensor ``[n, F, T]`` (and
    the per-window maxima ``[n]`` with ``return_max``).  Enqueued on the current stream; nothing synchronises."""
    t1 = np.ascontiguousarray(t1, dtype=np.float64).reshape(-1)
    t2 = np.ascontiguousarray(t2, dtype=np.float64).reshape(-1)
    n = t1.shape[0]
    file_idx = np.ascontiguousarray(file_idx, dtype=np.int32).reshape(-1)
    target_times = np.ascontiguousarray(target_times, dtype=np.float64)
    if target_freqs is None:
        target_freqs = target_freqs_of(p)
    target_freqs = np.ascontiguousarray(target_freqs, dtype=np.float64).reshape(-1)
    F, T = target_freqs.shape[0], target_times.shape[-1]
    if t2.shape[0] != n or file_idx.shape[0] != n or target_times.shape != (n, T):
        raise ValueError("inconsistent batch shapes")
    if n == 0:
        raise ValueError("empty batch")
    if file_idx.min() < 0 or file_idx.max() >= len(audio):
        raise IndexError("file index out of range")
    if max_dur is None:
        max_dur = p['max_dur']
    too_long = t2 - t1 > max_dur + 1e-4                                             # utils.py:54-58
    if too_long.any():
        i = int(np.argmax(too_long))
        warnings.warn("Found segment longer than max_dur: " + str(t2[i] - t1[i]) + "s, max_dur = " + str(max_dur) + "s")
    s1, s2 = np.rint(t1 * fs), np.rint(t2 * fs)                                     # int(round(.)): half to even
    assert (s1 < s2).all(), "s1 >= s2 for window %d" % int(np.argmin(s2 - s1))     # utils.py:60-61
    nperseg, noverlap = int(p['nperseg']), int(p['noverlap'])
    if nperseg < 64 or nperseg > 2048 or not 0 <= noverlap < nperseg:
        raise NotImplementedError("device get_spec needs 64 <= nperseg <= 2048 and 0 <= noverlap < nperseg")
    if T > 512:
        raise NotImplementedError("device get_spec handles at most 512 target times per window")
    max_samples = int((s2 - s1).max())
    lib, dev = _lib.load(), audio.device
    window, scale = _stft_constants(nperseg, dev)
    # one small upload per batch out of a page-locked ring: [t1 | t2 | target_freqs | target_times] float64, file_idx int32
    nd = 2 * n + F + n * T
    slot = _staging(dev, 8 * nd + 4 * n)
    hd = slot[0][:8 * nd].numpy().view(np.float64)
    hd[:n] = t1; hd[n:2 * n] = t2; hd[2 * n:2 * n + F] = target_freqs; hd[2 * n + F:] = target_times.reshape(-1)
    slot[0][8 * nd:8 * nd + 4 * n].numpy().view(np.int32)[:] = file_idx
    params = slot[0][:8 * nd + 4 * n].to(dev, non_blocking=True)
    if slot[1] is not None:
        slot[1].record()
    pd = params[:8 * nd].view(torch.float64)
    fidx = params[8 * nd:].view(torch.int32)
    d_t1, d_t2, d_tf, d_tt = pd[:n], pd[n:2 * n], pd[2 * n:2 * n + F], pd[2 * n + F:]
    normalize, q_lo, q_gamma = 0, 0, 0.0
    if p.get('within_syll_normalize', False):                                      # utils.py:104-108
        q = float(p['normalize_quantile'])
        if not 0.0 <= q <= 1.0:
            raise ValueError("Quantiles must be in the range [0, 1]")               # np.quantile's own check
        # numpy's 'linear' method: virtual index n q + (alpha + q (1 - alpha - beta)) - 1 with alpha = beta = 1
        cnt = F * T
        virtual = cnt * q + (1 + q * (1 - 1 - 1)) - 1
        if virtual >= cnt - 1:
            q_lo, q_gamma = cnt - 1, 0.0
        else:
            q_lo = int(np.floor(virtual))
            q_gamma = float(virtual - np.floor(virtual))
        normalize = 1
    nbytes = lib.ava_spec_workspace_bytes(n, max_samples, nperseg, noverlap, F, T, normalize)
    ws = _workspace(dev, nbytes)
    out = torch.empty((n, F, T), dtype=torch.float32, device=dev)
    omax = torch.empty(n, dtype=torch.float32, device=dev) if return_max else None
    rc = lib.ava_get_spec_batch(audio.samples.data_ptr(), audio.code, audio.file_off.data_ptr(), audio.file_len.data_ptr(),
                                fidx.data_ptr(), d_t1.data_ptr(), d_t2.data_ptr(), d_tt.data_ptr(), n, max_samples,
                                float(fs), nperseg, noverlap, window.data_ptr(), scale, d_tf.data_ptr(), F, T,
                                float(p['spec_min_val']), float(p['spec_max_val']), float(fill_value),
                                1 if remove_dc_offset else 0, normalize, q_lo, q_gamma, out.data_ptr(),
                                omax.data_ptr() if return_max else None,
                                ws.data_ptr(), ws.numel(), _lib.stream())
    _lib.check(rc, "ava_get_spec_batch")
    return (out, omax) if return_max else out


def get_spec(t1, t2, audio, p, fs=32000, target_freqs=None, target_times=None, fill_value=-1 / EPSILON, max_dur=None,
             remove_dc_offset=True):
    """Drop-in for ``ava.preprocessing.utils.get_spec`` (same arguments, returns ``(spec, True)``), computed on the
    device; ``audio`` is a numpy array (uploaded per call: use ``get_spec_batch`` with a ``DeviceAudio`` in loops) or a
    ``DeviceAudio`` holding one recording.  ``spec`` is a numpy float64 array ``[num_freq_bins, num_time_bins]``
    holding the fp32 values the device produced."""
    if max_dur is None:
        max_dur = p['max_dur']
    if target_times is None:                                                        # utils.py:89-95
        duration = t2 - t1
        if p['time_stretch']:
            duration = np.sqrt(duration * max_dur)
        shoulder = 0.5 * (max_dur - duration)
        target_times = np.linspace(t1 - shoulder, t2 + shoulder, p['num_time_bins'])
    dev_audio = audio if isinstance(audio, DeviceAudio) else DeviceAudio([np.asarray(audio)])
    out = get_spec_batch(dev_audio, [0], [t1], [t2], p, fs, np.asarray(target_times, dtype=np.float64)[None, :],
                         target_freqs=target_freqs, fill_value=fill_value, max_dur=max_dur,
                         remove_dc_offset=remove_dc_offset)
    return out[0].cpu().numpy().astype(np.float64), True


class DeviceWindowDataset:
    """``FixedWindowDataset`` (window_vae_dataset.py:143-256) with the audio resident in HBM and the spectrograms made
    on the device.  Same constructor arguments (``transform`` is accepted and ignored: the items already are fp32
    device tensors, which is what ``numpy_to_tensor`` + ``.to(device)`` produce); ``from_arrays`` builds one from
    in-memory recordings instead of file names.

    ``dataset[index, ...]`` / ``__getitem__(index, seed=None, shoulder=0.05, return_seg_info=False)``: for a list
    ``index`` one device tensor ``[len(index), F, T]`` (the reference returns a list of arrays), for an int ``[F, T]``.
    With a ``seed`` the windows are the ones the reference draws for that seed (same generator, same order of draws,
    including the redraws for windows quieter than ``min_spec_val``)."""

    def __init__(self, audio_filenames, roi_filenames, p, transform=None, dataset_length=2048, min_spec_val=None,
                 device="cuda"):
        from scipy.io import wavfile
        from scipy.io.wavfile import WavFileWarning
        filenames = np.array(sorted(audio_filenames))                                # :164
        with warnings.catch_warnings():
            warnings.filterwarnings("ignore", category=WavFileWarning)
            audio = [wavfile.read(fn)[1] for fn in filenames]                        # :167
            fs = wavfile.read(audio_filenames[0])[0]                                 # :168
        rois = [np.loadtxt(i, ndmin=2) for i in roi_filenames]                       # :173
        self._setup(audio, fs, rois, p, dataset_length, min_spec_val, device)
        self.filenames = filenames
        self.roi_filenames = roi_filenames
        self.transform = transform

    @classmethod
    def from_arrays(cls, audio, fs, rois, p, dataset_length=2048, min_spec_val=None, device="cuda"):
        self = cls.__new__(cls)
        self._setup(audio, fs, [np.asarray(r, dtype=np.float64).reshape(-1, 2) for r in rois], p, dataset_length,
                    min_spec_val, device)
        self.filenames = np.array(["<array %d>" % i for i in range(len(audio))])
        self.roi_filenames = None
        self.transform = None
        return self

    def _setup(self, audio, fs, rois, p, dataset_length, min_spec_val, device):
        self.audio = DeviceAudio(audio, device)
        self.fs = fs
        self.dataset_length = dataset_length
        self.min_spec_val = min_spec_val
        self.p = p
        self.rois = rois
        self.file_weights = np.array([np.sum(np.diff(i)) for i in self.rois])        # :174-175
        self.file_weights /= np.sum(self.file_weights)
        self.roi_weights = []
        for i in range(len(self.rois)):                                              # :176-179
            temp = np.diff(self.rois[i]).flatten()
            self.roi_weights.append(temp / np.sum(temp))
        # numpy's legacy ``choice(a, p=p)`` draws ONE uniform and looks it up in the normalised cumulative sum
        self._file_cdf = self._cdf(self.file_weights)
        self._roi_cdf = [self._cdf(w) for w in self.roi_weights]
        self._target_freqs = target_freqs_of(p)

    @staticmethod
    def _cdf(pvals):
        cdf = np.asarray(pvals, dtype=np.float64).cumsum()
        cdf /= cdf[-1]
        return cdf

    def __len__(self):
        """NOTE: length is arbitrary (window_vae_dataset.py:184-186)"""
        return self.dataset_length

    def _draw(self, rs, m):
        """m candidate windows from the generator, consuming three uniforms each in the reference's order
        (file, segment, onset; window_vae_dataset.py:204-216)"""
        u = rs.random_sample((m, 3))
        file_index = self._file_cdf.searchsorted(u[:, 0], side='right').astype(np.int64)
        lo, hi = np.empty(m), np.empty(m)
        for f in np.unique(file_index):
            sel = file_index == f
            roi_index = self._roi_cdf[f].searchsorted(u[sel, 1], side='right')
            lo[sel] = self.rois[f][roi_index, 0]
            hi[sel] = self.rois[f][roi_index, 1]
        onset = lo + (hi - lo - self.p['window_length']) * u[:, 2]
        offset = onset + self.p['window_length']
        return file_index, onset, offset

    def _specs(self, file_index, onset, offset, shoulder, return_max):
        T = self.p['num_time_bins']
        target_times = np.linspace(onset, offset, T, axis=-1)                        # row i = linspace(onset_i, offset_i, T)
        return get_spec_batch(self.audio, file_index, np.maximum(0.0, onset - shoulder), offset + shoulder, self.p,
                              self.fs, target_times, target_freqs=self._target_freqs, return_max=return_max)

    def __getitem__(self, index, seed=None, shoulder=0.05, return_seg_info=False):
        single_index = False
        try:
            iter(index)
        except TypeError:
            index = [index]
            single_index = True
        n = len(index)
        rs = np.random.RandomState(seed)          # the stream np.random.seed(seed) starts (window_vae_dataset.py:203)
        if self.min_spec_val is None:
            file_index, onset, offset = self._draw(rs, n)
            specs = self._specs(file_index, onset, offset, shoulder, False)
        else:
            # the reference redraws a window until it is loud enough before it moves on (:229-231), i.e. it keeps the
            # first n non-silent candidates of the stream: draw candidates in chunks and filter in order
            got_s, got_f, got_on, got_off, have = [], [], [], [], 0
            while have < n:
                m = max(n - have, 8)
                f_c, on_c, off_c = self._draw(rs, m)
                s_c, mx = self._specs(f_c, on_c, off_c, shoulder, True)
                keep = np.flatnonzero(~(mx.cpu().numpy() < self.min_spec_val))[:n - have]
                if keep.size:
                    got_s.append(s_c[torch.from_numpy(keep).to(s_c.device)])
                    got_f.append(f_c[keep]); got_on.append(on_c[keep]); got_off.append(off_c[keep])
                    have += keep.size
            specs = torch.cat(got_s) if len(got_s) > 1 else got_s[0]
            file_index, onset, offset = np.concatenate(got_f), np.concatenate(got_on), np.concatenate(got_off)
        if return_seg_info:
            if single_index:
                return specs[0], int(file_index[0]), float(onset[0]), float(offset[0])
            return specs, [int(i) for i in file_index], [float(i) for i in onset], [float(i) for i in offset]
        return specs[0] if single_index else specs


class DeviceWindowLoader:
    """What ``DataLoader(FixedWindowDataset(...), batch_size, shuffle, num_workers)`` is to the epoch loops
    (window_vae_dataset.py:127-138): ``len(dataset) / batch_size`` batches per epoch of freshly drawn windows, as fp32
    ``[batch, F, T]`` tensors -- here already on the device (``device_resident``: ``VAE._feed`` hands them through
    without a copy stream)."""
    device_resident = True

    def __init__(self, dataset, batch_size=64, shuffle=False, num_workers=0):
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.shuffle = shuffle          # every item is a random window: the order of indices carries no information
        self.num_workers = num_workers

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.dataset)
        for start in range(0, n, self.batch_size):
            yield self.dataset[list(range(start, min(n, start + self.batch_size)))]


def get_fixed_window_data_loaders(partition, p, batch_size=64, shuffle=(True, False), num_workers=4, min_spec_val=None,
                                  device="cuda"):
    """Mirror of window_vae_dataset.py:102-139: ``{'train': loader, 'test': loader or None}`` over the files of
    ``get_window_partition``'s output, with device-side spectrograms instead of CPU workers."""
    train_dataset = DeviceWindowDataset(partition['train']['audio'], partition['train']['rois'], p,
                                        min_spec_val=min_spec_val, device=device)
    train_loader = DeviceWindowLoader(train_dataset, batch_size=batch_size, shuffle=shuffle[0], num_workers=num_workers)
    if not partition['test']:
        return {'train': train_loader, 'test': None}
    test_dataset = DeviceWindowDataset(partition['test']['audio'], partition['test']['rois'], p,
                                       min_spec_val=min_spec_val, device=device)
    test_loader = DeviceWindowLoader(test_dataset, batch_size=batch_size, shuffle=shuffle[1], num_workers=num_workers)
    return {'train': train_loader, 'test': test_loader}
