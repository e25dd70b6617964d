"""GPU parity tests of the shotgun-spectrogram row: the HIP path (through the C ABI, via
ava_amd.spec) against oracle/spec_oracle.py on the same windows.

Tolerance.  The device computes in fp64 like the reference and emits fp32, so the comparison is against the ORACLE'S
VALUES ROUNDED TO fp32: at most one fp32 ulp (6e-8 on [0, 1]) anywhere, and bit-identical for all but a handful of
pixels (an fp64 discrepancy of 1e-13 crosses an fp32 rounding boundary with probability ~1e-6 per pixel)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from ava_amd import synthetic as syn
from oracle import spec_oracle as so

pytestmark = pytest.mark.gpu

ULP = 6e-8
CASES = {"finch": (syn.FINCH_PARAMS, 2.0), "mouse": (syn.MOUSE_PARAMS, 1.0)}


def _recordings(name, dtype=np.int16):
    p, seconds = CASES[name]
    return syn.recordings(n_files=3, fs=p['fs'], seconds=seconds, dtype=dtype)


def _oracle_batch(audio, fidx, t1, t2, p, fs, tts, **kw):
    return np.stack([so.get_spec(t1[i], t2[i], audio[fidx[i]], p, fs=fs, target_times=tts[i], **kw)[0]
                     for i in range(len(t1))])


def _check(dev, want, max_mismatch_frac=1e-3):
    got = dev.cpu().numpy()
    w32 = want.astype(np.float32)
    assert got.shape == w32.shape and got.dtype == np.float32
    assert np.abs(got.astype(np.float64) - want).max() <= ULP
    assert (got != w32).mean() <= max_mismatch_frac


@pytest.mark.parametrize("name", ["finch", "mouse"])
def test_batch_matches_oracle_on_the_reference_windows(name):
    """the windows the real FixedWindowDataset drew (golden), plus windows at the edges of the recordings"""
    from ava_amd import spec as sp
    G = load_golden("shotgun.npz")
    p = dict(CASES[name][0])
    fs, T = p['fs'], p['num_time_bins']
    audio, _ = _recordings(name)
    fidx = list(G[name + ".plain.file_indices"]) + [0, 1, 2, 0, 1]
    on = list(G[name + ".plain.onsets"])
    dur = [len(a) / fs for a in audio]
    wl = p['window_length']
    on += [0.0, dur[1] - wl - 0.01, dur[2] - 0.5 * wl, -0.3 * wl, dur[1] + 1.0]     # start, near end, over the end, before 0, outside
    on = np.array(on)
    off = on + wl
    t1, t2 = np.maximum(0.0, on - 0.05), off + 0.05
    tts = np.linspace(on, off, T, axis=-1)
    dev = sp.get_spec_batch(sp.DeviceAudio(audio), fidx, t1, t2, p, fs, tts)
    want = _oracle_batch(audio, fidx, t1, t2, p, fs, tts)
    _check(dev, want)
    assert want[:16].max() > 0.5 and (want[:16] > 0).mean() > 0.02        # the fixtures are not blank
    assert not dev[-1].any().item()                                        # window outside the recording: zeros (utils.py:68-69)


@pytest.mark.parametrize("dtype,exact", [(np.int16, True), (np.int32, True), (np.float64, True), (np.float32, False)])
def test_audio_dtypes(dtype, exact):
    """integer and float64 recordings follow the reference's float64 arithmetic; float32 recordings are transformed in
    single precision by the reference (scipy.signal.stft keeps complex64) and in fp64 here: agreement to fp32 STFT noise
    where the spectrogram is above its clip floor"""
    from ava_amd import spec as sp
    p = dict(syn.FINCH_PARAMS)
    fs, T = p['fs'], p['num_time_bins']
    audio, _ = syn.recordings(n_files=2, fs=fs, seconds=1.0, dtype=dtype)
    if dtype == np.int32:
        audio = [a * 4096 for a in audio]
        p['spec_min_val'] += np.log(4096.0)
        p['spec_max_val'] += np.log(4096.0)
    if dtype in (np.float32, np.float64):
        p['spec_min_val'] -= np.log(32768.0)
        p['spec_max_val'] -= np.log(32768.0)
    on = np.array([0.1, 0.33, 0.5, 0.71])
    off = on + p['window_length']
    fidx = [0, 1, 0, 1]
    t1, t2 = np.maximum(0.0, on - 0.05), off + 0.05
    tts = np.linspace(on, off, T, axis=-1)
    dev = sp.get_spec_batch(sp.DeviceAudio(audio), fidx, t1, t2, p, fs, tts)
    want = _oracle_batch(audio, fidx, t1, t2, p, fs, tts)
    assert want.max() > 0.5
    if exact:
        _check(dev, want)
    else:
        assert np.abs(dev.cpu().numpy() - want).max() < 2e-3


@pytest.mark.parametrize("nperseg,noverlap,mel", [(256, 128, False), (512, 384, True), (1024, 0, False), (128, 64, True), (2048, 1024, False),
                                                  # lengths that are not a power of two (missing item 3): the reference hands any
                                                  # nperseg to scipy.signal.stft (utils.py:66-68); the device takes the direct transform
                                                  (400, 200, False), (500, 125, True), (1000, 500, False), (513, 256, False), (96, 32, True), (1500, 750, False)])
def test_stft_shapes_and_frequency_spacing(nperseg, noverlap, mel):
    from ava_amd import spec as sp
    p = dict(syn.FINCH_PARAMS)
    p.update(nperseg=nperseg, noverlap=noverlap, mel=mel, num_freq_bins=96, num_time_bins=80)
    fs = p['fs']
    audio, _ = syn.recordings(n_files=2, fs=fs, seconds=1.0)
    on = np.array([0.05, 0.4, 0.62])
    off = on + 0.15
    t1, t2 = np.maximum(0.0, on - 0.05), off + 0.05
    tts = np.linspace(on, off, 80, axis=-1)
    dev = sp.get_spec_batch(sp.DeviceAudio(audio), [0, 1, 1], t1, t2, p, fs, tts)
    want = _oracle_batch(audio, [0, 1, 1], t1, t2, p, fs, tts)
    assert dev.shape == (3, 96, 80)
    _check(dev, want)


def test_get_spec_mirror_defaults_and_options():
    """the one-window mirror with the reference's default arguments (target times from t1, t2, max_dur; time_stretch),
    explicit target frequencies, a custom fill value and remove_dc_offset=False"""
    from ava_amd import spec as sp
    p = dict(syn.FINCH_PARAMS)
    p['max_dur'] = 0.3
    fs = p['fs']
    audio = syn.recordings(n_files=1, fs=fs, seconds=1.0)[0][0]
    for kw in ({}, {"remove_dc_offset": False}, {"target_freqs": np.linspace(500.0, 9000.0, 64)},
               {"fill_value": 4.0}, {"max_dur": 0.5}):
        for stretch in (False, True):
            p['time_stretch'] = stretch
            got, flag = sp.get_spec(0.21, 0.39, audio, p, fs=fs, **kw)
            want, _ = so.get_spec(0.21, 0.39, audio, p, fs=fs, **kw)
            assert flag is True and got.dtype == np.float64 and got.shape == want.shape
            # without the mean subtraction int16 samples reach scipy.signal.stft as integers, which it transforms in
            # SINGLE precision (result_type(int16, complex64)); the device stays in fp64
            assert np.abs(got - want).max() <= (1e-6 if kw.get("remove_dc_offset") is False else ULP)
    p['time_stretch'] = False
    with pytest.warns(UserWarning, match="longer than max_dur"):
        sp.get_spec(0.1, 0.6, audio, p, fs=fs)
    with pytest.raises(AssertionError):
        sp.get_spec(0.3, 0.3, audio, p, fs=fs)
    q = dict(p)
    q['nperseg'] = 4096
    with pytest.raises(NotImplementedError):
        sp.get_spec(0.21, 0.39, audio, q, fs=fs)


@pytest.mark.parametrize("quantile", [0.0, 0.25, 0.5, 0.9, 0.99993, 1.0])
def test_within_syll_normalize(quantile):
    """utils.py:104-108: subtract np.quantile(spec, q) (numpy's default linear method), floor at zero, divide by the
    maximum + 1e-12; a window the reference answers with zeros stays zeros"""
    from ava_amd import spec as sp
    p = dict(syn.FINCH_PARAMS)
    p.update(within_syll_normalize=True, normalize_quantile=quantile, spec_min_val=1.0)
    fs, T = p['fs'], p['num_time_bins']
    audio, _ = syn.recordings(n_files=2, fs=fs, seconds=1.0)
    on = np.array([0.1, 0.33, 0.5, 0.71, 5.0])
    off = on + p['window_length']
    fidx = [0, 1, 0, 1, 0]
    t1, t2 = np.maximum(0.0, on - 0.05), off + 0.05
    tts = np.linspace(on, off, T, axis=-1)
    dev, mx = sp.get_spec_batch(sp.DeviceAudio(audio), fidx, t1, t2, p, fs, tts, return_max=True)
    want = _oracle_batch(audio, fidx, t1, t2, p, fs, tts)
    if quantile <= 0.9:                    # (at the saturated top quantiles everything is floored to zero)
        assert want[:4].max() > 0.99 and (want[:4] > 0).mean() > 0.01
    _check(dev, want)
    assert not dev[4].any().item()
    assert np.allclose(mx.cpu().numpy(), want.reshape(5, -1).max(axis=1).astype(np.float32), atol=ULP)
    with pytest.raises(ValueError):
        q = dict(p)
        q['normalize_quantile'] = 1.5
        sp.get_spec_batch(sp.DeviceAudio(audio), fidx, t1, t2, q, fs, tts)


@pytest.mark.parametrize("name", ["finch", "mouse"])
def test_dataset_seeded_batches_match_reference_selection_and_oracle_spectrograms(name):
    """DeviceWindowDataset with a seed: the windows are the ones the real FixedWindowDataset drew (golden) and their
    spectrograms are the oracle's"""
    from ava_amd import spec as sp
    G = load_golden("shotgun.npz")
    p = dict(CASES[name][0])
    audio, rois = _recordings(name)
    ds = sp.DeviceWindowDataset.from_arrays(audio, p['fs'], rois, p, dataset_length=64)
    specs, fidx, on, off = ds.__getitem__(list(range(16)), seed=11, return_seg_info=True)
    assert np.array_equal(np.array(fidx), G[name + ".plain.file_indices"])
    assert np.array_equal(np.array(on), G[name + ".plain.onsets"])
    assert np.array_equal(np.array(off), G[name + ".plain.offsets"])
    oracle = so.FixedWindowOracle(audio, p['fs'], rois, p, dataset_length=64)
    ospecs, ofidx, oon, _ = oracle.getitem(list(range(16)), seed=11)
    assert ofidx == fidx and oon == on
    _check(specs, np.stack(ospecs))
    one = ds.__getitem__(0, seed=13)
    assert one.shape == (p['num_freq_bins'], p['num_time_bins']) and one.is_cuda
    o1, f1, on1, _ = oracle.getitem([0], seed=13)
    assert on1[0] == float(G[name + ".single.onsets"][0])
    _check(one[None], np.stack(o1))


def test_dataset_redraws_silent_windows_like_the_reference():
    """min_spec_val: the first n candidates of the stream that are loud enough, in order (window_vae_dataset.py:229-231)"""
    from ava_amd import spec as sp
    p = dict(syn.FINCH_PARAMS)
    audio, rois = _recordings("finch")
    # the synthetic recordings alternate bursts and near silence: a threshold in between rejects a good share
    oracle_all = so.FixedWindowOracle(audio, p['fs'], rois, p)
    specs, _, _, _ = oracle_all.getitem(list(range(64)), seed=5)
    mx = np.sort([s.max() for s in specs])
    thr = float(0.5 * (mx[20] + mx[21]))
    oracle = so.FixedWindowOracle(audio, p['fs'], rois, p, min_spec_val=thr)
    ospecs, ofidx, oon, ooff = oracle.getitem(list(range(24)), seed=5)
    ds = sp.DeviceWindowDataset.from_arrays(audio, p['fs'], rois, p, min_spec_val=thr)
    dspecs, fidx, on, off = ds.__getitem__(list(range(24)), seed=5, return_seg_info=True)
    assert fidx == ofidx and on == oon and off == ooff
    _check(dspecs, np.stack(ospecs))
    assert float(dspecs.amax(dim=(1, 2)).min()) >= thr


def test_loader_feeds_the_train_step_on_the_device():
    """get_fixed_window_data_loaders' counterpart: batches are born on the device and go through train_epoch / the
    reference's train_loop call unchanged; two runs from the same state and seeds are bit-identical"""
    from ava_amd import spec as sp
    from ava_amd.vae import VAE
    p = dict(syn.FINCH_PARAMS)
    audio, rois = _recordings("finch")
    ds = sp.DeviceWindowDataset.from_arrays(audio, p['fs'], rois, p, dataset_length=96)
    loader = sp.DeviceWindowLoader(ds, batch_size=32)
    assert len(loader) == 3 and loader.dataset is ds
    batches = list(loader)
    assert [tuple(b.shape) for b in batches] == [(32, 128, 128)] * 3 and all(b.is_cuda and b.dtype == torch.float32 for b in batches)
    assert not torch.equal(batches[0], batches[1])                 # fresh windows every batch
    torch.manual_seed(0)
    model = VAE(save_dir="", z_dim=32, device_name="cuda")
    losses = [model.train_epoch(loader) for _ in range(4)]
    assert all(np.isfinite(l) for l in losses) and losses[-1] < losses[0]
    model.test_epoch(loader)
    lat = model.get_latent(loader)
    assert lat.shape == (96, 32) and np.isfinite(lat).all()


def test_windows_at_the_size_extension_feed_a_256x256_model():
    """configs[4]'s geometry end to end: 256 x 256 windows made on the device (the grid is just num_freq_bins x
    num_time_bins) through a VAE(x_shape=(256, 256)) train step"""
    from ava_amd import spec as sp
    from ava_amd.vae import VAE
    p = dict(syn.FINCH_PARAMS)
    p.update(num_freq_bins=256, num_time_bins=256)
    audio, rois = _recordings("finch")
    ds = sp.DeviceWindowDataset.from_arrays(audio, p['fs'], rois, p, dataset_length=16)
    specs, fidx, on, off = ds.__getitem__(list(range(4)), seed=21, return_seg_info=True)
    want, ofidx, oon, _ = so.FixedWindowOracle(audio, p['fs'], rois, p).getitem(list(range(4)), seed=21)
    assert fidx == ofidx and on == oon
    _check(specs, np.stack(want))
    model = VAE(save_dir="", z_dim=16, device_name="cuda", x_shape=(256, 256))
    loader = sp.DeviceWindowLoader(ds, batch_size=8)
    losses = [model.train_epoch(loader) for _ in range(3)]
    assert all(np.isfinite(l) for l in losses) and losses[-1] < losses[0]


def test_c_abi_argument_checks():
    from ava_amd import _lib
    lib = _lib.load()
    assert lib.ava_spec_workspace_bytes(4, 8000, 5000, 250, 16, 16, 0) == 0
    assert lib.ava_spec_workspace_bytes(4, 8000, 500, 250, 16, 16, 0) > 0        # any length in 64..2048 since round 6
    assert lib.ava_spec_workspace_bytes(4, 8000, 512, 512, 16, 16, 0) == 0
    nbytes = lib.ava_spec_workspace_bytes(4, 8000, 512, 256, 16, 16, 0)
    assert lib.ava_spec_workspace_bytes(4, 8000, 512, 256, 16, 16, 1) == nbytes + 4 * 256 * 8
    assert nbytes > 4 * 33 * 257 * 8
    dev = torch.device("cuda")
    audio = torch.zeros(16000, dtype=torch.int16, device=dev)
    off = torch.zeros(1, dtype=torch.int64, device=dev)
    ln = torch.full((1,), 16000, dtype=torch.int64, device=dev)
    fidx = torch.zeros(4, dtype=torch.int32, device=dev)
    t1 = torch.zeros(4, dtype=torch.float64, device=dev)
    t2 = torch.full((4,), 0.2, dtype=torch.float64, device=dev)
    tt = torch.zeros(4, 16, dtype=torch.float64, device=dev)
    tf = torch.linspace(400, 8000, 16, dtype=torch.float64, device=dev)
    win = torch.ones(512, dtype=torch.float64, device=dev)
    out = torch.empty(4, 16, 16, device=dev)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)

    def call(nperseg=512, noverlap=256, wsb=nbytes, smax=6.5, dtype=0, n=4, norm=0, q_lo=0, gamma=0.0):
        return lib.ava_get_spec_batch(audio.data_ptr(), dtype, off.data_ptr(), ln.data_ptr(), fidx.data_ptr(), t1.data_ptr(),
                                      t2.data_ptr(), tt.data_ptr(), n, 8000, 32000.0, nperseg, noverlap, win.data_ptr(),
                                      1.0 / 512, tf.data_ptr(), 16, 16, 2.0, smax, -1e12, 1, norm, q_lo, gamma,
                                      out.data_ptr(), None, ws.data_ptr(), wsb, _lib.stream())
    assert call() == 0
    torch.cuda.synchronize()
    assert not out.any().item()                # silence: log(1e-12) is far below spec_min_val
    assert call(nperseg=5000) == -1
    assert call(nperseg=48) == -1
    assert call(noverlap=512) == -1
    assert call(smax=2.0) == -1
    assert call(dtype=7) == -1
    assert call(n=0) == -1
    assert call(wsb=nbytes // 2) == -3
    assert call(norm=1) == -3                  # the normalising variant needs the larger workspace
    assert call(norm=1, q_lo=256) == -1
    assert call(norm=1, gamma=1.5) == -1
