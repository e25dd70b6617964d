"""CPU tests of the shotgun-spectrogram row: the oracle against the fixtures the REAL
FixedWindowDataset produced (tests/golden/make_golden.py: shotgun_case), against closed-form properties of get_spec, and
the product's host-side window selection against the same fixtures (no GPU involved)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ava_amd import synthetic as syn              # noqa: E402
from oracle import spec_oracle as so              # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "shotgun.npz"))
CASES = {"finch": (syn.FINCH_PARAMS, 2.0), "mouse": (syn.MOUSE_PARAMS, 1.0)}


def recorder_loudness(t1):
    """the loudness rule of the recorder the golden script handed to the reference as p['get_spec']"""
    return 0.25 + 0.75 * ((t1 * 1e3) % 1.0)


def _recordings(name):
    p, seconds = CASES[name]
    return syn.recordings(n_files=3, fs=p['fs'], seconds=seconds)


@pytest.mark.parametrize("name", ["finch", "mouse"])
@pytest.mark.parametrize("tag,min_spec_val,seed,n", [("plain", None, 11, 16), ("retry", 0.6, 12, 16), ("single", None, 13, 1)])
def test_oracle_window_selection_matches_reference(name, tag, min_spec_val, seed, n):
    p = dict(CASES[name][0])
    audio, rois = _recordings(name)
    calls = []

    def rec(t1, t2, a, pp, fs=32000, target_times=None):
        calls.append((t1, t2, len(a), fs, target_times[0], target_times[-1], len(target_times), recorder_loudness(t1)))
        return np.full((pp['num_freq_bins'], pp['num_time_bins']), recorder_loudness(t1)), True

    ds = so.FixedWindowOracle(audio, p['fs'], rois, p, dataset_length=64, min_spec_val=min_spec_val, get_spec_fn=rec)
    specs, fidx, on, off = ds.getitem(list(range(n)), seed=seed)
    k = "%s.%s." % (name, tag)
    assert np.array_equal(np.array(fidx), GOLD[k + "file_indices"])
    assert np.array_equal(np.array(on), GOLD[k + "onsets"])                 # bit-exact: same generator, same arithmetic
    assert np.array_equal(np.array(off), GOLD[k + "offsets"])
    assert np.array_equal(np.array(calls), GOLD[k + "calls"])               # rejected candidates included
    assert np.allclose(ds.file_weights, GOLD[name + ".file_weights"], rtol=0, atol=0)


@pytest.mark.parametrize("name", ["finch", "mouse"])
@pytest.mark.parametrize("tag,min_spec_val,seed,n", [("plain", None, 11, 16), ("retry", 0.6, 12, 16), ("single", None, 13, 1)])
def test_product_window_selection_matches_reference(name, tag, min_spec_val, seed, n, monkeypatch):
    """DeviceWindowDataset's vectorised draws (three uniforms per candidate, first n loud candidates kept) pick the
    windows the reference picks.  The device call is replaced by the recorder's loudness rule: host logic only."""
    from ava_amd import spec as sp
    p = dict(CASES[name][0])
    audio, rois = _recordings(name)
    ds = sp.DeviceWindowDataset.from_arrays(audio, p['fs'], rois, p, dataset_length=64, min_spec_val=min_spec_val,
                                            device="cpu")
    seen = []

    def fake_specs(file_index, onset, offset, shoulder, return_max):
        t1 = np.maximum(0.0, onset - shoulder)
        seen.append(np.stack([t1, offset + shoulder], axis=1))
        loud = torch.from_numpy(recorder_loudness(t1)).to(torch.float32)
        specs = loud[:, None, None].expand(len(onset), 4, 4).contiguous()
        return (specs, loud) if return_max else specs

    monkeypatch.setattr(ds, "_specs", fake_specs)
    index = list(range(n)) if n > 1 else 0
    specs, fidx, on, off = ds.__getitem__(index, seed=seed, return_seg_info=True)
    if n == 1:
        fidx, on, off = [fidx], [on], [off]
    k = "%s.%s." % (name, tag)
    assert np.array_equal(np.array(fidx), GOLD[k + "file_indices"])
    assert np.array_equal(np.array(on), GOLD[k + "onsets"])
    assert np.array_equal(np.array(off), GOLD[k + "offsets"])
    # the arguments get_spec receives (t1, t2) for the accepted windows are the reference's
    calls = GOLD[k + "calls"]
    accepted = calls[~(calls[:, 7] < (min_spec_val if min_spec_val is not None else -1.0))][:n]
    got = np.concatenate(seen)
    got_loud = recorder_loudness(got[:, 0])
    got = got[~(got_loud < (min_spec_val if min_spec_val is not None else -1.0))][:n]
    assert np.array_equal(got, accepted[:, :2])


def test_linspace_rows_equal_scalar_linspace():
    """the batched target times (np.linspace with array end points) are the per-window np.linspace of the reference"""
    on = GOLD["finch.plain.onsets"]
    off = GOLD["finch.plain.offsets"]
    rows = np.linspace(on, off, 128, axis=-1)
    for i in range(len(on)):
        assert np.array_equal(rows[i], np.linspace(on[i], off[i], 128))


def _plain_bilinear(t, f, L, xq, yq, fill):
    out = np.empty((len(yq), len(xq)))
    for a, y in enumerate(yq):
        for b, x in enumerate(xq):
            if x < t[0] or x > t[-1] or y < f[0] or y > f[-1]:
                out[a, b] = fill
                continue
            i = min(max(np.searchsorted(t, x, side='right') - 1, 0), len(t) - 2)
            j = min(max(np.searchsorted(f, y, side='right') - 1, 0), len(f) - 2)
            wx = (x - t[i]) / (t[i + 1] - t[i])
            wy = (y - f[j]) / (f[j + 1] - f[j])
            out[a, b] = (1 - wy) * ((1 - wx) * L[j, i] + wx * L[j, i + 1]) + wy * ((1 - wx) * L[j + 1, i] + wx * L[j + 1, i + 1])
    return out


def test_oracle_interpolation_is_bilinear_with_interp2d_bounds():
    """the FITPACK linear spline of the oracle equals textbook bilinear interpolation; out-of-range points get the fill
    value, border points do not (interp2d's rule)"""
    rng = np.random.default_rng(3)
    t = np.cumsum(rng.uniform(0.5, 1.5, 9))
    f = np.cumsum(rng.uniform(0.5, 1.5, 7))
    L = rng.standard_normal((7, 9))
    xq = np.sort(np.concatenate([rng.uniform(t[0] - 1, t[-1] + 1, 20), [t[0], t[-1], t[3]]]))
    yq = np.sort(np.concatenate([rng.uniform(f[0] - 1, f[-1] + 1, 20), [f[0], f[-1]]]))
    got = so._interp2d_linear(t, f, L, xq, yq, -7.0)
    want = _plain_bilinear(t, f, L, xq, yq, -7.0)
    assert np.abs(got - want).max() < 1e-13
    yin = (yq >= f[0]) & (yq <= f[-1])
    assert (got[:, xq < t[0]] == -7.0).all() and (got[np.ix_(yin, xq == t[0])] != -7.0).all()
    assert (got[np.ix_(yq == f[-1], (xq >= t[0]) & (xq <= t[-1]))] != -7.0).all()


def test_oracle_get_spec_properties():
    p = dict(syn.FINCH_PARAMS)
    fs = p['fs']
    n = int(0.5 * fs)
    tone = 3000.0
    audio = np.rint(8000 * np.sin(2 * np.pi * tone * np.arange(n) / fs)).astype(np.int16)
    tt = np.linspace(0.2, 0.32, 128)
    spec, flag = so.get_spec(0.15, 0.37, audio, p, fs=fs, target_times=tt)
    assert flag is True and spec.shape == (128, 128) and spec.min() >= 0.0 and spec.max() <= 1.0
    ridge = so.target_freqs_of(p)[np.argmax(spec.mean(axis=1))]
    assert abs(ridge - tone) < 100.0
    # utils.py:68-69: a slice shorter than nperseg gives zeros
    z, _ = so.get_spec(0.0, 0.01, audio, p, fs=fs, target_times=np.linspace(0.0, 0.01, 128))
    assert z.shape == (128, 128) and not z.any()
    z, _ = so.get_spec(0.6, 0.8, audio, p, fs=fs, target_times=np.linspace(0.6, 0.8, 128))      # beyond the recording
    assert not z.any()
    # target times outside the STFT's frames are filled with -1/EPSILON and clip to 0
    wide, _ = so.get_spec(0.15, 0.37, audio, p, fs=fs, target_times=np.linspace(0.0, 0.5, 128))
    assert not wide[:, :30].any() and wide[:, 60].any()
    # the DC offset does not matter when it is removed
    shifted, _ = so.get_spec(0.15, 0.37, audio + np.int16(500), p, fs=fs, target_times=tt)
    assert np.abs(shifted - spec).max() < 1e-9
