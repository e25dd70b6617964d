"""MMD^2 estimators: the numpy oracle against goldens produced by the reference's own functions (CPU),
and the HIP kernels against both (GPU)."""
import numpy as np
import pytest

from conftest import load_golden
from ava_amd import synthetic as syn
from oracle import mmd_oracle as MO


def _sets():
    latent, cond = syn.latent_conditions()
    return latent, [np.argwhere(cond == c).flatten() for c in range(3)]


def relerr(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-300)


def test_oracle_matches_reference_golden():
    G = load_golden("mmd.npz")
    latent, idx = _sets()
    assert relerr(MO.estimate_median_sigma(latent, n=2000), G["sigma_default_seed"]) < 1e-14
    assert relerr(MO.estimate_median_sigma(latent, n=500, seed=7), G["sigma_seed7"]) < 1e-14
    sigma = float(G["sigma_default_seed"])
    for a, b in ((0, 1), (0, 2), (1, 2)):
        assert relerr(MO.estimate_mmd2(latent, idx[a].copy(), idx[b].copy(), sigma=sigma), G["quad_%d%d" % (a, b)]) < 1e-11
        assert relerr(MO.estimate_mmd2_linear_time(latent, idx[a], idx[b], sigma=sigma), G["lin_%d%d" % (a, b)]) < 1e-11
    i1, i2 = idx[0].copy(), idx[2].copy()
    assert relerr(MO.estimate_mmd2(latent, i1, i2, sigma=sigma, max_n=40, seed=3), G["quad_02_max40_seed3"]) < 1e-11
    assert np.array_equal(i1, G["i1_after_shuffle"]) and np.array_equal(i2, G["i2_after_shuffle"])
    assert relerr(MO.estimate_mmd2(latent, idx[1][:26].copy(), idx[1][26:].copy(), sigma=0.5 * sigma), G["quad_same"]) < 1e-10
    assert relerr(MO.estimate_mmd2(latent, idx[0].copy(), idx[1].copy()), G["quad_sigma_none"]) < 1e-11


@pytest.mark.gpu
def test_hip_matches_reference_golden_and_oracle():
    from ava_amd import mmd
    G = load_golden("mmd.npz")
    latent, idx = _sets()
    assert relerr(mmd.estimate_median_sigma(latent, n=2000), G["sigma_default_seed"]) < 1e-13
    assert relerr(mmd.estimate_median_sigma(latent, n=500, seed=7), G["sigma_seed7"]) < 1e-13
    sigma = float(G["sigma_default_seed"])
    for a, b in ((0, 1), (0, 2), (1, 2)):
        assert relerr(mmd._estimate_mmd2(latent, idx[a].copy(), idx[b].copy(), sigma=sigma), G["quad_%d%d" % (a, b)]) < 1e-11
        assert relerr(mmd._estimate_mmd2_linear_time(latent, idx[a], idx[b], sigma=sigma), G["lin_%d%d" % (a, b)]) < 1e-11
    i1, i2 = idx[0].copy(), idx[2].copy()
    assert relerr(mmd._estimate_mmd2(latent, i1, i2, sigma=sigma, max_n=40, seed=3), G["quad_02_max40_seed3"]) < 1e-11
    assert np.array_equal(i1, G["i1_after_shuffle"]) and np.array_equal(i2, G["i2_after_shuffle"])      # in-place shuffle kept
    assert relerr(mmd._estimate_mmd2(latent, idx[1][:26].copy(), idx[1][26:].copy(), sigma=0.5 * sigma), G["quad_same"]) < 1e-10
    assert relerr(mmd._estimate_mmd2(latent, idx[0].copy(), idx[1].copy()), G["quad_sigma_none"]) < 1e-11
    # the condition-by-condition matrix of _calculate_mmd2 (mmd_plots.py:395-418)
    _, cond = syn.latent_conditions()
    M, conds = mmd.mmd2_matrix(latent, cond, sigma=sigma)
    assert list(conds) == [0, 1, 2] and M.shape == (3, 3) and np.allclose(M, M.T) and M[0, 0] == 0
    assert relerr(M[0, 2], G["quad_02"]) < 1e-11 and relerr(M[1, 2], G["quad_12"]) < 1e-11
    with pytest.raises(NotImplementedError):
        mmd.mmd2_matrix(latent, cond, alg="cubic", sigma=sigma)
    with pytest.raises(ZeroDivisionError):
        mmd._estimate_mmd2(latent, idx[0][:1].copy(), idx[1].copy(), sigma=sigma)
    with pytest.raises(AssertionError):
        mmd._estimate_mmd2_linear_time(latent, idx[0][:1], idx[1], sigma=sigma)


@pytest.mark.gpu
@pytest.mark.parametrize("n1,n2,z", [(2, 2, 1), (65, 64, 32), (129, 300, 64), (1000, 777, 128), (2500, 2500, 32)])
def test_hip_pairwise_terms_vs_oracle_ragged_sizes(n1, n2, z):
    """tile edges (64-row tiles), the largest latent size and sizes where the pair count is in the millions: each of the
    three terms against the numpy oracle, 1e-11 relative (fp64 both sides, different summation order)."""
    import torch
    from ava_amd import mmd
    latent = (syn.gauss((n1 + n2) * z, 4711).reshape(n1 + n2, z) * 1.3).astype(np.float64)
    latent[n1:] += 0.4
    i1 = np.arange(n1)[::-1].copy()                         # non-trivial index lists
    i2 = n1 + np.arange(n2)
    sigma = 0.9 * np.sqrt(z)
    got = mmd._terms(mmd._latent_dev(latent), i1, i2, sigma)
    want = MO.estimate_mmd2_terms(latent, i1, i2, sigma)
    for g, w in zip(got, want):
        assert abs(g - w) <= 1e-11 * max(abs(w), 1e-3), (got, want)
    # determinism: fixed-order reductions
    again = mmd._terms(mmd._latent_dev(latent), i1, i2, sigma)
    assert np.array_equal(got, again)
    torch.cuda.synchronize()
