"""CPU oracle for the MMD^2 estimators -- TEST INFRASTRUCTURE ONLY (see oracle/vae_oracle.py's header: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/).

numpy float64 restatement of ``ava/plotting/mmd_plots.py``: ``_estimate_mmd2`` (:255-296),
``_estimate_mmd2_linear_time`` (:299-312), ``estimate_median_sigma`` (:450-474).  Same arithmetic as the reference's
double loops (direct differences, ``exp(A * dist)``), vectorised over the pairs; summation order differs, so results
agree to float64 rounding (pinned at 1e-12 relative by tests/golden/mmd.npz, which tests/golden/make_golden.py writes
by calling the reference's own functions)."""
import numpy as np

EPSILON = 1e-8          # mmd_plots.py:34


def estimate_median_sigma(latent, n=10000, seed=42):
    """mmd_plots.py:450-474: n random index pairs (randint twice per pair), median squared distance."""
    np.random.seed(seed)
    arr = np.zeros(n)
    for i in range(n):
        i1, i2 = np.random.randint(len(latent)), np.random.randint(len(latent))
        arr[i] = np.sum(np.power(latent[i1] - latent[i2], 2))
    np.random.seed(None)
    return np.sqrt(np.median(arr) + EPSILON)


def _gram_sum(X, Y, A, upper_only):
    d = ((X[:, None, :] - Y[None, :, :]) ** 2).sum(axis=2)
    K = np.exp(A * d)
    if upper_only:
        return np.triu(K, k=1).sum()
    return K.sum()


def estimate_mmd2_terms(latent, i1, i2, sigma):
    """The three normalised terms of mmd_plots.py:276-295 and their combination."""
    A = -0.5 / (sigma ** 2)
    n1, n2 = len(i1), len(i2)
    X, Y = latent[np.asarray(i1)], latent[np.asarray(i2)]
    t1 = _gram_sum(X, X, A, True) * (2 / (n1 * (n1 - 1)))
    t2 = _gram_sum(Y, Y, A, True) * (2 / (n2 * (n2 - 1)))
    t3 = _gram_sum(X, Y, A, False) * (2 / (n1 * n2))
    return t1, t2, t3, t1 + t2 - t3


def estimate_mmd2(latent, i1, i2, sigma=None, max_n=None, seed=None):
    """mmd_plots.py:255-296 including the in-place shuffle / truncation under ``max_n``."""
    if sigma is None:
        sigma = estimate_median_sigma(latent)
    n1, n2 = len(i1), len(i2)
    if max_n is not None:
        np.random.seed(seed)
        n1, n2 = min(max_n, n1), min(max_n, n2)
        if n1 < len(i1):
            np.random.shuffle(i1)
            i1 = i1[:n1]
        if n2 < len(i2):
            np.random.shuffle(i2)
            i2 = i2[:n2]
        np.random.seed(None)
    return estimate_mmd2_terms(latent, i1, i2, sigma)[3]


def estimate_mmd2_linear_time(latent, i1, i2, sigma=None):
    """mmd_plots.py:299-312."""
    if sigma is None:
        sigma = estimate_median_sigma(latent)
    A = -0.5 / (sigma ** 2)
    n = min(len(i1), len(i2))
    m = n // 2
    assert m > 0
    x1, y1 = latent[np.asarray(i1[0:2 * m:2])], latent[np.asarray(i2[0:2 * m:2])]
    x2, y2 = latent[np.asarray(i1[1:2 * m:2])], latent[np.asarray(i2[1:2 * m:2])]
    k = lambda a, b: np.exp(A * ((a - b) ** 2).sum(axis=1))
    return float((k(x1, x2) + k(y1, y2) - k(x1, y2) - k(x2, y1)).sum() / m)
