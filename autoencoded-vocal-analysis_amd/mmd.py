"""MMD^2 between conditions on the MI355X.

Host-side mirror of the estimator functions of the reference's ``ava/plotting/mmd_plots.py`` -- same names, same
arguments, same random-number use and error behaviour -- with the O(n^2) Python double loops replaced by the HIP
kernels of ``csrc/mmd.hip`` (fp64, like the reference's numpy arithmetic):

=============================  ==========================================  =============================
reference (mmd_plots.py)       here                                        C entry point
=============================  ==========================================  =============================
``estimate_median_sigma``      :func:`estimate_median_sigma`  (:450-474)   ``ava_pair_sqdist``
``_estimate_mmd2``             :func:`_estimate_mmd2`         (:255-296)   ``ava_mmd2``
``_estimate_mmd2_linear_time`` :func:`_estimate_mmd2_linear_time` (:299-312) ``ava_mmd2_linear``
loop of ``_calculate_mmd2``    :func:`mmd2_matrix`            (:395-418)   the above per condition pair
=============================  ==========================================  =============================

``install()`` swaps the three estimator functions of an imported ``ava.plotting.mmd_plots`` for these, so the
reference's plotting functions (``mmd_matrix_plot_DC`` ...) run unchanged on top of them.  There is no CPU fallback:
without the HIP library / a GPU every function raises ``AvaHipError``.
"""
import numpy as np
import torch

from . import _lib

EPSILON = 1e-8          # ava/plotting/mmd_plots.py:34

__all__ = ["estimate_median_sigma", "_estimate_mmd2", "_estimate_mmd2_linear_time", "mmd2_matrix", "install", "EPSILON"]


def _device():
    if not torch.cuda.is_available():
        raise _lib.AvaHipError("the MMD kernels only run on an MI355X: there is no CPU fallback in this package")
    return torch.device("cuda", torch.cuda.current_device())


def _latent_dev(latent):
    """[N, z] float64 on the device (accepts the numpy array VAE.get_latent returns, or a tensor already there)."""
    dev = _device()
    t = latent if torch.is_tensor(latent) else torch.from_numpy(np.ascontiguousarray(latent, dtype=np.float64))
    t = t.to(device=dev, dtype=torch.float64).contiguous()
    if t.dim() != 2 or t.shape[1] < 1 or t.shape[1] > 128:
        raise ValueError("latent must be [N, z] with 1 <= z <= 128")
    return t


def _index_dev(idx, n_rows):
    a = np.ascontiguousarray(np.asarray(idx), dtype=np.int64)
    if a.size and (a.min() < -n_rows or a.max() >= n_rows):
        raise IndexError("index out of bounds for latent with %d rows" % n_rows)      # numpy would raise the same
    a = np.where(a < 0, a + n_rows, a)
    return torch.from_numpy(a).to(_device())


def estimate_median_sigma(latent, n=10000, seed=42):
    """Median pairwise Euclidean distance of ``n`` random pairs (mmd_plots.py:450-474): the same ``np.random`` draws
    in the same order (``randint`` twice per pair), distances on the device, median on the host."""
    L = _latent_dev(latent)
    np.random.seed(seed)
    pairs = np.random.randint(len(L), size=2 * n).reshape(n, 2)          # == n x (randint, randint), same stream
    np.random.seed(None)
    a = torch.from_numpy(np.ascontiguousarray(pairs[:, 0])).to(L.device)
    b = torch.from_numpy(np.ascontiguousarray(pairs[:, 1])).to(L.device)
    out = torch.empty(n, dtype=torch.float64, device=L.device)
    _lib.check(_lib.load().ava_pair_sqdist(L.data_ptr(), L.shape[1], a.data_ptr(), b.data_ptr(), n, out.data_ptr(),
                                           _lib.stream()), "ava_pair_sqdist")
    return np.sqrt(np.median(out.cpu().numpy()) + EPSILON)


def _terms(L, i1, i2, sigma):
    n1, n2 = len(i1), len(i2)
    if n1 * (n1 - 1) == 0 or n2 * (n2 - 1) == 0:
        raise ZeroDivisionError("division by zero")                       # the reference's 2/(n*(n-1))
    lib = _lib.load()
    d1, d2 = _index_dev(i1, len(L)), _index_dev(i2, len(L))
    nbytes = lib.ava_mmd2_workspace_bytes(n1, n2)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=L.device)
    out = torch.empty(4, dtype=torch.float64, device=L.device)
    _lib.check(lib.ava_mmd2(L.data_ptr(), L.shape[1], d1.data_ptr(), n1, d2.data_ptr(), n2, float(sigma),
                            out.data_ptr(), ws.data_ptr(), nbytes, _lib.stream()), "ava_mmd2")
    return out.cpu().numpy()


def _estimate_mmd2(latent, i1, i2, sigma=None, max_n=None, seed=None):
    """Unbiased quadratic-time MMD^2 estimate (Gretton et al. 2012; mmd_plots.py:255-296).  Like the reference, with
    ``max_n`` the index arrays are shuffled IN PLACE under ``np.random.seed(seed)`` and truncated."""
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
    return float(_terms(_latent_dev(latent), i1, i2, sigma)[3])


def _estimate_mmd2_linear_time(latent, i1, i2, sigma=None):
    """Linear-time estimate (mmd_plots.py:299-312)."""
    if sigma is None:
        sigma = estimate_median_sigma(latent)
    n = min(len(i1), len(i2))
    m = n // 2
    assert m > 0
    L = _latent_dev(latent)
    lib = _lib.load()
    d1, d2 = _index_dev(i1[:2 * m], len(L)), _index_dev(i2[:2 * m], len(L))
    ws = torch.empty((min((m + 255) // 256, 1024) + 8) * 8, dtype=torch.uint8, device=L.device)
    out = torch.empty(1, dtype=torch.float64, device=L.device)
    _lib.check(lib.ava_mmd2_linear(L.data_ptr(), L.shape[1], d1.data_ptr(), d2.data_ptr(), m, float(sigma), out.data_ptr(),
                                   ws.data_ptr(), ws.numel(), _lib.stream()), "ava_mmd2_linear")
    return float(out.item())


def mmd2_matrix(latent, condition, alg='quadratic', sigma=None, max_n=None):
    """The condition-by-condition loop of ``_calculate_mmd2`` (mmd_plots.py:395-418, serial branch): returns
    ``(result [n, n], all_conditions)`` with ``result[i, j] = result[j, i] = MMD^2(condition i, condition j)``.  The
    latent means are uploaded once for all pairs."""
    condition = np.asarray(condition)
    all_conditions = np.unique(condition)
    n = len(all_conditions)
    result = np.zeros((n, n))
    if sigma is None:
        sigma = estimate_median_sigma(latent)
    L = _latent_dev(latent)
    for i in range(n - 1):
        for j in range(i + 1, n):
            i1 = np.argwhere(condition == all_conditions[i]).flatten()
            i2 = np.argwhere(condition == all_conditions[j]).flatten()
            if alg == 'linear':
                temp = _estimate_mmd2_linear_time(L, i1, i2, sigma=sigma)
            elif alg == 'quadratic':
                temp = _estimate_mmd2(L, i1, i2, sigma=sigma, max_n=max_n)
            else:
                raise NotImplementedError
            result[i, j] = temp
            result[j, i] = temp
    return result, all_conditions


def install(module=None):
    """Point ``ava.plotting.mmd_plots``'s estimators at this module (call after importing the reference package)."""
    if module is None:
        import ava.plotting.mmd_plots as module
    module.estimate_median_sigma = estimate_median_sigma
    module._estimate_mmd2 = _estimate_mmd2
    module._estimate_mmd2_linear_time = _estimate_mmd2_linear_time
    return module
