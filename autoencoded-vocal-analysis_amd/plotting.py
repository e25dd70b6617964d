"""Grid plot of spectrogram / reconstruction pairs used by ``VAE.visualize``.

The reference calls ``ava.plotting.grid_plot.grid_plot`` (``ava/plotting/grid_plot.py:48-95``)
from ``VAE.visualize`` (``ava/models/vae.py:515``).  Plotting is outside the accelerated
path; this is a small matplotlib equivalent with the same call signature and output file.
"""
import numpy as np


def grid_plot(specs, gap=3, vmin=0.0, vmax=1.0, ax=None, save_and_close=True, filename='temp.pdf'):
    """``specs``: array ``[rows, cols, height, width]``; rows are drawn bottom-up separated by
    ``gap`` (int or (vertical, horizontal)) blank pixels."""
    import matplotlib
    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt
    specs = np.asarray(specs)
    if specs.ndim != 4:
        raise ValueError("grid_plot expects a 4-d array, got shape %s" % (specs.shape,))
    gy, gx = (gap, gap) if isinstance(gap, int) else gap
    rows, cols, h, w = specs.shape
    canvas = np.full((rows * h + (rows - 1) * gy, cols * w + (cols - 1) * gx), np.nan)
    for r in range(rows):
        for c in range(cols):
            y0, x0 = r * (h + gy), c * (w + gx)
            canvas[y0:y0 + h, x0:x0 + w] = specs[rows - 1 - r, c]
    if ax is None:
        ax = plt.gca()
    ax.imshow(canvas, aspect='equal', origin='lower', interpolation='none', vmin=vmin, vmax=vmax)
    ax.axis('off')
    if save_and_close:
        plt.tight_layout()
        plt.savefig(filename)
        plt.close('all')
