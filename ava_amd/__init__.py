"""Importable alias for the ``autoencoded-vocal-analysis_amd/`` package directory.

The package directory name mandated for this repo contains hyphens and therefore
cannot be imported by name; this alias package points its ``__path__`` at that
directory so ``import ava_amd.vae`` resolves to
``autoencoded-vocal-analysis_amd/vae.py``.
"""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                         "autoencoded-vocal-analysis_amd")
__path__ = [_PKG_DIR]
__version__ = "0.1.0"
