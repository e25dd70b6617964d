"""Namespace shim: lets the reference's import line ``from ava.models.vae import X_SHAPE, X_DIM, VAE``
(``examples/mouse_sylls_mwe.py:22``) resolve to the MI355X-native implementation in this repo, while every
OTHER ``ava.*`` module (``ava.data``, ``ava.preprocessing``, ``ava.segmenting``, ``ava.plotting``,
``ava.models.vae_dataset`` ...; ``examples/mouse_sylls_mwe.py:20-30``) keeps resolving to the reference
package found later on ``sys.path``.

This directory only holds ``models/vae.py``.  ``pkgutil.extend_path`` appends the ``ava/`` directories of
every later ``sys.path`` entry to ``__path__``; this one stays first, so ``ava.models.vae`` here wins and
nothing else is shadowed.  The reference's own ``ava/__init__.py`` is not executed (only its ``__version__``
would be lost); it is mirrored below when a reference package is present.
"""
import os as _os
import pkgutil as _pkgutil

__path__ = _pkgutil.extend_path(__path__, __name__)

for _p in __path__[1:]:
    _init = _os.path.join(_p, "__init__.py")
    if _os.path.exists(_init):
        # the reference's ava/__init__.py only defines metadata (``__version__``, ava/__init__.py:33); pick it up
        # without executing reference code
        with open(_init) as _f:
            for _line in _f:
                if _line.startswith("__version__"):
                    __version__ = _line.split("=", 1)[1].strip().strip("\"'")
        break
else:
    __version__ = "0.3.1+mi355x"
