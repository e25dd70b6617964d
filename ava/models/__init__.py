"""``ava.models``: only ``vae`` is provided here; the other submodules (``vae_dataset``, ``window_vae_dataset``,
``utils`` ...) resolve to the reference package through the extended ``__path__`` (see ``ava/__init__.py``)."""
import pkgutil as _pkgutil

__path__ = _pkgutil.extend_path(__path__, __name__)
