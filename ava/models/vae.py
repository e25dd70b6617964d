"""Drop-in for the reference module ``ava/models/vae.py``: same names, MI355X-native implementation."""
from ava_amd.vae import VAE, X_SHAPE, X_DIM  # noqa: F401
