"""Namespace shim: lets the reference's import line ``from ava.models.vae import X_SHAPE, X_DIM, VAE``
(``examples/mouse_sylls_mwe.py:22``) resolve to the MI355X-native implementation in this repo."""
