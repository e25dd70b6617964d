"""MI355X-native VAE training hot path (import as ``ava_amd``; see ava_amd/__init__.py)."""
