"""arvae_amd: MI355X-native AR-VAE training path (HIP kernels behind a C-ABI).

Importable as ``arvae_amd`` (see /arvae_amd.py).  Sub-modules are imported
lazily so that host-only utilities (``synthetic``) work without torch/HIP.
"""
__version__ = "0.1.0"
