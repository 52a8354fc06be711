"""tfhe_aes_amd -- MI355X-native engine for the WoPBS S-Box hot path of FHE AES-128.

Layout:  csrc/ (HIP kernels + C ABI, host Client)   params.py   client.py   server.py   aes_clear.py
"""
from .params import PARAM_OPT, PARAM_TOY, WopbsParameters  # noqa: F401

__all__ = ["PARAM_OPT", "PARAM_TOY", "WopbsParameters"]
