"""blackbox_amd: MI355X-native per-image reduction hot path of BlackBOX.

Importing the package is cheap (no torch, no HIP library load); the HIP C-ABI
library is loaded on first use by :mod:`blackbox_amd._lib` and its absence is a
hard error for every stage function (there is no CPU fallback in the product
path -- the numpy restatement lives under ``oracle/`` and is test
infrastructure only).
"""
__version__ = '0.1.0'
