"""cmdiad_amd -- MI355X-native hot path of CMDIAD behind the reference's own Python API.

Layout: csrc/ (HIP kernels + C-ABI, built into libcmdiad_hip.so), _native.py (ctypes
binding), runtime/ops (device plumbing over torch tensors), and the drop-in mirrors
``feature_extractors/``, ``models/``, ``utils/`` (same module paths, class names and
method signatures as the reference).
"""
__version__ = "0.1.0"
