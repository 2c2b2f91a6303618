"""cmdiad_amd -- MI355X-native hot path of CMDIAD behind the reference's own Python API.

Layout: csrc/ (HIP kernels + C-ABI, built into libcmdiad_hip.so), _native.py (ctypes
binding), runtime/ops (device plumbing over torch tensors), and the drop-in mirrors
``feature_extractors/``, ``models/``, ``utils/`` (same module paths, class names and
method signatures as the reference).
"""
__version__ = "0.1.0"

_DROPIN = {
    "feature_extractors.features": "cmdiad_amd.feature_extractors.features",
    "feature_extractors.multiple_features": "cmdiad_amd.feature_extractors.multiple_features",
    "models.models": "cmdiad_amd.models.models",
    "models.pointnet2_utils": "cmdiad_amd.models.pointnet2_utils",
    "models.hallucination_network": "cmdiad_amd.models.hallucination_network",
    "models.hrnet": "cmdiad_amd.models.hrnet",
    "utils.utils": "cmdiad_amd.utils.utils",
    "utils.lr_sched": "cmdiad_amd.utils.lr_sched",
    "utils.au_pro_util": "cmdiad_amd.utils.au_pro_util",
    "utils.mvtec3d_util": "cmdiad_amd.utils.mvtec3d_util",
}


def install_dropin():
    """Redirect the reference's module paths (``feature_extractors.features``, ``models.models``, ...) to this
    package, submodule by submodule, so ``cmdiad_runner.py`` / ``hallucination_network_pretrain.py`` import the
    MI355X implementation without being edited.  Parent packages that the reference provides (e.g. its own
    ``utils`` with ``utils.misc``) stay importable; missing parents are created as empty packages."""
    import importlib
    import sys
    import types
    for name, target in _DROPIN.items():
        mod = importlib.import_module(target)
        sys.modules[name] = mod
        parent, _, leaf = name.rpartition(".")
        try:
            pkg = importlib.import_module(parent)
        except ImportError:
            pkg = types.ModuleType(parent)
            pkg.__path__ = []
            sys.modules[parent] = pkg
        setattr(pkg, leaf, mod)
