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


class _ParentFallback:
    """Last entry of sys.meta_path: serves `feature_extractors`, `models`, `utils` as EMPTY packages when nothing else provides
    them (this package used without the reference's tree).  Being last, the reference's own directories -- namespace packages,
    found by the regular path finder whenever its tree is on sys.path at import time -- always win, so its `utils.misc`,
    `utils.heatmap`, ... stay importable whether install_dropin() ran before or after sys.path was set up."""

    parents = frozenset(name.rpartition(".")[0] for name in _DROPIN)

    def find_spec(self, name, path=None, target=None):
        if name not in self.parents:
            return None
        import importlib.machinery
        spec = importlib.machinery.ModuleSpec(name, None, is_package=True)
        spec.submodule_search_locations = []
        return spec


def install_dropin():
    """Redirect the reference's module paths (``feature_extractors.features``, ``models.models``, ...) to this
    package, submodule by submodule, so ``cmdiad_runner.py`` / ``hallucination_network_pretrain.py`` import the
    MI355X implementation without being edited.  Only the listed submodules are registered (``sys.modules``); their parent
    packages are NOT imported here: the import system resolves them when first needed -- to the reference's own directories if
    its tree is on ``sys.path`` by then (so ``utils.misc``, ``dataset.py`` keep working), to empty packages otherwise."""
    import importlib
    import sys
    for name, target in _DROPIN.items():
        mod = importlib.import_module(target)
        sys.modules[name] = mod
        parent, _, leaf = name.rpartition(".")
        if parent in sys.modules:
            setattr(sys.modules[parent], leaf, mod)
    if not any(isinstance(f, _ParentFallback) for f in sys.meta_path):
        sys.meta_path.append(_ParentFallback())
