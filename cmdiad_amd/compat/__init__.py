"""Stand-ins for the two CUDA-only wheels the reference imports at models/models.py:5-6 -- ``pointnet2_ops`` and
``knn_cuda`` -- backed by the HIP kernels of libcmdiad_hip.so.  ``install()`` registers them in ``sys.modules`` so that the
REFERENCE'S OWN models/models.py (Group, fps, PointTransformer ...) runs unchanged on an MI355X, where those wheels do not
exist.  (``cmdiad_amd.install_dropin()`` replaces the reference's modules wholesale instead; this is the narrower swap.)"""
import sys
import types


def install():
    from . import knn_cuda, pointnet2_utils
    pkg = types.ModuleType("pointnet2_ops")
    pkg.pointnet2_utils = pointnet2_utils
    pkg.__path__ = []
    sys.modules["pointnet2_ops"] = pkg
    sys.modules["pointnet2_ops.pointnet2_utils"] = pointnet2_utils
    sys.modules["knn_cuda"] = knn_cuda
