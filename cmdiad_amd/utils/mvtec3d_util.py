"""The reference's utils/mvtec3d_util.py surface (its dataset.py does `from utils.mvtec3d_util import *`, dataset.py:9, and
calls all four at dataset.py:106-110,153-157,228-231): with the drop-in installed that import resolves HERE, so every name must
exist.  Host-side, one sample at a time, before the hot path starts; no GPU work.

  organized_pc_to_unorganized_pc   mvtec3d_util.py:5-6    [H,W,3] -> [H*W,3]
  read_tiff_organized_pc           mvtec3d_util.py:9-11   the MVTec 3D-AD xyz tiff as the array tifffile returns
  resize_organized_pc              mvtec3d_util.py:14-22  nearest-neighbour resize, [3,h,w] tensor or [h,w,3] array
  organized_pc_to_depth_map        mvtec3d_util.py:25-26  the z channel
"""
import numpy as np
import torch


def organized_pc_to_unorganized_pc(organized_pc):
    return organized_pc.reshape(organized_pc.shape[0] * organized_pc.shape[1], organized_pc.shape[2])


def read_tiff_organized_pc(path):
    try:
        import tifffile
    except ImportError as exc:   # (the reference imports it at module import; here only the reader needs it)
        raise ImportError("read_tiff_organized_pc needs the `tifffile` package (MVTec 3D-AD xyz tiffs are 3-channel float32)") from exc
    return tifffile.imread(path)


def _nearest_index(n_in, n_out):
    """Source index of every output position under torch's mode='nearest': floor(dst * scale) with scale = n_in / n_out held
    in float32 (as the CPU and GPU kernels of F.interpolate hold it), clamped to the last row."""
    scale = np.float32(n_in) / np.float32(n_out)
    src = np.floor(np.arange(n_out, dtype=np.float32) * scale).astype(np.int64)
    return np.minimum(src, n_in - 1)


def resize_organized_pc(organized_pc, target_height=224, target_width=224, tensor_out=True):
    """[H,W,C] array (or tensor) -> [C,h,w] contiguous tensor (tensor_out) or [h,w,C] numpy array: nearest neighbour, no
    interpolation of coordinates (a resized point is a point of the scan, invalid points stay exact zeros)."""
    pc = organized_pc if isinstance(organized_pc, torch.Tensor) else torch.as_tensor(np.asarray(organized_pc))
    rows = torch.from_numpy(_nearest_index(pc.shape[0], target_height))
    cols = torch.from_numpy(_nearest_index(pc.shape[1], target_width))
    out = pc.index_select(0, rows).index_select(1, cols)             # [h,w,C]
    if tensor_out:
        return out.permute(2, 0, 1).contiguous()
    return out.contiguous().numpy()


def organized_pc_to_depth_map(organized_pc):
    return organized_pc[:, :, 2]
