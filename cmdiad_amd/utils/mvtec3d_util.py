"""Reference utils/mvtec3d_util.py:5-6: [H,W,3] -> [H*W,3] (the tiff readers stay out of scope)."""


def organized_pc_to_unorganized_pc(organized_pc):
    return organized_pc.reshape(organized_pc.shape[0] * organized_pc.shape[1], organized_pc.shape[2])
