"""Oracle: restatement of the reference's oblique-slice geometry (numpy + scipy).

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Follows
/root/reference/interactive_unet/slicer.py: orientation vectors :141-156 (+ the
Rodrigues rotation :55-73 and the 15-decimal rounding/normalisation :22-35),
interpolation coordinates :94-115, slice extraction :196-228, write-back :230-257.
Written as pure functions of (rotation_vector, origin) instead of a stateful class.
"""
import numpy as np
from scipy import ndimage


def _unit(v):
    return v / np.linalg.norm(v)


def rotation_matrix(src, dst):
    """slicer.py:55-73."""
    src, dst = _unit(src), _unit(dst)
    v = np.cross(src, dst)
    s = np.linalg.norm(v)
    c = np.dot(src, dst)
    vm = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    return np.eye(3) + vm + np.dot(vm, vm) * ((1 - c) / (s ** 2))


def orientation_vectors(rotation_vector, eps=np.finfo(float).eps):
    """slicer.py:141-156 then :22-35.  Returns rot_vec, rot_mat, u, v, w."""
    rot_vec = np.asarray(rotation_vector).astype(float)
    rv = rot_vec + np.ones(3) * eps
    R = np.around(rotation_matrix(np.array([1, 0, 0]), rv), decimals=15)
    u, v, w = rv, R @ np.array([0, 1, 0]), R @ np.array([0, 0, 1])
    rot_vec, u, v, w = [_unit(np.around(t, decimals=15)) for t in (rot_vec, u, v, w)]
    return rot_vec, R, u, v, w


def interpolation_coords(u, v, w, origin, slice_width=256):
    """slicer.py:94-115: three coordinate grids [3 planes][3 xyz][sw][sw]."""
    start = int(-np.floor(slice_width / 2))
    r = np.linspace(start, start + slice_width - 1, slice_width)
    o = np.asarray(origin, float)[:, None, None]

    def plane(a, b):
        return a[:, None, None] * r[None, :, None] + b[:, None, None] * r[None, None, :] + o
    return np.array([plane(v, w), plane(u, w), plane(u, v)])


def get_slice(volume, u, v, w, origin, axis=0, slice_width=256, order=0, sampling_axis='random'):
    """slicer.py:196-228: bounding-box crop then map_coordinates (outside -> 0)."""
    coords = interpolation_coords(u, v, w, origin, slice_width)[axis]
    lower = np.floor(np.min(coords, axis=(1, 2))).astype(int)
    upper = np.ceil(np.max(coords, axis=(1, 2))).astype(int)
    lo = [max(0, int(l)) for l in lower]
    hi = [min(volume.shape[a], int(upper[a])) for a in range(3)]
    if sampling_axis in ('x', 'y', 'z'):
        hi['xyz'.index(sampling_axis)] += 1
    shift = np.array(lo)
    crop = volume[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
    return ndimage.map_coordinates(crop, coords - shift[:, None, None], order=order)


def update_volume(data, volume, u, v, w, origin, axis=0):
    """slicer.py:230-257: nearest-voxel scatter of a slice back into the volume."""
    coords = interpolation_coords(u, v, w, origin, data.shape[0])[axis]
    sc = np.round(coords).reshape(3, -1).astype(int)
    sc = np.array([np.clip(sc[i], 0, volume.shape[i] - 1) for i in range(3)])
    flat = data.ravel() if data.ndim == 2 else data.reshape(-1, data.shape[2])
    volume[sc[0], sc[1], sc[2]] = flat
    return volume
