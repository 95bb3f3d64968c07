"""Drop-in for interactive_unet/slicer.py (row a19 of SURVEY.md section 8).  Same class, attributes and method
signatures; `get_slice` on a uint8 volume that is a CUDA tensor runs on the device (libiunet: iunet_slice_gather,
bit-exact with the scipy path for orders 0 and 1 -- SURVEY 8f "Slicer on device"), numpy volumes take the reference's
scipy path;
geometry written as small pure helpers (orientation from a rotation vector by Rodrigues'
formula, plane grids, nearest/linear sampling through scipy) with the reference's numeric
conventions: 15-decimal rounding then normalisation (slicer.py:22-35), eps-shifted rotation
vector (slicer.py:141-147), grids centred at -floor(sw/2) (slicer.py:99-102).
"""
import numpy as np
from scipy import ndimage

_E1, _E2, _E3 = np.eye(3)


def _unit(v):
    return v / np.linalg.norm(v)


def _rodrigues(src, dst):
    a, b = _unit(src), _unit(dst)
    k = np.cross(a, b)
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + K + (K @ K) * ((1 - np.dot(a, b)) / (np.linalg.norm(k) ** 2))


class Slicer(object):

    def __init__(self, volume_shape=[512, 512, 512]):
        self.volume_shape = np.array(volume_shape)
        self.update_orientation_vectors(np.array([1, 0, 0]))
        self.origin = self.volume_shape / 2
        self._normalize_vectors()
        self.sampling_axis = 'random'

    def _normalize_vectors(self):
        self.rot_vec, self.u, self.v, self.w = [_unit(np.around(t, decimals=15))
                                                for t in (self.rot_vec, self.u, self.v, self.w)]

    def _generate_uniformly_random_unit_vector(self, ndim=3):
        while True:
            u = np.random.normal(size=ndim)
            if np.linalg.norm(u) >= 0.0001:
                return _unit(u)

    def _compute_rotation_matrix_from_vectors(self, src, dst):
        return _rodrigues(src, dst)

    def to_dict(self):
        return {'RotationVector': self.rot_vec.tolist(), 'RotationMatrix': self.rot_mat.tolist(),
                'Origin': self.origin.tolist(), 'VolumeShape': self.volume_shape.tolist()}

    def from_dict(self, slicer_dict):
        self.rot_vec = np.array(slicer_dict['RotationVector'])
        self.rot_mat = np.array(slicer_dict['RotationMatrix'])
        self.origin = np.array(slicer_dict['Origin'])
        self.volume_shape = np.array(slicer_dict['VolumeShape'])
        self.update_orientation_vectors(self.rot_vec)

    def get_interpolation_coords(self, slice_width=256):
        start = int(-np.floor(slice_width / 2))
        r = np.linspace(start, start + slice_width - 1, slice_width)
        o = self.origin[:, None, None]
        grid = lambda a, b: a[:, None, None] * r[None, :, None] + b[:, None, None] * r[None, None, :] + o
        return np.array([grid(self.v, self.w), grid(self.u, self.w), grid(self.u, self.v)])

    def get_origin_candidates(self, volume):
        classes = np.unique(volume)
        candidates = [np.argwhere(volume == c) for c in classes]
        counts = np.array([c.shape[0] for c in candidates])
        weights = np.max(counts) / counts
        return candidates, weights / np.sum(weights)

    def update_orientation_vectors(self, rotation_vector, eps=np.finfo(float).eps):
        self.rot_vec = rotation_vector.astype(float)
        shifted = rotation_vector.astype(float) + np.ones(3) * eps
        self.rot_mat = np.around(_rodrigues(_E1, shifted), decimals=15)
        self.u, self.v, self.w = shifted, self.rot_mat @ _E2, self.rot_mat @ _E3
        self._normalize_vectors()

    def randomize(self, candidates=None, class_weights=None, origin_shift_range=0.8, sampling_mode='random',
                  sampling_axis='random'):
        if sampling_mode == 'grid':
            self.sampling_axis = 'xyz'[np.random.randint(3)] if sampling_axis == 'random' else sampling_axis
            rotation_vector = {'x': _E1, 'y': _E2, 'z': _E3}[self.sampling_axis].astype(int)
        elif sampling_mode == 'random':
            rotation_vector = self._generate_uniformly_random_unit_vector()
        else:
            raise ValueError('sampling_mode must be either "random" or "grid".')
        self.update_orientation_vectors(rotation_vector)
        if candidates is not None:
            n = len(candidates)
            if class_weights is None:
                class_weights = np.ones(n) / n
            c = np.random.choice(np.arange(n), p=class_weights)
            self.origin = candidates[c][np.random.randint(candidates[c].shape[0])]
        else:
            self.origin = np.random.rand(3) * self.volume_shape * origin_shift_range \
                + self.volume_shape * (1 - origin_shift_range)
        return self.rot_vec, self.u, self.v, self.w, self.origin

    def _plane_vectors(self, axis):
        return ((self.v, self.w), (self.u, self.w), (self.u, self.v))[axis]

    def _get_slice_device(self, volume, axis, slice_width, order):
        """slicer.py:196-228 on a resident uint8 volume: only the 4 corner points are computed on the host (each
        coordinate is monotone in r_i and r_j, rounding included, so the bounding box of the grid is the bounding box
        of its corners); the gather runs in one kernel and the slice stays on the device."""
        import ctypes
        import torch
        from . import _native as nv
        if volume.dtype != torch.uint8 or volume.dim() != 3:
            raise NotImplementedError('device get_slice handles uint8 [Z, Y, X] volumes')
        if order not in (0, 1):
            raise NotImplementedError('device get_slice implements spline orders 0 and 1 (the reference uses those)')
        volume = volume.contiguous()
        a, b = self._plane_vectors(axis)
        start = int(-np.floor(slice_width / 2))
        ends = np.array([start, start + slice_width - 1], dtype=float)
        corners = a[:, None, None] * ends[None, :, None] + b[:, None, None] * ends[None, None, :] \
            + self.origin[:, None, None]
        shape = np.array(volume.shape)
        lo = np.maximum(np.floor(corners.min(axis=(1, 2))).astype(int), 0)
        hi = np.minimum(np.ceil(corners.max(axis=(1, 2))).astype(int), shape)
        if self.sampling_axis in ('x', 'y', 'z'):
            hi['xyz'.index(self.sampling_axis)] += 1
        hi = np.minimum(hi, shape)                             # what the slicing of the crop does
        lo = np.minimum(lo, shape)
        length = np.maximum(hi - lo, 0)
        out = torch.empty((slice_width, slice_width), dtype=torch.uint8, device=volume.device)
        geom = (ctypes.c_double * 9)(*[float(t) for t in (*a, *b, *np.asarray(self.origin, float))])
        with torch.cuda.device(volume.device):
            nv.call('iunet_slice_gather', nv.ptr(volume), int(shape[0]), int(shape[1]), int(shape[2]), geom,
                    nv.int_array(lo), nv.int_array(length), int(slice_width), start, int(order), nv.ptr(out), nv.stream())
        return out

    def get_slice(self, volume, axis=0, slice_width=256, order=0):
        if hasattr(volume, 'is_cuda') and volume.is_cuda:
            return self._get_slice_device(volume, axis, slice_width, order)
        coords = self.get_interpolation_coords(slice_width=slice_width)[axis]
        lo = np.maximum(np.floor(coords.min(axis=(1, 2))).astype(int), 0)
        hi = np.minimum(np.ceil(coords.max(axis=(1, 2))).astype(int), np.array(volume.shape))
        if self.sampling_axis in ('x', 'y', 'z'):          # keep axis-aligned slices non-empty (slicer.py:215-221)
            hi['xyz'.index(self.sampling_axis)] += 1
        crop = volume[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
        return ndimage.map_coordinates(crop, coords - lo[:, None, None], order=order)

    def _update_volume_device(self, data, volume, axis):
        """slicer.py:230-257 on a resident uint8 volume ([Z, Y, X] or [Z, Y, X, C]): one scatter on the device, duplicates
        resolved like numpy's assignment (the last pixel in row-major order wins)."""
        import ctypes
        import torch
        from . import _native as nv
        if volume.dtype != torch.uint8 or volume.dim() not in (3, 4) or not volume.is_contiguous():
            raise NotImplementedError('device update_volume handles contiguous uint8 [Z, Y, X] / [Z, Y, X, C] volumes')
        d = data if torch.is_tensor(data) else torch.from_numpy(np.ascontiguousarray(data))
        d = d.to(device=volume.device, dtype=torch.uint8).contiguous()
        C = 1 if volume.dim() == 3 else int(volume.shape[3])
        sw = int(d.shape[0])
        if d.numel() != sw * sw * C:
            raise ValueError(f'shape mismatch: data {tuple(d.shape)} for a volume with {C} channel(s)')
        a, b = self._plane_vectors(axis)
        start = int(-np.floor(sw / 2))
        geom = (ctypes.c_double * 9)(*[float(t) for t in (*a, *b, *np.asarray(self.origin, float))])
        ws = torch.empty(int(nv.lib().iunet_slice_scatter_workspace_bytes(sw)), dtype=torch.uint8, device=volume.device)
        with torch.cuda.device(volume.device):
            nv.call('iunet_slice_scatter', nv.ptr(volume), int(volume.shape[0]), int(volume.shape[1]), int(volume.shape[2]), C,
                    geom, sw, start, nv.ptr(d), nv.ptr(ws), nv.stream())
        return volume

    def update_volume(self, data, volume, axis=0):
        if hasattr(volume, 'is_cuda') and volume.is_cuda:
            return self._update_volume_device(data, volume, axis)
        coords = self.get_interpolation_coords(slice_width=data.shape[0])[axis]
        idx = np.round(coords).reshape(3, -1).astype(int)
        idx = np.array([np.clip(idx[i], 0, volume.shape[i] - 1) for i in range(3)])
        flat = data.ravel() if data.ndim == 2 else data.reshape(-1, data.shape[2])
        volume[idx[0], idx[1], idx[2]] = flat
        return volume

    def shift_origin(self, shift_amount=[0, 0, 0]):
        self.origin += np.dot(self.rot_mat, shift_amount)
