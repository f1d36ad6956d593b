"""Kernel-point dispositions (Predator_APR/kernels/kernel_points.py:388-470 `load_kernels`).

The reference reads the optimised 15-point disposition from
`kernels/dispositions/k_015_center_3D.ply`, applies a RANDOM z-rotation and N(0, 0.01) noise
from the global NumPy RNG, and scales by the radius; the result is stored in the module as the
non-trainable parameter `kernel_points` (so trained checkpoints carry their own).  The 15x3
table ships here as data (`k_015_center_3D.npy`, extracted from that .ply); the same NumPy calls
are made in the same order, so under the same `np.random.seed` the points equal the reference's.
"""
import os

import numpy as np

_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "k_015_center_3D.npy")


def load_kernels(radius, num_kpoints, dimension, fixed, lloyd=False):
    if num_kpoints != 15 or dimension != 3 or fixed != 'center':
        raise NotImplementedError("only the k_015_center_3D disposition used by the APR configs is shipped")
    kernel_points = np.load(_TABLE)
    theta = np.random.rand() * 2 * np.pi
    c, s = np.cos(theta), np.sin(theta)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float32)
    kernel_points = kernel_points + np.random.normal(scale=0.01, size=kernel_points.shape)
    kernel_points = radius * kernel_points
    kernel_points = np.matmul(kernel_points, R)
    return kernel_points.astype(np.float32)
