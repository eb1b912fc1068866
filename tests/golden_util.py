import glob
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    w = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    i = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in::")}
    o = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("out::")}
    return w, i, o


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "_*.npz")))


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
