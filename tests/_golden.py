"""Helpers shared by the parity tests: load a committed fixture."""
import glob
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def T(a, device="cpu"):
    return torch.tensor(np.asarray(a), dtype=torch.float64, device=device)


def cores(d, prefix="core", device="cpu"):
    out, i = [], 0
    while f"{prefix}{i}" in d:
        out.append(T(d[f"{prefix}{i}"], device))
        i += 1
    return out


def inputs(d, device="cpu"):
    if int(d["x_is_list"]):
        xs, i = [], 0
        while f"x{i}" in d:
            xs.append(T(d[f"x{i}"], device))
            i += 1
        return xs
    return T(d["x"], device)


def kwargs(d):
    out = {}
    for k, v in d.items():
        if k.startswith("kw_"):
            v = v.tolist() if v.ndim else v.item()
            out[k[3:]] = v
    return out


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()
