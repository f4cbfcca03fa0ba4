"""Checkpoint loading with the reference's conventions (video_sci_proxgrad.py:210-227):
pickle dict {'solver_state_dict', 'epoch', ...} whose keys are `nonlinear_op.<net keys>`, optional
DataParallel 'module.' prefixes; bare FFDNet state dicts (`net_gray.pth`, keys under 'module.');
and the plain `.npz` tensor archives shipped in deqsci_amd/weights/.  Unlike the reference, a
missing path is an error (the reference silently keeps random weights, :211)."""
import os

import numpy as np
import torch

WEIGHTS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "weights")


def _strip(k, prefixes=("module.",)):
    for p in prefixes:
        if k.startswith(p):
            k = k[len(p):]
    return k


def read_state_dict(path):
    """-> (state_dict with 'module.' stripped, epoch or None)"""
    if not os.path.exists(path):
        raise FileNotFoundError(f"checkpoint {path!r} not found (the reference would silently run with "
                                "random weights here; this build refuses)")
    if path.endswith(".npz"):
        arc = np.load(path)
        sd = {k: torch.from_numpy(arc[k]) for k in arc.files if not k.startswith("__")}
        epoch = int(arc["__epoch__"]) if "__epoch__" in arc.files else None
    else:
        obj = torch.load(path, map_location="cpu", weights_only=False)
        epoch = obj.get("epoch") if isinstance(obj, dict) else None
        sd = obj["solver_state_dict"] if isinstance(obj, dict) and "solver_state_dict" in obj else obj
    return {_strip(k): v for k, v in sd.items()}, epoch


def load_solver(solver, path):
    """solver.load_state_dict for a solver checkpoint; a bare denoiser state dict (no 'nonlinear_op.'
    prefix) is loaded into solver.nonlinear_op instead."""
    sd, epoch = read_state_dict(path)
    if all(k.startswith("nonlinear_op.") for k in sd):
        solver.load_state_dict(sd)
    else:
        solver.nonlinear_op.load_state_dict(sd)
    return epoch


def shipped(name):
    """Path of a shipped archive: 'cnn', 'rsn_cnn' (reference models/*.ckpt) or 'ffdnet_gray'
    (networks/ffdnet/models/net_gray.pth - substitute for the reference's missing ffdnet.ckpt)."""
    return os.path.join(WEIGHTS_DIR, name + ".npz")
