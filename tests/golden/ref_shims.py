"""Import shims that let the *reference* (IndigoPurple/DEQSCI, mounted read-only at
/root/reference) be imported on CPU in this container so golden vectors can be generated
from its own code.  TEST INFRASTRUCTURE ONLY - contains no reference source, is never
imported by the product (deqsci_amd/) and is useless on the GPU box (no /root/reference).

What is shimmed:
  * absent third-party modules the reference imports at module scope but the SCI path
    never calls: imageio, h5py, cv2, skimage.restoration, torch.utils.tensorboard
  * skimage.metrics.peak_signal_noise_ratio - restated from its definition for float
    input with data_range 1 (training/sci_equilibrium_training.py:182-183)
  * torch.solve (removed in torch>=1.13)  -> torch.linalg.solve, same (solution, LU) tuple
  * .cuda() on tensors/modules -> identity, torch.cuda.FloatTensor -> torch.FloatTensor
  * scipy.io.matlab private names moved in scipy>=1.8
"""
import sys
import types

import numpy as np
import torch

REFERENCE_ROOT = "/root/reference"

PSNR_LOG = []  # (psnr, ) appended by the shimmed peak_signal_noise_ratio


def _psnr(image_true, image_test, data_range=None):
    a = np.asarray(image_true)
    b = np.asarray(image_test)
    err = np.mean((a - b) ** 2, dtype=np.float64)
    val = 10.0 * np.log10(1.0 / err)
    PSNR_LOG.append(float(val))
    return val


def install():
    if getattr(install, "_done", False):
        return
    for name in ("imageio", "h5py", "cv2"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            sys.modules[name] = m
    sys.modules["cv2"].imwrite = lambda *a, **k: True

    sk = types.ModuleType("skimage")
    sk_rest = types.ModuleType("skimage.restoration")
    sk_rest.denoise_tv_chambolle = lambda *a, **k: (_ for _ in ()).throw(
        NotImplementedError("TV initialiser is not on the SCI hot path"))
    sk_met = types.ModuleType("skimage.metrics")
    sk_met.peak_signal_noise_ratio = _psnr
    sk.restoration = sk_rest
    sk.metrics = sk_met
    sys.modules.setdefault("skimage", sk)
    sys.modules.setdefault("skimage.restoration", sk_rest)
    sys.modules.setdefault("skimage.metrics", sk_met)

    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules.setdefault("torch.utils.tensorboard", tb)
    torch.utils.tensorboard = tb

    if not hasattr(torch, "solve") or True:
        torch.solve = lambda B, A: (torch.linalg.solve(A, B), None)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    import scipy.io.matlab as siom
    try:
        from scipy.io.matlab import _mio, _miobase
        mio = types.ModuleType("scipy.io.matlab.mio")
        mio._open_file = _mio._open_file
        miobase = types.ModuleType("scipy.io.matlab.miobase")
        miobase.get_matfile_version = _miobase.get_matfile_version
        sys.modules["scipy.io.matlab.mio"] = mio
        sys.modules["scipy.io.matlab.miobase"] = miobase
        siom.mio = mio
        siom.miobase = miobase
    except ImportError:
        pass

    import matplotlib
    matplotlib.use("Agg")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    install._done = True
