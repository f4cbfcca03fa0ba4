"""Autograd wrappers of the SCI kernels, used only when a tape is being recorded (training-mode
DEQFixedPoint, solvers/new_equilibrium_utils_yaping.py:268-280 in the reference).

The operators are linear in the image / measurement, so every backward is again one of the HIP kernels:

    y = A(x, Phi)                      grad_x = At(grad_y, Phi)
    x = At(y, Phi)                     grad_y = A(grad_x, Phi)
    z1 = z + At((y - A z)/Phi_sum)     grad_z = g - At(A(g)/Phi_sum)  (= the same GAP kernel with y = 0: I - Phi^T D Phi is symmetric)
                                       grad_y = A(g) / Phi_sum

Masks (Phi, Phi_sum) are data: no gradient is produced for them (the reference never asks for one).
(bsz,H,W,B) layout, fp32, GPU - like the forward kernels; there is no CPU path.
"""
import torch

from . import _hip
from ._hip import LAYOUT_HWB


class _SCIForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Phi):
        Phi = _hip.f32c(Phi)
        ctx.save_for_backward(Phi)
        return _hip.sci_forward(_hip.f32c(x), Phi, LAYOUT_HWB)

    @staticmethod
    def backward(ctx, gy):
        (Phi,) = ctx.saved_tensors
        return _hip.sci_adjoint(_hip.f32c(gy), Phi, LAYOUT_HWB), None


class _SCIAdjoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, Phi):
        Phi = _hip.f32c(Phi)
        ctx.save_for_backward(Phi)
        return _hip.sci_adjoint(_hip.f32c(y), Phi, LAYOUT_HWB)

    @staticmethod
    def backward(ctx, gx):
        (Phi,) = ctx.saved_tensors
        return _hip.sci_forward(_hip.f32c(gx), Phi, LAYOUT_HWB), None


class _GapUpdate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, y, Phi, Phi_sum):
        Phi, Phi_sum = _hip.f32c(Phi), _hip.f32c(Phi_sum)
        ctx.save_for_backward(Phi, Phi_sum)
        return _hip.gap_update(_hip.f32c(z), Phi, _hip.f32c(y), Phi_sum, LAYOUT_HWB, LAYOUT_HWB)

    @staticmethod
    def backward(ctx, g):
        Phi, Phi_sum = ctx.saved_tensors
        g = _hip.f32c(g)
        gz = gy = None
        if ctx.needs_input_grad[0]:
            zero_y = torch.zeros(g.shape[:3], device=g.device, dtype=torch.float32)
            gz = _hip.gap_update(g, Phi, zero_y, Phi_sum, LAYOUT_HWB, LAYOUT_HWB)
        if ctx.needs_input_grad[1]:
            gy = _hip.sci_forward(g, Phi, LAYOUT_HWB) / Phi_sum
        return gz, gy, None, None


def taping(*tensors):
    """True when autograd is recording and one of the tensors takes part in it."""
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


def sci_forward(x, Phi):
    return _SCIForward.apply(x, Phi)


def sci_adjoint(y, Phi):
    return _SCIAdjoint.apply(y, Phi)


def gap_update(z, y, Phi, Phi_sum):
    return _GapUpdate.apply(z, y, Phi, Phi_sum)
