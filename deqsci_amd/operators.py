"""SCI sensing operators with the reference's call signatures, executed by the HIP kernels.

    A_torch_(x, Phi)  -> y      utils/cg_utils.py:85-90     (K1)
    At_torch_(y, Phi) -> x      utils/cg_utils.py:124-129   (K2)
    initial_point(y, Phi, Phi_sum, gt)   utils/cg_utils.py:228-229
    phi_sum(Phi)                 training/sci_equilibrium_training.py:162-163
    LinearOperator / SCIOperator operators/operator.py:3-14 (API type; SCIOperator is the working
                                 subclass the reference's `measurement_sci` stub :34-42 never became)

Tensors are (bsz,H,W,B) / (bsz,H,W) fp32 on the GPU, exactly as the reference passes them.
"""
import torch

from . import _hip, autograd as _ag


def A_torch_(x, Phi):
    """Forward model of snapshot compressive imaging: y = sum_b x_b * Phi_b."""
    if _ag.taping(x):
        return _ag.sci_forward(x, Phi)
    return _hip.sci_forward(_hip.f32c(x), _hip.f32c(Phi), _hip.LAYOUT_HWB)


def At_torch_(y, Phi):
    """Transpose of the forward model: x_b = y * Phi_b."""
    if _ag.taping(y):
        return _ag.sci_adjoint(y, Phi)
    return _hip.sci_adjoint(_hip.f32c(y), _hip.f32c(Phi), _hip.LAYOUT_HWB)


def initial_point(y, Phi, Phi_sum=None, gt=None):
    """x0 = At(y, Phi); Phi_sum and gt are accepted and ignored, as in the reference."""
    return At_torch_(y, Phi)


def phi_sum(Phi):
    """sum over the frame axis with zeros replaced by one."""
    return _hip.phi_sum(_hip.f32c(Phi), _hip.LAYOUT_HWB)


def gap_update(z, y, Phi, Phi_sum):
    """z + At((y - A(z,Phi)) / Phi_sum, Phi) in one kernel (solvers/equilibrium_solvers_yaping.py:399-400)."""
    return _hip.gap_update(_hip.f32c(z), _hip.f32c(Phi), _hip.f32c(y), _hip.f32c(Phi_sum), _hip.LAYOUT_HWB)


class LinearOperator(torch.nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, x):
        pass

    def adjoint(self, x):
        pass

    def gramian(self, x):
        return self.adjoint(self.forward(x))


class SCIOperator(LinearOperator):
    """Phi as a LinearOperator: forward = A_torch_(., Phi), adjoint = At_torch_(., Phi)."""

    def __init__(self, Phi):
        super().__init__()
        self.register_buffer("Phi", _hip.f32c(Phi))

    def forward(self, x):
        return A_torch_(x, self.Phi)

    def adjoint(self, y):
        return At_torch_(y, self.Phi)
