"""Drop-in solver surface of the reference, backed by the HIP kernels.

    EquilibriumProxGradSCI(A, At, nonlinear_operator, eta, minval, maxval).forward(z, y, Phi, Phi_sum)
                                   solvers/equilibrium_solvers_yaping.py:382-436
    andersonexp(f, x0, m, lam, max_iter, tol, beta) -> (z, res)
                                   solvers/new_equilibrium_utils_yaping.py:153-189
    forward_iteration(f, x0, max_iter, tol) -> (z, [res])            ibid. :213-222
    DEQFixedPoint(f, solver, **kwargs).forward(y, Phi, Phi_sum, initial_point, train_flag)
                                   ibid. :241-281 (inference; with a tape: the training forward + backward hook)

Same names, argument meaning and error behaviour, tensors in the reference's (bsz,H,W,B) layout.
`andersonexp` / `forward_iteration` accept ANY callable f (kernels K4-K7 on flat (bsz,N) views, one
residual read-back per iteration exactly like the reference's `.item()`); when DEQFixedPoint is
given this module's EquilibriumProxGradSCI together with one of these two iterators it routes the
whole loop to deqsci_amd.engine.DEQSCIEngine (planar state, fused kernels, no per-iteration sync).
"""
import torch
import torch.nn as nn

from . import _hip, autograd as _ag
from ._hip import LAYOUT_BHW, LAYOUT_HWB
from .engine import SIGMA0, SIGMA_DECAY, DEQSCIEngine
from .operators import A_torch_, At_torch_


class EquilibriumProxGradSCI(nn.Module):
    def __init__(self, A, At, nonlinear_operator, eta, minval=-1, maxval=1):
        super().__init__()
        self.A = A
        self.At = At
        self.nonlinear_op = nonlinear_operator
        self.minval = minval
        self.maxval = maxval
        self.y = 0
        self.noise_sigma = None
        self._y_ref = None          # the measurement tensor of the previous call (held, so its storage cannot be recycled)
        self._y_ver = -1

    def _sigma(self, y, n):
        """sigma bookkeeping of :408-413: restart at 60/255 when y.mean() changes, else *0.971.
        The reference compares the means on every call (one host sync each); here the comparison is skipped only when
        the SAME tensor object, unmodified, is passed again - which is how the DEQ loop calls f."""
        if y is self._y_ref and y._version == self._y_ver:
            changed = False
        else:
            ym = y.mean()
            changed = bool(torch.as_tensor(self.y != ym))
            self.y = ym
            self._y_ref, self._y_ver = y, y._version
        if changed or self.noise_sigma is None:
            self.noise_sigma = torch.full((1,), SIGMA0, dtype=torch.float32, device=y.device).expand(n)
        else:
            self.noise_sigma = self.noise_sigma * SIGMA_DECAY
        return self.noise_sigma

    def _forward_taped(self, z, y, Phi, Phi_sum):
        """The same map with autograd recording (training: :268-272 of the DEQ wrapper call f with the tape on): the GAP
        projection is the HIP kernel behind an autograd.Function, the denoiser is the PyTorch module itself and the
        layout shuffles are torch views, exactly the operations of :399-420."""
        bsz, w, h, c = z.shape
        op = self.nonlinear_op
        tag = getattr(op, "tag", None)
        if self.A is A_torch_ and self.At is At_torch_:
            z1 = _ag.gap_update(z, y, Phi, Phi_sum)
        else:
            z1 = z + self.At((y - self.A(z, Phi)) / Phi_sum, Phi)
        zp = z1.permute(0, 3, 1, 2).contiguous()
        if tag == 'conv2d':
            return op(zp.view(bsz * c, 1, w, h)).view(bsz, c, w, h).permute(0, 2, 3, 1)
        if tag == 'conv3d':
            return op(zp.view(bsz, 1, c, w, h)).view(bsz, c, w, h).permute(0, 2, 3, 1)
        if tag == 'ffdnet':
            noise = op(zp.view(bsz * c, 1, w, h), self._sigma(y, bsz * c))
        elif tag == 'denoiser':
            noise = op(zp.view(bsz * c, 1, w, h))
        elif tag == '3d_denoiser':
            noise = op(zp.view(bsz, 1, c, w, h))
        else:
            print('unknown nonlinear_op tag!')
            raise UnboundLocalError("local variable 'z_tplus1' referenced before assignment")
        return z1 - noise.view(bsz, c, w, h).permute(0, 2, 3, 1)

    def forward(self, z, y, Phi, Phi_sum):
        bsz, w, h, c = z.shape
        op = self.nonlinear_op
        tag = getattr(op, "tag", None)
        if torch.is_grad_enabled() and (z.requires_grad or any(p.requires_grad for p in op.parameters())):
            return self._forward_taped(z, y, Phi, Phi_sum)
        if self.A is A_torch_ and self.At is At_torch_:
            # K3 fused with the permute(0,3,1,2).contiguous() of :415/:419
            z1 = _hip.gap_update(_hip.f32c(z), _hip.f32c(Phi), _hip.f32c(y), _hip.f32c(Phi_sum), LAYOUT_HWB, LAYOUT_BHW)
        else:
            fb = self.A(z, Phi)
            z1 = _hip.transpose(_hip.f32c(z + self.At((y - fb) / Phi_sum, Phi)), LAYOUT_BHW)
        if tag == 'conv2d':
            out = op(z1.view(bsz * c, 1, w, h))
            return _hip.transpose(_hip.f32c(out.reshape(bsz, c, w, h)), LAYOUT_HWB)
        if tag == 'conv3d':
            out = op(z1.view(bsz, 1, c, w, h))
            return _hip.transpose(_hip.f32c(out.reshape(bsz, c, w, h)), LAYOUT_HWB)
        if tag == 'ffdnet':
            noise = op(z1.view(bsz * c, 1, w, h), self._sigma(y, bsz * c))
        elif tag == 'denoiser':
            noise = op(z1.view(bsz * c, 1, w, h))
        elif tag == '3d_denoiser':
            noise = op(z1.view(bsz, 1, c, w, h))
        else:
            print('unknown nonlinear_op tag!')
            raise UnboundLocalError("local variable 'z_tplus1' referenced before assignment")
        return _hip.residual_out(z1, _hip.f32c(noise.reshape(bsz, c, w, h)), LAYOUT_HWB)


def andersonexp(f, x0, m=5, lam=1e-4, max_iter=50, tol=1e-5, beta=1.0, *, anderson_arith="reference"):
    """Anderson acceleration for fixed point iteration (generic f).
    anderson_arith (this build's keyword): "reference" - G G^T accumulated in fp32 along N and an fp32 LU, the arithmetic of the reference's
    torch.bmm + torch.solve (solvers/new_equilibrium_utils_yaping.py:177-180); "float64" - the exactly accumulated Gram."""
    if anderson_arith not in ("reference", "float64"):
        raise ValueError(f"anderson_arith={anderson_arith!r}: expected 'reference' or 'float64'")
    ref = anderson_arith == "reference"
    bsz = x0.shape[0]
    shape = x0.shape
    xa = _hip.f32c(x0).reshape(bsz, -1)
    N = xa.shape[1]
    if m < 2:
        raise IndexError("index 1 is out of bounds for dimension 1 with size %d" % m)
    ws = _hip.AndersonWorkspace(bsz, N, m, x0.device)
    xb = torch.empty_like(xa)
    flat = lambda t: _hip.f32c(t).reshape(bsz, N)
    _hip.residual_store(ws, flat(f(xa.view(shape))), None, xa, 0, 1, xb, ref=ref)       # X1 = F0
    _hip.anderson_solve(ws, 0, 1, 0, lam, 1e-5, ref=ref)
    _hip.residual_store(ws, flat(f(xb.view(shape))), None, xb, 1, 2, None, ref=ref)
    _hip.anderson_solve(ws, 1, 2, 2, lam, 1e-5, ref=ref)
    bufs = [xa.clone(), xb]
    cur = bufs[0]                                                              # X[:, 0] = x0 if the loop is skipped
    res = None
    for k in range(2, max_iter):
        n = min(k, m)
        cur = bufs[k % 2]
        _hip.anderson_mix(ws, cur, beta, n)
        nf = min(k + 1, m)
        _hip.residual_store(ws, flat(f(cur.view(shape))), None, cur, k % m, nf, None, ref=ref)
        _hip.anderson_solve(ws, k % m, nf, nf, lam, 1e-5, ref=ref)
        res = ws.res[0, 0].item()
        if res < tol:
            break
    if res is None:
        raise UnboundLocalError("local variable 'res' referenced before assignment")
    return cur.view(shape), res


def forward_iteration(f, x0, max_iter=50, tol=1e-5):
    bsz = x0.shape[0]
    shape = x0.shape
    f0 = f(x0)
    N = f0[0].numel()
    ws = _hip.AndersonWorkspace(bsz, N, 1, x0.device)
    res = []
    for k in range(max_iter):
        x = f0
        f0 = f(x)
        _hip.residual_store(ws, _hip.f32c(f0).reshape(bsz, N), None, _hip.f32c(x).reshape(bsz, N), 0, 1, None)
        _hip.anderson_solve(ws, 0, 1, 0, 0.0, 1e-7)
        res.append(ws.res[0, 0].item())
        if res[-1] < tol:
            break
    return f0, res


class DEQFixedPoint(nn.Module):
    def __init__(self, f, solver, **kwargs):
        super().__init__()
        self.f = f
        self.solver = solver
        self.kwargs = kwargs
        self.forward_res = None
        self.use_engine = True
        self.engine_options = {}          # extra DEQSCIEngine arguments of this build, e.g. {"conv64": "fast32"} (fp32-MFMA kernels only)
        self._engine = None

    def _engine_for(self):
        f = self.f.module if isinstance(self.f, nn.DataParallel) else self.f
        if not (self.use_engine and isinstance(f, EquilibriumProxGradSCI) and f.A is A_torch_ and f.At is At_torch_):
            return None
        if getattr(f.nonlinear_op, "tag", None) not in ("conv2d", "conv3d", "ffdnet", "denoiser", "3d_denoiser"):
            return None
        kw = dict(self.kwargs)
        if self.solver is andersonexp:
            # (the reference's API gets the reference's arithmetic for alpha - an fp32 Gram - unless told otherwise: DESIGN section 5, "Config 2")
            cfg = dict(iterator="anderson", m=kw.pop("m", 5), lam=kw.pop("lam", 1e-4), max_iter=kw.pop("max_iter", 50),
                       tol=kw.pop("tol", 1e-5), beta=kw.pop("beta", 1.0), anderson_arith=kw.pop("anderson_arith", "reference"))
        elif self.solver is forward_iteration:
            cfg = dict(iterator="picard", max_iter=kw.pop("max_iter", 50), tol=kw.pop("tol", 1e-5))
        else:
            return None
        if kw:
            raise TypeError(f"{self.solver.__name__}() got an unexpected keyword argument '{next(iter(kw))}'")
        key = (id(f.nonlinear_op), f.nonlinear_op.training, tuple(sorted(cfg.items())), tuple(sorted(self.engine_options.items())))
        if self._engine is None or self._engine[0] != key:
            self._engine = (key, DEQSCIEngine(f.nonlinear_op, **{**cfg, **self.engine_options}))
        return self._engine[1]

    def forward(self, x, Phi, Phi_sum, initial_point=None, train_flag=True):
        """x is the measurement y.  Without a tape (torch.no_grad(), or no parameter of f requiring a gradient) this is the
        inference path; with one it is the reference's training forward (:249-281): solve without tape, one taped f call,
        and the implicit-differentiation hook that solves  g = J_f^T g + grad  with the same solver and settings
        (`self.backward_res`).  `train_flag=False` skips the tape even when one could be recorded (the reference has that
        switch commented out, :277-279, and always records)."""
        init_point = torch.zeros_like(x) if initial_point is None else initial_point
        if train_flag and torch.is_grad_enabled() and any(p.requires_grad for p in self.f.parameters()):
            with torch.no_grad():
                z, self.forward_res = self.solver(lambda z: self.f(z, x, Phi, Phi_sum), init_point, **self.kwargs)
            z = self.f(z, x, Phi, Phi_sum)                                     # re-engage the tape (:268)
            z0 = z.clone().detach().requires_grad_()
            f0 = self.f(z0, x, Phi, Phi_sum)                                   # Jacobian-vector products come from this graph

            def backward_hook(grad):
                g, self.backward_res = self.solver(
                    lambda v: torch.autograd.grad(f0, z0, v, retain_graph=True)[0] + grad, grad, **self.kwargs)
                return g
            z.register_hook(backward_hook)
            return z
        eng = self._engine_for()
        if eng is not None:
            z = eng.reconstruct(x, Phi, Phi_sum, initial_point=init_point)
            info = eng.last_info
            if eng.iterator == "picard":
                rows = eng._ws[next(iter(eng._ws))].host_res
                self.forward_res = [float(v) for v in rows[1:info["iterations"] + 1, 0]]
            else:
                self.forward_res = info["res"]
            return z
        with torch.no_grad():
            z, self.forward_res = self.solver(lambda z: self.f(z, x, Phi, Phi_sum), init_point, **self.kwargs)
            z = self.f(z, x, Phi, Phi_sum)
            self.f(z, x, Phi, Phi_sum)      # the reference's f0 = f(z0): advances the sigma state (:271-272)
        return z


# ----------------------------------------------------------------------------- ADMM variant
class EquilibriumADMMSCI(nn.Module):
    """solvers/equilibrium_solvers_yaping.py:438-465: one ADMM iterate (z,u) -> (z',u').  The projection is
    the same K3 kernel applied to (z+u) with Phi_sum + 1e-8; the denoiser sees z' - u and returns the CLEAN
    image (dispatch on `nonlinear_op.conv3d`, as in the reference)."""

    def __init__(self, A, At, nonlinear_operator, eta, minval=-1, maxval=1):
        super().__init__()
        self.A, self.At = A, At
        self.nonlinear_op = nonlinear_operator
        self.minval, self.maxval = minval, maxval

    def forward(self, z, u, y, Phi, Phi_sum):
        bsz, w, h, c = z.shape
        zu = _hip.f32c(z + u)
        if self.A is A_torch_ and self.At is At_torch_:
            zp = _hip.gap_update(zu, _hip.f32c(Phi), _hip.f32c(y), _hip.f32c(Phi_sum + 1e-8), LAYOUT_HWB, LAYOUT_BHW)
        else:
            fb = self.A(zu, Phi)
            zp = _hip.transpose(_hip.f32c(zu + self.At((y - fb) / (Phi_sum + 1e-8), Phi)), LAYOUT_BHW)
        vin = zp - _hip.transpose(_hip.f32c(u), LAYOUT_BHW)
        if not self.nonlinear_op.conv3d:
            den = self.nonlinear_op(vin.view(bsz * c, 1, w, h)).reshape(bsz, c, w, h)
        else:
            den = self.nonlinear_op(vin.view(bsz, 1, c, w, h)).reshape(bsz, c, w, h)
        z_new = _hip.transpose(zp, LAYOUT_HWB)
        u_new = u - _hip.residual_out(zp, _hip.f32c(den), LAYOUT_HWB)          # u - (z - z_tplus1)
        return z_new, u_new


def initial_point_admm(y, Phi, Phi_sum=None, gt=None):
    """utils/cg_utils.py:238-239."""
    return [At_torch_(y, Phi), torch.zeros_like(Phi)]


def admmexp(f, x0, m=5, lam=1e-4, max_iter=50, tol=1e-2, beta=1.0):
    """solvers/new_equilibrium_utils_yaping.py:396-413: plain ADMM fixed-point iteration; on convergence the
    PREVIOUS (X,U) is returned, as in the reference.  m, lam, beta are accepted and unused there too."""
    X, U = x0[0], x0[1]
    bsz, N = X.shape[0], X[0].numel()
    ws = _hip.AndersonWorkspace(bsz, N, 1, X.device)
    res = None
    for k in range(2, max_iter):
        new_X, new_U = f(X, U)
        _hip.residual_store(ws, _hip.f32c(new_X).reshape(bsz, N), None, _hip.f32c(X).reshape(bsz, N), 0, 1, None)
        _hip.anderson_solve(ws, 0, 1, 0, 0.0, 1e-5)
        res = ws.res[0, 0].item()
        if res < tol:
            break
        X, U = new_X, new_U
    if res is None:
        raise UnboundLocalError("local variable 'res' referenced before assignment")
    return X, U, res


class DEQFixedPointADMM(nn.Module):
    """solvers/new_equilibrium_utils_yaping.py:416-451 (inference part)."""

    def __init__(self, f, solver1, solver2, **kwargs):
        super().__init__()
        self.f, self.solver1, self.solver2, self.kwargs = f, solver1, solver2, kwargs
        self.forward_res = None

    def forward(self, x, Phi, Phi_sum, initial_point=None, train_flag=True):
        init_point = [torch.zeros_like(x), torch.zeros_like(x)] if initial_point is None else initial_point
        with torch.no_grad():
            z, u, self.forward_res = self.solver1(lambda z, u: self.f(z, u, x, Phi, Phi_sum), init_point, **self.kwargs)
        return z
