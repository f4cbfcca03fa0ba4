"""ctypes binding of the C-ABI HIP library (include/deqsci_hip.h -> lib/libdeqsci_hip.so).

PyTorch is used for device memory and streams only: every function takes torch CUDA(HIP)
tensors, passes raw device pointers + the current HIP stream to the library and returns
torch tensors.  There is NO CPU fallback: a missing library or a non-GPU tensor raises.
"""
import ctypes
import math
import os

import torch

LAYOUT_HWB = 0   # (bsz,H,W,B)  reference API layout
LAYOUT_BHW = 1   # (bsz,B,H,W)  planar / denoiser layout
MAX_M = 8
PART_STRIDE = MAX_M + 1

# DEQSCI_HIP_LIB: another build of the SAME library (tools only: the -DDEQSCI_DIAG variant of `make diag`, an ablation build)
_LIB_PATH = os.environ.get("DEQSCI_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libdeqsci_hip.so")
_lib = None

_i64, _int, _f32, _ptr = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p

# name -> argtypes; the single source of truth for tests/test_cabi_exports.py too
SIGNATURES = {
    "deqsci_sci_forward_f32": [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _int, _int, _ptr],
    "deqsci_sci_adjoint_f32": [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _int, _int, _ptr],
    "deqsci_phi_sum_f32": [_ptr, _ptr, _i64, _i64, _i64, _i64, _int, _ptr],
    "deqsci_gap_update_f32": [_ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _int, _int, _int, _ptr],
    "deqsci_transpose_f32": [_ptr, _ptr, _i64, _i64, _i64, _i64, _int, _ptr],
    "deqsci_residual_out_f32": [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _int, _ptr],
    "deqsci_residual_store_f32": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _int, _int, _int, _ptr],
    "deqsci_anderson_solve_f32": [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _int, _int, _int, _int, _f32, _f32, _ptr],
    "deqsci_anderson_solve_gram_f32": [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _int, _int, _int, _int, _f32, _f32, _ptr, _ptr],
    "deqsci_anderson_solve_ref_f32": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _int, _int, _int, _int, _f32, _f32, _ptr],
    "deqsci_residual_store_ref_f32": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _int, _int, _int, _ptr],
    "deqsci_anderson_apply_solve_ref_f32": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _int, _int, _int, _int, _f32, _f32, _ptr],
    "deqsci_gram_ref_fusable": [_i64, _i64],
    "deqsci_gram_row_chain16_f32": [_ptr, _ptr, _ptr, _i64, _i64, _int, _int, _int, _int, _ptr],
    "deqsci_anderson_mix_f32": [_ptr, _ptr, _ptr, _ptr, _f32, _int, _i64, _i64, _int, _ptr],
    "deqsci_anderson_mix_gap_f32": [_ptr, _ptr, _ptr, _f32, _int, _int, _ptr, _ptr, _ptr, _ptr, _ptr,
                                    _i64, _i64, _i64, _i64, _int, _int, _ptr],
    "deqsci_anderson_mix_gap_timed_f32": [_ptr, _ptr, _ptr, _f32, _int, _int, _ptr, _ptr, _ptr, _ptr, _ptr,
                                          _i64, _i64, _i64, _i64, _int, _int, _ptr, _ptr, _ptr],
    "deqsci_bias_relu_f32": [_ptr, _ptr, _i64, _i64, _i64, _int, _int, _ptr],
    "deqsci_ffdnet_tail_f32": [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _ptr],
    "deqsci_ffdnet_head_f32": [_ptr, _ptr, _ptr, _i64, _ptr, _i64, _i64, _i64, _ptr],
    "deqsci_conv3x3_c64_to_1_f32": [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _ptr],
    "deqsci_conv3x3_c1_to_64_f32": [_ptr, _ptr, _ptr, _i64, _i64, _i64, _int, _ptr],
    "deqsci_conv3x3_c64_winograd_f32": [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _int, _ptr],
    "deqsci_conv3x3_c64_winograd_timed_f32": [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _int, _ptr, _ptr, _ptr],
    "deqsci_conv3x3_c64_winograd44_f32": [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _int, _ptr],
    "deqsci_conv3x3_c64_winograd44_timed_f32": [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _int, _ptr, _ptr, _ptr],
    "deqsci_conv3x3_c64_winograd44_layout_f32": [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _int, _int, _int, _ptr, _ptr, _ptr],
    "deqsci_conv3x3_c64_split16": [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _int, _int, _ptr, _int, _ptr, _int, _ptr, _int, _ptr, _ptr, _ptr],
    "deqsci_conv3x3_c64_split16_stack": [_ptr, _ptr, _ptr, _ptr, _int, _i64, _i64, _i64, _ptr, _i64, _int, _int, _ptr, _ptr, _ptr, _ptr],
    "deqsci_conv3x3_c64_wino16": [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _int, _int, _ptr, _int, _ptr, _int, _ptr, _ptr, _ptr],
    "deqsci_conv3x3_c64_wino16_stack": [_ptr, _ptr, _ptr, _ptr, _int, _i64, _i64, _i64, _ptr, _i64, _int, _int, _ptr, _ptr, _ptr, _ptr],
    "deqsci_f32_to_split16": [_ptr, _ptr, _i64, _i64, _i64, _ptr, _int, _ptr],
    "deqsci_absmax_f32": [_ptr, _i64, _i64, _ptr, _ptr],
    "deqsci_ffdnet_tail_split16": [_ptr, _ptr, _ptr, _i64, _i64, _i64, _int, _ptr, _int, _ptr],
    "deqsci_ffdnet_head_split16": [_ptr, _ptr, _ptr, _i64, _ptr, _i64, _i64, _i64, _int, _ptr, _int, _ptr, _int, _ptr, _ptr],
    "deqsci_conv3x3_c64_to_1_split16": [_ptr, _ptr, _ptr, _i64, _i64, _i64, _int, _ptr, _int, _ptr],
    "deqsci_conv3x3_c64_to_1_p32": [_ptr, _ptr, _ptr, _i64, _i64, _i64, _int, _ptr, _int, _ptr],
    "deqsci_conv3x3_c1_to_64_p32": [_ptr, _ptr, _ptr, _i64, _i64, _i64, _int, _ptr, _int, _ptr, _ptr],
    "deqsci_ffdnet_head_p32": [_ptr, _ptr, _ptr, _i64, _ptr, _i64, _i64, _i64, _int, _ptr, _int, _ptr, _int, _ptr],
    "deqsci_ffdnet_tail_p32": [_ptr, _ptr, _ptr, _i64, _i64, _i64, _int, _ptr, _int, _ptr],
    "deqsci_conv3x3_c1_to_64_sp16": [_ptr, _ptr, _ptr, _i64, _i64, _i64, _int, _ptr, _int, _ptr, _ptr],
    "deqsci_event_create": [ctypes.POINTER(_ptr)],
    "deqsci_event_destroy": [_ptr],
    "deqsci_event_elapsed_ms": [_ptr, _ptr, ctypes.POINTER(_f32)],
}
OTHER_EXPORTS = ("deqsci_version", "deqsci_error_string", "deqsci_anderson_chunks",
                 "deqsci_partials_bytes", "deqsci_gram_bytes", "deqsci_gram_ref_bytes")


class DeqsciHipError(RuntimeError):
    pass


def lib_path():
    return _LIB_PATH


def load():
    """Load libdeqsci_hip.so (after torch, so it binds to the HIP runtime torch already loaded)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise DeqsciHipError(
            f"{_LIB_PATH} is missing: build it with `make` (or __graft_entry__.build()). "
            "deqsci_amd has no CPU / eager fallback for its HIP kernels.")
    lib = ctypes.CDLL(_LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _int
    lib.deqsci_version.restype = ctypes.c_char_p
    lib.deqsci_error_string.restype = ctypes.c_char_p
    lib.deqsci_error_string.argtypes = [_int]
    lib.deqsci_anderson_chunks.restype = _i64
    lib.deqsci_anderson_chunks.argtypes = [_i64, _i64]
    lib.deqsci_partials_bytes.restype = ctypes.c_size_t
    lib.deqsci_partials_bytes.argtypes = [_i64, _i64]
    lib.deqsci_gram_bytes.restype = ctypes.c_size_t
    lib.deqsci_gram_bytes.argtypes = [_i64]
    lib.deqsci_gram_ref_bytes.restype = ctypes.c_size_t
    lib.deqsci_gram_ref_bytes.argtypes = [_i64, _i64]
    _lib = lib
    return lib


def _check(code, what):
    if code != 0:
        msg = load().deqsci_error_string(code).decode()
        raise DeqsciHipError(f"{what} failed with code {code}: {msg}")


def _p(t, name="tensor", allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise DeqsciHipError(f"{name} is None")
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise DeqsciHipError(f"{name} must be a torch tensor on a HIP device (got "
                             f"{getattr(t, 'device', type(t))}); deqsci_amd has no CPU path")
    if t.dtype != torch.float32:
        raise DeqsciHipError(f"{name} must be float32, got {t.dtype}")
    if not t.is_contiguous():
        raise DeqsciHipError(f"{name} must be contiguous")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NULL = _NullCtx()


def _dev(t):
    """Device guard only when the tensor is not on the current device (the common case is free)."""
    if not t.is_cuda or t.device.index == torch.cuda.current_device():
        return _NULL
    return torch.cuda.device(t.device)


def f32c(t):
    """fp32 contiguous view/copy (the reference's tensors are fp32 throughout)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _dims(layout, shape):
    if layout == LAYOUT_HWB:
        bsz, H, W, B = shape
    else:
        bsz, B, H, W = shape
    return bsz, H, W, B


def _shape(layout, bsz, H, W, B):
    return (bsz, H, W, B) if layout == LAYOUT_HWB else (bsz, B, H, W)


def _phi_shared(phi, bsz):
    if phi.dim() == 3:
        return 1
    if phi.shape[0] == 1 and bsz > 1:
        return 1
    if phi.shape[0] != bsz:
        raise DeqsciHipError(f"Phi batch {phi.shape[0]} does not match batch {bsz}")
    return 0


# ----------------------------------------------------------------------------- operators
def sci_forward(x, phi, layout=LAYOUT_HWB, out=None):
    bsz, H, W, B = _dims(layout, x.shape)
    if tuple(phi.shape[-3:]) != tuple(x.shape[-3:]):
        raise DeqsciHipError(f"x {tuple(x.shape)} and Phi {tuple(phi.shape)} do not match")
    y = out if out is not None else torch.empty((bsz, H, W), device=x.device, dtype=torch.float32)
    with _dev(x):
        _check(load().deqsci_sci_forward_f32(_p(x, "x"), _p(phi, "Phi"), _p(y, "y"), bsz, H, W, B, layout,
                                             _phi_shared(phi, bsz), _stream()), "sci_forward")
    return y


def sci_adjoint(y, phi, layout=LAYOUT_HWB, out=None):
    bsz = y.shape[0]
    _, H, W, B = _dims(layout, (1,) + tuple(phi.shape[-3:]))
    if tuple(y.shape) != (bsz, H, W):
        raise DeqsciHipError(f"y {tuple(y.shape)} and Phi {tuple(phi.shape)} do not match")
    x = out if out is not None else torch.empty(_shape(layout, bsz, H, W, B), device=y.device, dtype=torch.float32)
    with _dev(y):
        _check(load().deqsci_sci_adjoint_f32(_p(y, "y"), _p(phi, "Phi"), _p(x, "x"), bsz, H, W, B, layout,
                                             _phi_shared(phi, bsz), _stream()), "sci_adjoint")
    return x


def phi_sum(phi, layout=LAYOUT_HWB):
    p4 = phi if phi.dim() == 4 else phi.unsqueeze(0)
    nb, H, W, B = _dims(layout, p4.shape)
    out = torch.empty((nb, H, W), device=phi.device, dtype=torch.float32)
    with _dev(phi):
        _check(load().deqsci_phi_sum_f32(_p(p4, "Phi"), _p(out), nb, H, W, B, layout, _stream()), "phi_sum")
    return out if phi.dim() == 4 else out[0]


def gap_update(z, phi, y, phisum, layout_in=LAYOUT_HWB, layout_out=None, out=None):
    layout_out = layout_in if layout_out is None else layout_out
    bsz, H, W, B = _dims(layout_in, z.shape)
    if tuple(phi.shape[-3:]) != tuple(z.shape[-3:]) or tuple(y.shape) != (bsz, H, W):
        raise DeqsciHipError(f"gap_update shapes z {tuple(z.shape)} Phi {tuple(phi.shape)} y {tuple(y.shape)}")
    shared = _phi_shared(phi, bsz)
    if phisum.numel() != (1 if shared else bsz) * H * W:
        raise DeqsciHipError(f"Phi_sum {tuple(phisum.shape)} does not match Phi {tuple(phi.shape)} (shared={shared})")
    z1 = out if out is not None else torch.empty(_shape(layout_out, bsz, H, W, B), device=z.device, dtype=torch.float32)
    with _dev(z):
        _check(load().deqsci_gap_update_f32(_p(z, "z"), _p(phi, "Phi"), _p(y, "y"), _p(phisum, "Phi_sum"), _p(z1, "z1"),
                                            bsz, H, W, B, layout_in, layout_out, _phi_shared(phi, bsz), _stream()),
               "gap_update")
    return z1


def transpose(t, to_layout, out=None):
    """(bsz,H,W,B) -> (bsz,B,H,W) when to_layout == LAYOUT_BHW, and back."""
    from_layout = LAYOUT_HWB if to_layout == LAYOUT_BHW else LAYOUT_BHW
    bsz, H, W, B = _dims(from_layout, t.shape)
    o = out if out is not None else torch.empty(_shape(to_layout, bsz, H, W, B), device=t.device, dtype=torch.float32)
    with _dev(t):
        _check(load().deqsci_transpose_f32(_p(t, "in"), _p(o, "out"), bsz, H, W, B, to_layout, _stream()), "transpose")
    return o


def residual_out(z1, noise, layout_out=LAYOUT_HWB, out=None):
    """out = z1 - noise; z1, noise planar (bsz,B,H,W); out in layout_out."""
    bsz, B, H, W = z1.shape
    o = out if out is not None else torch.empty(_shape(layout_out, bsz, H, W, B), device=z1.device, dtype=torch.float32)
    with _dev(z1):
        code = load().deqsci_residual_out_f32(_p(z1, "z1"), _p(noise, "noise"), _p(o, "out"), bsz, H, W, B, layout_out, _stream())
    if code == -4 and layout_out == LAYOUT_HWB:       # odd B / H*W: planar subtract, then the generic LDS transpose
        tmp = residual_out(z1, noise, LAYOUT_BHW)
        return transpose(tmp, LAYOUT_HWB, out=o)
    _check(code, "residual_out")
    return o


# ----------------------------------------------------------------------------- Anderson workspace + steps
class AndersonWorkspace:
    """Caller-owned buffers for K4-K7 (the library never allocates)."""

    def __init__(self, bsz, N, m, device, res_rows=1):
        if m > MAX_M:
            raise DeqsciHipError(f"Anderson history m={m} exceeds DEQSCI_MAX_M={MAX_M}")
        lib = load()
        self.bsz, self.N, self.m = bsz, N, m
        self.F = torch.zeros((bsz, m, N), device=device, dtype=torch.float32)
        self.G = torch.zeros((bsz, m, N), device=device, dtype=torch.float32)
        self.nchunks = lib.deqsci_anderson_chunks(bsz, N)
        self.partials = torch.empty(lib.deqsci_partials_bytes(bsz, N) // 4, device=device, dtype=torch.float32)
        self.gram = torch.zeros(lib.deqsci_gram_bytes(bsz) // 8, device=device, dtype=torch.float64)
        self.alpha = torch.zeros((bsz, MAX_M), device=device, dtype=torch.float32)
        self.res = torch.zeros((res_rows, 1 + bsz), device=device, dtype=torch.float32)
        self.gram32 = None                # (anderson_arith = "reference": the persistent fp32 Gram of deqsci_anderson_solve_ref_f32, see gram32_state)
        self.ref_fusable = bool(lib.deqsci_gram_ref_fusable(bsz, N))      # K4 can leave the records of the reference Gram's first pass itself
        self._rounded = None              # (slot, n_filled) of a residual_store(ref=True) whose records await anderson_solve(ref=True)

    def ref_state(self):
        """The caller-owned state of deqsci_anderson_solve_ref_f32 (allocated and zeroed on first use): per sample the persistent fp32 Gram
        (MAX_M x MAX_M), the 16 chain sums per entry of the last call, and the block records of the two-pass form."""
        if self.gram32 is None:
            self.gram32 = torch.zeros(load().deqsci_gram_ref_bytes(self.bsz, self.N) // 4, device=self.F.device, dtype=torch.float32)
        return self.gram32

    def gram32_state(self):
        """(bsz, MAX_M, MAX_M) view of the persistent fp32 Gram in ref_state()."""
        st = self.ref_state().view(self.bsz, -1)
        return st[:, :MAX_M * MAX_M].view(self.bsz, MAX_M, MAX_M)

    def chain_sums(self):
        """(bsz, MAX_M, 16) view of the chain sums of the last deqsci_gram_row_chain16_f32 / deqsci_anderson_solve_ref_f32 call."""
        st = self.ref_state().view(self.bsz, -1)
        return st[:, MAX_M * MAX_M:MAX_M * MAX_M + 16 * MAX_M].view(self.bsz, MAX_M, 16)

    def chains_walked(self):
        """(bsz, MAX_M, 16) int32 view (diagnostic): the blocks each chain was walked through term by term in the last two-pass call (-1: serial form)."""
        st = self.ref_state().view(self.bsz, -1)
        return st[:, MAX_M * MAX_M + 16 * MAX_M:MAX_M * MAX_M + 32 * MAX_M].view(torch.int32).view(self.bsz, MAX_M, 16)


def residual_store(ws, z1, noise, x_cur, slot, n_filled, x_next=None, ref=False):
    """ref: the caller will form alpha with anderson_solve(ref=True) next - where the shape allows (ws.ref_fusable) K4's blocks then also leave
    the records of the reference Gram's first pass (deqsci_residual_store_ref_f32: the history rows are in their registers anyway), and
    that anderson_solve skips the pass.  Same F / G / x_next / partials either way."""
    ws._rounded = None
    if ref and ws.ref_fusable:
        with _dev(z1):
            _check(load().deqsci_residual_store_ref_f32(_p(z1, "z1"), _p(noise, "noise", True), _p(x_cur, "x_cur"), _p(ws.F), _p(ws.G),
                                                        _p(x_next, "x_next", True), _p(ws.partials), _p(ws.ref_state()), ws.bsz, ws.N, ws.m, slot,
                                                        n_filled, _stream()), "residual_store_ref")
        ws._rounded = (slot, n_filled)
        return
    with _dev(z1):
        _check(load().deqsci_residual_store_f32(_p(z1, "z1"), _p(noise, "noise", True), _p(x_cur, "x_cur"), _p(ws.F), _p(ws.G),
                                                _p(x_next, "x_next", True), _p(ws.partials), ws.bsz, ws.N, ws.m, slot,
                                                n_filled, _stream()), "residual_store")


def anderson_solve(ws, slot, n_filled, n, lam, eps, res_row=0, gram32=None, ref=False):
    """gram32: (bsz, n, n) fp32 - alpha from THAT Gram block with an fp32 LU (the reference's arithmetic, :177-180) instead of the float64 sums.
    ref: the same arithmetic without a GEMM library - the new Gram row summed by the build's own kernel in the order of the reference's torch.bmm
    (sixteen interleaved FMA chains per entry, csrc/anderson.hip gram_row_chain16_kernel) from the history residual_store just wrote."""
    if ref and ws._rounded == (slot, n_filled):
        ws._rounded = None
        with _dev(ws.F):
            _check(load().deqsci_anderson_apply_solve_ref_f32(_p(ws.G), _p(ws.partials), _p(ws.ref_state()), ws.gram.data_ptr(), _p(ws.alpha),
                                                              _p(ws.res[res_row]), ws.bsz, ws.N, ws.m, slot, n_filled, n, float(lam), float(eps), _stream()),
                   "anderson_apply_solve_ref")
        return
    ws._rounded = None
    if ref:
        with _dev(ws.F):
            _check(load().deqsci_anderson_solve_ref_f32(_p(ws.G), _p(ws.partials), _p(ws.ref_state()), ws.gram.data_ptr(), _p(ws.alpha),
                                                        _p(ws.res[res_row]), ws.bsz, ws.N, ws.m, slot, n_filled, n, float(lam), float(eps), _stream()),
                   "anderson_solve_ref")
        return
    if gram32 is not None and (tuple(gram32.shape) != (ws.bsz, n, n) or gram32.dtype != torch.float32 or not gram32.is_contiguous() or not gram32.is_cuda):
        raise DeqsciHipError(f"anderson_solve: gram32 must be a contiguous fp32 GPU tensor of shape {(ws.bsz, n, n)}")
    with _dev(ws.F):
        _check(load().deqsci_anderson_solve_gram_f32(_p(ws.partials), ws.gram.data_ptr(), _p(ws.alpha), _p(ws.res[res_row]),
                                                     ws.bsz, ws.N, ws.m, slot, n_filled, n, float(lam), float(eps),
                                                     None if gram32 is None else gram32.data_ptr(), _stream()), "anderson_solve")


def gram_row_chain16(ws, slot, n_filled, serial=False):
    """The 16 chain sums of every entry of the new Gram row (ws.chain_sums()) from the history and block sums residual_store just wrote:
    serial=True as the chains are written, False in the two-pass form (the same bits; tests hold them equal)."""
    with _dev(ws.F):
        _check(load().deqsci_gram_row_chain16_f32(_p(ws.G), _p(ws.partials), _p(ws.ref_state()), ws.bsz, ws.N, ws.m, slot, n_filled, int(bool(serial)),
                                                  _stream()), "gram_row_chain16")
    return ws.chain_sums()


def anderson_mix(ws, x_out, beta, n):
    with _dev(ws.F):
        _check(load().deqsci_anderson_mix_f32(_p(ws.F), _p(ws.G), _p(ws.alpha), _p(x_out, "x_out"), float(beta), n,
                                              ws.bsz, ws.N, ws.m, _stream()), "anderson_mix")


def anderson_mix_gap(ws, beta, n, phi, y, phisum, x_out, z1, layout):
    bsz, H, W, B = _dims(layout, z1.shape)
    with _dev(ws.F):
        _check(load().deqsci_anderson_mix_gap_f32(_p(ws.F), _p(ws.G), _p(ws.alpha), float(beta), n, ws.m, _p(phi, "Phi"),
                                                  _p(y, "y"), _p(phisum, "Phi_sum"), _p(x_out, "x_out"), _p(z1, "z1"),
                                                  bsz, H, W, B, layout, _phi_shared(phi, bsz), _stream()), "anderson_mix_gap")


def bias_relu_(h, bias, relu=True):
    """In-place h = max(h + bias[c], 0) for a 4-D activation, NCHW-contiguous or channels_last."""
    n, c, hh, ww = h.shape
    if h.is_contiguous():
        cl = 0
    elif h.is_contiguous(memory_format=torch.channels_last):
        cl = 1
    else:
        raise DeqsciHipError("bias_relu_: activation must be contiguous (NCHW or channels_last)")
    if h.dtype != torch.float32 or not h.is_cuda:
        raise DeqsciHipError("bias_relu_: fp32 GPU tensor required")
    with _dev(h):
        _check(load().deqsci_bias_relu_f32(h.data_ptr(), _p(bias, "bias"), n, c, hh * ww, cl, 1 if relu else 0, _stream()),
               "bias_relu")
    return h


def pack_tail_weights(w):
    """(4,64,3,3) conv weight -> [half(2)][tap(9)][cin(32)][cout(4)] for deqsci_ffdnet_tail_f32."""
    if tuple(w.shape) != (4, 64, 3, 3):
        raise DeqsciHipError(f"ffdnet tail expects a (4,64,3,3) weight, got {tuple(w.shape)}")
    return w.detach().float().permute(2, 3, 1, 0).reshape(9, 2, 32, 4).permute(1, 0, 2, 3).contiguous()


def ffdnet_tail(h, w_packed, out=None, in_bias=None):
    """h (n,64,H,W) channels_last -> planar noise (n,1,2H,2W) = pixel_shuffle(conv3x3(h', w, pad=1), 2) with
    h' = h, or relu(h + in_bias[c]) when in_bias is given (previous layer's epilogue fused into the read).  (An Sp16 activation goes to
    tail_split16.)"""
    n, c, H, W = h.shape
    if c != 64 or not h.is_contiguous(memory_format=torch.channels_last) or h.dtype != torch.float32 or not h.is_cuda:
        raise DeqsciHipError("ffdnet_tail: fp32 channels_last GPU activation with 64 channels required")
    o = out if out is not None else torch.empty((n, 1, 2 * H, 2 * W), device=h.device, dtype=torch.float32)
    with _dev(h):
        _check(load().deqsci_ffdnet_tail_f32(h.data_ptr(), _p(w_packed, "w_packed"), _p(in_bias, "in_bias", True), _p(o, "out"),
                                             n, H, W, _stream()), "ffdnet_tail")
    return o


def pack_c64_to_1_weights(w):
    """(1,64,3,3) conv weight -> [half(2)][tap(9)][cin(32)] for deqsci_conv3x3_c64_to_1_f32."""
    if tuple(w.shape) != (1, 64, 3, 3):
        raise DeqsciHipError(f"expected a (1,64,3,3) weight, got {tuple(w.shape)}")
    return w.detach().float().permute(2, 3, 1, 0).reshape(9, 2, 32).permute(1, 0, 2).contiguous()


def conv3x3_c64_to_1(h, w_packed, out=None, in_bias=None):
    """h (n,64,H,W) channels_last -> planar (n,1,H,W) = conv3x3(h', w, pad=1), h' = h or relu(h + in_bias[c]).  (An Sp16 activation
    goes to tail_split16.)"""
    n, c, H, W = h.shape
    if c != 64 or not h.is_contiguous(memory_format=torch.channels_last) or h.dtype != torch.float32 or not h.is_cuda:
        raise DeqsciHipError("conv3x3_c64_to_1: fp32 channels_last GPU activation with 64 channels required")
    o = out if out is not None else torch.empty((n, 1, H, W), device=h.device, dtype=torch.float32)
    with _dev(h):
        _check(load().deqsci_conv3x3_c64_to_1_f32(h.data_ptr(), _p(w_packed, "w_packed"), _p(in_bias, "in_bias", True), _p(o, "out"),
                                                  n, H, W, _stream()), "conv3x3_c64_to_1")
    return o


def pack_c1_to_64_weights(w):
    """(64,1,3,3) conv weight -> [tap(9)][cout//4(16)][cout%4(4)] for deqsci_conv3x3_c1_to_64_f32."""
    if tuple(w.shape) != (64, 1, 3, 3):
        raise DeqsciHipError(f"expected a (64,1,3,3) weight, got {tuple(w.shape)}")
    return w.detach().float().reshape(16, 4, 9).permute(2, 0, 1).contiguous()


def conv3x3_c1_to_64(x, w_packed, relu=True, out=None, sp16=False, out_rng=None, out_exp=None, track=None, p32=False):
    """x (n,1,H,W) planar -> [relu](conv3x3(x, w, pad=1)) as a channels_last (n,64,H,W) activation, or as an Sp16 (sp16=True) / a P32
    (p32=True: in front of conv3x3_c64_wino16 layers) with the range (out_rng, out_exp); track: a range slot that receives max |output|
    (the measurement of the first f-call)."""
    n, c, H, W = x.shape
    if c != 1:
        raise DeqsciHipError(f"conv3x3_c1_to_64: (n,1,H,W) image required, got {tuple(x.shape)}")
    if p32:
        o = out if out is not None else P32.empty(n, H, W, x.device)
        if not isinstance(o, P32) or (o.n, o.H, o.W) != (n, H, W):
            raise DeqsciHipError("conv3x3_c1_to_64: out must be a P32 of the output's shape")
        o.rng, o.exp = out_rng, SP16_DEFAULT_EXP if out_exp is None else int(out_exp)
        with _dev(x):
            _check(load().deqsci_conv3x3_c1_to_64_p32(_p(x, "x"), _p(w_packed, "w_packed"), o.t.data_ptr(), n, H, W, 1 if relu else 0,
                                                      _rng(o.rng, n), o.exp, _rng(track, n), _stream()), "conv3x3_c1_to_64_p32")
        return o
    if sp16:
        o = out if out is not None else Sp16.empty(n, H, W, x.device)
        o.rng, o.exp = out_rng, SP16_DEFAULT_EXP if out_exp is None else int(out_exp)
        with _dev(x):
            _check(load().deqsci_conv3x3_c1_to_64_sp16(_p(x, "x"), _p(w_packed, "w_packed"), o.t.data_ptr(), n, H, W, 1 if relu else 0,
                                                       _rng(o.rng, n), o.exp, _rng(track, n), _stream()), "conv3x3_c1_to_64_sp16")
        return o
    o = out if out is not None else torch.empty((n, 64, H, W), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
    with _dev(x):
        _check(load().deqsci_conv3x3_c1_to_64_f32(_p(x, "x"), _p(w_packed, "w_packed"), o.data_ptr(), n, H, W, 1 if relu else 0,
                                                  _stream()), "conv3x3_c1_to_64")
    return o


def pack_head_weights(w):
    """(64,5,3,3) conv weight -> [ch*9+tap (45)][cout//4 (16)][cout%4 (4)] for deqsci_ffdnet_head_f32."""
    if tuple(w.shape) != (64, 5, 3, 3):
        raise DeqsciHipError(f"ffdnet head expects a (64,5,3,3) weight, got {tuple(w.shape)}")
    return w.detach().float().reshape(16, 4, 45).permute(2, 0, 1).contiguous()


def ffdnet_head(x, w_packed, sigma, out=None):
    """x (n,1,2H,2W) planar, sigma (n,) or (1,) -> relu(conv3x3(cat(sigma map, pixel_unshuffle(x,2)), w)) as a
    channels_last (n,64,H,W) activation.  (In front of a run of split-fp16 layers: ffdnet_head_split16.)"""
    n, c, H2, W2 = x.shape
    if c != 1 or H2 % 2 or W2 % 2:
        raise DeqsciHipError(f"ffdnet_head: (n,1,even,even) image required, got {tuple(x.shape)}")
    if sigma.numel() not in (1, n) or sigma.dtype != torch.float32 or not sigma.is_cuda:
        raise DeqsciHipError("ffdnet_head: sigma must be a fp32 GPU tensor with 1 or n elements")
    H, W = H2 // 2, W2 // 2
    o = out if out is not None else torch.empty((n, 64, H, W), device=x.device, dtype=torch.float32,
                                                  memory_format=torch.channels_last)
    with _dev(x):
        _check(load().deqsci_ffdnet_head_f32(_p(x, "x"), _p(w_packed, "w_packed"), sigma.data_ptr(),
                                             0 if sigma.numel() == 1 else sigma.stride(0), o.data_ptr(), n, H, W, _stream()),
               "ffdnet_head")
    return o


def _check_packed(u_packed, positions, packer):
    """The kernels DMA the whole transformed-weight array by size: a wrong pack (the other kernel's, a slice) would be read out of bounds."""
    if not isinstance(u_packed, torch.Tensor) or u_packed.numel() != 64 * 64 * positions:
        raise DeqsciHipError(f"u_packed must hold 64*64*{positions} floats (the output of _hip.{packer}), got "
                             f"{tuple(getattr(u_packed, 'shape', ()))}")


def pack_winograd_weights(w):
    """(64,64,3,3) conv weight -> U = G g G^T of Winograd F(2x2,3x3) in the kernel's MFMA-lane order
    [cin chunk c (8)][xi (16)][cout half wn (2)][q (4)][i (16)][j (2)][s (2)] with cout = 32 wn + 16 j + i and
    cin = 8 c + 2 q + s, so that LDS staging is a linear copy and a lane's B operands are 16 contiguous bytes."""
    if tuple(w.shape) != (64, 64, 3, 3):
        raise DeqsciHipError(f"winograd conv expects a (64,64,3,3) weight, got {tuple(w.shape)}")
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64, device=w.device)
    U = G @ w.detach().double() @ G.t()                      # (cout, cin, 4, 4)
    U = U.permute(2, 3, 0, 1).reshape(16, 2, 2, 16, 8, 4, 2)  # [xi][wn][j][i][c][q][s]
    return U.permute(4, 0, 1, 5, 3, 2, 6).contiguous().float()   # [c][xi][wn][q][i][j][s]


def conv3x3_c64_winograd(x, u_packed, bias=None, relu=True, out=None, events=None):
    """x (n,64,H,W) channels_last -> relu(conv3x3(x, w, pad=1) + bias) as a new channels_last tensor (Winograd F(2x2,3x3), fp32 MFMA)."""
    n, c, H, W = x.shape
    if c != 64 or not x.is_contiguous(memory_format=torch.channels_last) or x.dtype != torch.float32 or not x.is_cuda:
        raise DeqsciHipError("conv3x3_c64_winograd: fp32 channels_last GPU activation with 64 channels required")
    _check_packed(u_packed, 16, "pack_winograd_weights")
    o = out if out is not None else torch.empty_like(x, memory_format=torch.channels_last)
    ev = _hook_events("f22", n, H, W, events)
    with _dev(x):
        if ev is None:
            _check(load().deqsci_conv3x3_c64_winograd_f32(x.data_ptr(), _p(u_packed, "u_packed"), _p(bias, "bias", True), o.data_ptr(),
                                                          n, H, W, 1 if relu else 0, _stream()), "conv3x3_c64_winograd")
        else:
            _check(load().deqsci_conv3x3_c64_winograd_timed_f32(x.data_ptr(), _p(u_packed, "u_packed"), _p(bias, "bias", True), o.data_ptr(),
                                                                n, H, W, 1 if relu else 0, _stream(), ev[0], ev[1]), "conv3x3_c64_winograd_timed")
    return o


def pack_winograd44_weights(w):
    """(64,64,3,3) conv weight -> U = G g G^T of Winograd F(4x4,3x3) (6x6 per cout, cin; computed in float64, rounded once) in the
    LDS order of csrc/winograd44.hip: [cin chunk c (8)][s (18)][rg (2)][cgp (2)][q (4)][i (16)][j (2)][ks (2)] with transform
    position (3 rg + s // 6, s % 6), cout = 32 cgp + 16 j + i, cin = 8 c + 2 q + ks."""
    if tuple(w.shape) != (64, 64, 3, 3):
        raise DeqsciHipError(f"winograd conv expects a (64,64,3,3) weight, got {tuple(w.shape)}")
    G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                      [0, 0, 1]], dtype=torch.float64, device=w.device)
    U = G @ w.detach().double() @ G.t()                             # (cout, cin, 6, 6)
    U = U.reshape(2, 2, 16, 8, 4, 2, 2, 3, 6)                       # [cgp][j][i][c][q][ks][rg][sr][sc]
    return U.permute(3, 7, 8, 6, 0, 4, 2, 1, 5).contiguous().float()   # [c][sr][sc][rg][cgp][q][i][j][ks]


ACT_NHWC, ACT_BLK32 = 0, 1

# Measurement hook (bench.py): when set, every launch of a 64->64 kernel asks it for a (start, stop) pair of raw hipEvent_t handles -
# hook(kind, n, H, W) -> (ev0, ev1) or None - and the dispatch's own begin / end timestamps are written to them.
CONV64_EVENT_HOOK = None


def _hook_events(kind, n, H, W, events, layers=1):
    """kind "s16stack": one launch that runs `layers` 64->64 layers over n images; the hook is told through its `layers` keyword (a hook
    that does not take it is not asked about those launches)."""
    if events is not None:
        return events
    if CONV64_EVENT_HOOK is not None:
        if layers == 1:
            return CONV64_EVENT_HOOK(kind, n, H, W)
        try:
            return CONV64_EVENT_HOOK(kind, n, H, W, layers=layers)
        except TypeError:
            return None
    return None



class Blk32:
    """An activation (n,64,H,W) in the "blk32" layout of csrc/winograd44.hip: planes of 8 channels, blocks of 32 columns in the
    kernel's staging order - t is (n, 8, H, ceil(W/32), 32, 8) float32.  Only ever exists between two 64->64 layers."""
    __slots__ = ("t", "n", "H", "W")

    def __init__(self, t, n, H, W):
        self.t, self.n, self.H, self.W = t, n, H, W

    is_cuda = property(lambda self: self.t.is_cuda)
    device = property(lambda self: self.t.device)
    shape = property(lambda self: (self.n, 64, self.H, self.W))

    @staticmethod
    def empty(n, H, W, device):
        return Blk32(torch.empty((n, 8, H, -(-W // 32), 32, 8), dtype=torch.float32, device=device), n, H, W)

    @staticmethod
    def _pos():
        m1 = torch.arange(32) + 1
        return 8 * (m1 & 3) + (m1 >> 2) - ((m1 & 3) == 0).long()        # position of column m inside its block

    @staticmethod
    def from_nchw(x):
        """(tests / tools) any (n,64,H,W) tensor -> blk32; padding columns are filled with NaN: the kernel must never read them."""
        n, c, H, W = x.shape
        Wb = -(-W // 32)
        xp = torch.full((n, 64, H, Wb * 32), float("nan"), dtype=torch.float32, device=x.device)
        xp[..., :W] = x
        t = xp.reshape(n, 8, 8, H, Wb, 32).permute(0, 1, 3, 4, 5, 2)      # (n, chunk, H, Wb, m, ch)
        out = torch.empty_like(t.contiguous())
        out[:, :, :, :, Blk32._pos().to(x.device), :] = t
        return Blk32(out.contiguous(), n, H, W)

    def to_nchw(self):
        t = self.t[:, :, :, :, Blk32._pos().to(self.t.device), :]          # back to column order
        x = t.permute(0, 1, 5, 2, 3, 4).reshape(self.n, 64, self.H, -1)
        return x[..., :self.W].contiguous(memory_format=torch.channels_last)


def conv3x3_c64_winograd44(x, u_packed, bias=None, relu=True, out=None, out_blk=False, events=None):
    """x (n,64,H,W) channels_last, or a Blk32 -> relu(conv3x3(x, w, pad=1) + bias) as a new channels_last tensor, or a Blk32 with
    out_blk=True: Winograd F(4x4,3x3), the large-batch kernel.  events: (start, stop) raw hipEvent_t handles (measurement)."""
    if isinstance(x, Blk32):
        n, H, W, xt, in_l = x.n, x.H, x.W, x.t, ACT_BLK32
        if tuple(xt.shape) != (n, 8, H, -(-W // 32), 32, 8) or not xt.is_contiguous():
            raise DeqsciHipError(f"conv3x3_c64_winograd44: malformed Blk32 {tuple(xt.shape)} for (n, H, W) = {(n, H, W)}")
    else:
        n, c, H, W = x.shape
        if c != 64 or not x.is_contiguous(memory_format=torch.channels_last):
            raise DeqsciHipError("conv3x3_c64_winograd44: fp32 channels_last GPU activation with 64 channels required")
        xt, in_l = x, ACT_NHWC
    if xt.dtype != torch.float32 or not xt.is_cuda:
        raise DeqsciHipError("conv3x3_c64_winograd44: fp32 channels_last GPU activation with 64 channels required")
    _check_packed(u_packed, 36, "pack_winograd44_weights")
    if bias is not None and bias.numel() < 64:
        raise DeqsciHipError(f"conv3x3_c64_winograd44: bias must have 64 elements, got {bias.numel()}")
    if out_blk:
        o = out if out is not None else Blk32.empty(n, H, W, xt.device)
        if not isinstance(o, Blk32) or (o.n, o.H, o.W) != (n, H, W) or o.t.device != xt.device or not o.t.is_contiguous():
            raise DeqsciHipError("conv3x3_c64_winograd44: out must be a Blk32 of the input's (n, H, W) on its device")
        ot = o.t
    else:
        o = out if out is not None else torch.empty((n, 64, H, W), dtype=torch.float32, device=xt.device, memory_format=torch.channels_last)
        if (isinstance(o, Blk32) or tuple(o.shape) != (n, 64, H, W) or o.dtype != torch.float32 or o.device != xt.device
                or not o.is_contiguous(memory_format=torch.channels_last)):
            raise DeqsciHipError("conv3x3_c64_winograd44: out must be a fp32 channels_last (n,64,H,W) tensor on the input's device")
        ot = o
    ev = _hook_events("f44", n, H, W, events) or (None, None)
    with _dev(xt):
        _check(load().deqsci_conv3x3_c64_winograd44_layout_f32(xt.data_ptr(), _p(u_packed, "u_packed"), _p(bias, "bias", True), ot.data_ptr(),
                                                               n, H, W, 1 if relu else 0, in_l, ACT_BLK32 if out_blk else ACT_NHWC, _stream(),
                                                               ev[0], ev[1]), "conv3x3_c64_winograd44")
    return o


# ----------------------------------------------------------------------------- split-fp16 direct convolution (csrc/conv_s16.hip)
# An sp16 activation holds the fp16 pieces hi + lo of 2^e x.  fp32 is scale-free, fp16 is not, so e follows the data, PER IMAGE of the
# batch (a measurement's result never depends on what else is in the batch): the RANGE of an activation is (rng, exp) - `rng` a fp32 GPU
# tensor of n elements, rng[i] = max |x| of image i of that activation as a kernel measured it (every kernel that writes or reads the
# activation derives e(i) = act_exp(rng[i]) from those device words: nothing crosses to the host, a captured hipGraph follows its
# inputs), or None: the fixed exponent `exp` for every image.  The mirror of csrc/common.hpp: sp16_act_exp.
SP16_DEFAULT_EXP, SP16_TARGET_EXP, SP16_EXP_LIMIT = 8, 11, 64
SP16_ACT_SCALE = 2.0 ** SP16_DEFAULT_EXP     # the fixed default: 2^8 x suits activations of a few units (|x| < 255.9)


def act_exp(amax):
    """The exponent e the kernels derive from max |x| = amax: 2^e amax in [2^11, 2^12) (a factor 16 below fp16's overflow; an element
    keeps all 22 bits of its split down to 2^-14 of the maximum).  amax zero / subnormal / not finite: the default 8."""
    import math
    import struct
    b = struct.unpack("<I", struct.pack("<f", float(amax)))[0]
    e = (b >> 23) & 0xff
    if e == 0 or e == 255 or math.isnan(float(amax)):
        return SP16_DEFAULT_EXP
    return max(-SP16_EXP_LIMIT, min(SP16_EXP_LIMIT, SP16_TARGET_EXP - (e - 127)))


def _rng(t, n):
    """Device pointer of the range slots of an n-image activation (a contiguous fp32 GPU tensor of n elements), or None."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float32 or t.numel() != n or not t.is_contiguous():
        raise DeqsciHipError(f"range slots must be a contiguous fp32 GPU tensor with one element per image ({n}), got "
                             f"{tuple(getattr(t, 'shape', ()))}")
    return t.data_ptr()


def absmax(x, slots):
    """slots[i] = max(slots[i], max |x[i]|) on the device (zero them first), x (n, ...): the ranges of an activation no sp16-writing
    kernel produced."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32):
        raise DeqsciHipError("absmax: fp32 GPU tensor required")
    if not (x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last))):    # any dense order per image does
        x = x.contiguous()
    n = x.shape[0]
    with _dev(x):
        _check(load().deqsci_absmax_f32(x.data_ptr(), n, x.numel() // n, _rng(slots, n), _stream()), "absmax")
    return slots


class Sp16:
    """An activation (n,64,H,W) in the "sp16" layout of csrc/conv_s16.hip: t is (n, 4, 2, 2, H, W, 8) float16 =
    [cin chunk][piece: hi, lo][8-channel block][H][W][8 channels], holding 2^e x as hi + lo, e from the range (rng, exp) (see above).
    Only exists between 64->64 layers."""
    __slots__ = ("t", "n", "H", "W", "rng", "exp")

    def __init__(self, t, n, H, W, rng=None, exp=SP16_DEFAULT_EXP):
        self.t, self.n, self.H, self.W, self.rng, self.exp = t, n, H, W, rng, exp

    is_cuda = property(lambda self: self.t.is_cuda)
    device = property(lambda self: self.t.device)
    shape = property(lambda self: (self.n, 64, self.H, self.W))

    @staticmethod
    def empty(n, H, W, device):
        return Sp16(torch.empty((n, 4, 2, 2, H, W, 8), dtype=torch.float16, device=device), n, H, W)

    def exponents(self):
        """The exponent of every image (host sync when the ranges are device slots: tests / tools)."""
        return [self.exp] * self.n if self.rng is None else [act_exp(v) for v in self.rng.tolist()]

    def to_nchw(self):
        """(tests / tools) back to an fp32 (n,64,H,W) channels_last tensor: (hi + lo) / 2^e(image), exactly."""
        sc = torch.tensor([2.0 ** (-e) for e in self.exponents()], dtype=torch.float32, device=self.t.device).view(-1, 1, 1, 1, 1, 1)
        v = (self.t[:, :, 0].float() + self.t[:, :, 1].float()) * sc                               # (n, 4, 2, H, W, 8)
        return v.permute(0, 1, 2, 5, 3, 4).reshape(self.n, 64, self.H, self.W).contiguous(memory_format=torch.channels_last)


def to_split16(x, out=None, rng=None, exp=SP16_DEFAULT_EXP):
    """x (n,64,H,W) fp32 channels_last -> Sp16 with the range (rng, exp) (HIP streaming kernel)."""
    n, c, H, W = x.shape
    if c != 64 or not x.is_contiguous(memory_format=torch.channels_last) or x.dtype != torch.float32 or not x.is_cuda:
        raise DeqsciHipError("to_split16: fp32 channels_last GPU activation with 64 channels required")
    o = out if out is not None else Sp16.empty(n, H, W, x.device)
    o.rng, o.exp = rng, int(exp)
    with _dev(x):
        _check(load().deqsci_f32_to_split16(x.data_ptr(), o.t.data_ptr(), n, H, W, _rng(rng, n), o.exp, _stream()), "f32_to_split16")
    return o


def _weight_exp(w):
    """The power of two that puts max |w| into [2^13, 2^14): the lo pieces of every weight that matters are then normal fp16 numbers."""
    import math
    amax = float(w.abs().max())
    return 0 if amax == 0.0 or not math.isfinite(amax) else 13 - math.floor(math.log2(amax))


class Split16Weights:
    """(64,64,3,3) fp32 conv weight as two fp16 pieces of 2^sw w in the LDS order of csrc/conv_s16.hip:
    [cin chunk c (4)][tap (9)][piece: hi, lo (2)][cout group g (2)][lane (64)][j (8)], cout = 32 g + lane % 32, cin = 16 c + 8 (lane // 32) + j."""
    __slots__ = ("packed", "sw")

    def __init__(self, w):
        if tuple(w.shape) != (64, 64, 3, 3):
            raise DeqsciHipError(f"split16 conv expects a (64,64,3,3) weight, got {tuple(w.shape)}")
        w = w.detach().float()
        self.sw = _weight_exp(w)
        ws = w * (2.0 ** self.sw)
        hi = ws.half()
        lo = (ws - hi.float()).half()
        if not bool(torch.isfinite(hi).all()):
            raise DeqsciHipError("split16 weights overflow fp16")
        p = torch.stack((hi, lo), 0)                                   # (hl, cout, cin, ky, kx)
        p = p.reshape(2, 2, 32, 4, 2, 8, 9)                            # [hl][g][m][c][kb][j][tap]
        self.packed = p.permute(3, 6, 0, 1, 4, 2, 5).contiguous()      # [c][tap][hl][g][kb][m][j]  (lane = 32 kb + m)


class TailSplit16Weights:
    """The last layer's (COUT,64,3,3) weight, COUT = 4 (FFDNet) or 1 (SimpleCNN), for the MFMA tail of csrc/conv_s16.hip: the taps go into
    the matrix N dimension - column 32 nt + lane % 32 = COUT tap + cout - as two fp16 pieces of 2^sw w in the order
    [cin chunk c (4)][piece (2)][N tile nt][lane (64)][j (8)], cin = 16 c + 8 (lane // 32) + j; columns >= 9 COUT are zero."""
    __slots__ = ("packed", "cout", "sw")

    def __init__(self, w):
        cout = w.shape[0]
        if tuple(w.shape) != (cout, 64, 3, 3) or cout not in (1, 4):
            raise DeqsciHipError(f"split16 tail expects a (4,64,3,3) or (1,64,3,3) weight, got {tuple(w.shape)}")
        w = w.detach().float()
        self.sw = _weight_exp(w)
        nt = (9 * cout + 31) // 32
        cols = torch.zeros(32 * nt, 64, dtype=torch.float32, device=w.device)             # [col][cin]
        cols[:9 * cout] = (w * 2.0 ** self.sw).permute(2, 3, 0, 1).reshape(9 * cout, 64)  # col = cout_count * tap + cout
        hi = cols.half()
        lo = (cols - hi.float()).half()
        p = torch.stack((hi, lo), 0).reshape(2, nt, 32, 4, 2, 8)                         # [hl][nt][m][c][kb][j]
        self.packed = p.permute(3, 0, 1, 4, 2, 5).contiguous()                           # [c][hl][nt][kb][m][j]  (lane = 32 kb + m)
        self.cout = cout


class HeadSplit16Weights:
    """FFDNet's first-layer (64,5,3,3) weight for the MFMA head of csrc/conv_s16.hip: two fp16 pieces of 2^sw w as
    [k step (3)][piece (2)][cout group g (2)][lane (64)][j (8)], cout = 32 g + lane % 32, k = 16 ks + 8 (lane // 32) + j = 9 ch + tap
    (k >= 45: zero)."""
    __slots__ = ("packed", "sw")

    def __init__(self, w):
        if tuple(w.shape) != (64, 5, 3, 3):
            raise DeqsciHipError(f"split16 head expects a (64,5,3,3) weight, got {tuple(w.shape)}")
        w = w.detach().float()
        self.sw = _weight_exp(w)
        wk = torch.zeros(64, 48, dtype=torch.float32, device=w.device)
        wk[:, :45] = (w * 2.0 ** self.sw).reshape(64, 45)                               # k = 9 ch + tap
        hi = wk.half()
        lo = (wk - hi.float()).half()
        p = torch.stack((hi, lo), 0).reshape(2, 2, 32, 3, 2, 8)                           # [hl][g][m][ks][kb][j]
        self.packed = p.permute(3, 0, 1, 4, 2, 5).contiguous()                           # [ks][hl][g][kb][m][j]


def ffdnet_head_split16(x, weights, sigma, out=None, in_rng=None, in_exp=SP16_DEFAULT_EXP, out_rng=None, out_exp=SP16_DEFAULT_EXP, track=None):
    """x (n,1,2H,2W) planar, sigma (n,) or (1,) -> relu(conv3x3(cat(sigma map, pixel_unshuffle(x,2)), w)) as an Sp16, on the f16 matrix
    cores with the split-fp16 arithmetic (`weights` = HeadSplit16Weights(w)).  in_rng: range slot holding max |x| of the image (the
    kernel adds sigma itself); (out_rng, out_exp): the range of the output; track: a slot that receives max |output|."""
    n, c, H2, W2 = x.shape
    if c != 1 or H2 % 2 or W2 % 2 or not isinstance(weights, HeadSplit16Weights):
        raise DeqsciHipError(f"ffdnet_head_split16: (n,1,even,even) image and HeadSplit16Weights required, got {tuple(x.shape)}")
    if sigma.numel() not in (1, n) or sigma.dtype != torch.float32 or not sigma.is_cuda:
        raise DeqsciHipError("ffdnet_head_split16: sigma must be a fp32 GPU tensor with 1 or n elements")
    H, W = H2 // 2, W2 // 2
    o = out if out is not None else Sp16.empty(n, H, W, x.device)
    o.rng, o.exp = out_rng, int(out_exp)
    wp = weights.packed if weights.packed.device == x.device else weights.packed.to(x.device)
    with _dev(x):
        _check(load().deqsci_ffdnet_head_split16(_p(x, "x"), wp.data_ptr(), sigma.data_ptr(), 0 if sigma.numel() == 1 else sigma.stride(0),
                                                 o.t.data_ptr(), n, H, W, weights.sw, _rng(in_rng, n), int(in_exp), _rng(out_rng, n), o.exp,
                                                 _rng(track, n), _stream()), "ffdnet_head_split16")
    return o


def tail_split16(h, weights, out=None):
    """h Sp16 -> the denoiser's last layer on the f16 matrix cores: COUT = 4: planar noise (n,1,2H,2W) = pixel_shuffle(conv3x3(h, w, pad=1), 2)
    (FFDNet); COUT = 1: (n,1,H,W) = conv3x3(h, w, pad=1) (SimpleCNN).  `weights` = TailSplit16Weights(w)."""
    if not isinstance(h, Sp16) or not isinstance(weights, TailSplit16Weights):
        raise DeqsciHipError("tail_split16: an Sp16 activation and TailSplit16Weights are required")
    f = 2 if weights.cout == 4 else 1
    o = out if out is not None else torch.empty((h.n, 1, f * h.H, f * h.W), device=h.t.device, dtype=torch.float32)
    wp = weights.packed if weights.packed.device == h.t.device else weights.packed.to(h.t.device)
    fn = load().deqsci_ffdnet_tail_split16 if weights.cout == 4 else load().deqsci_conv3x3_c64_to_1_split16
    with _dev(h.t):
        _check(fn(h.t.data_ptr(), wp.data_ptr(), _p(o, "out"), h.n, h.H, h.W, weights.sw, _rng(h.rng, h.n), h.exp, _stream()), "tail_split16")
    return o


def conv3x3_c64_split16(x, weights, bias=None, relu=True, out=None, out_f32=False, events=None, out_rng=None, out_exp=SP16_DEFAULT_EXP,
                        track=None):
    """x Sp16 -> relu(conv3x3(x, w, pad=1) + bias) as an Sp16 with the range (out_rng, out_exp) (out_f32=False) or an fp32 channels_last
    (n,64,H,W) tensor: the direct convolution on the f16 matrix cores with split operands (csrc/conv_s16.hip).  `weights` =
    Split16Weights(w).  track: a range slot - a MEASURING launch: max |output| is folded into the slot, nothing else is written, None is
    returned (run it, then the real launch with out_rng = that slot)."""
    if not isinstance(x, Sp16) or tuple(x.t.shape) != (x.n, 4, 2, 2, x.H, x.W, 8) or not x.t.is_contiguous() or x.t.dtype != torch.float16 or not x.t.is_cuda:
        raise DeqsciHipError("conv3x3_c64_split16: a contiguous Sp16 GPU activation is required")
    if not isinstance(weights, Split16Weights) or weights.packed.numel() != 4 * 9 * 2 * 2 * 64 * 8:
        raise DeqsciHipError("conv3x3_c64_split16: weights must be a Split16Weights")
    n, H, W, dev = x.n, x.H, x.W, x.t.device
    wp = weights.packed if weights.packed.device == dev else weights.packed.to(dev)
    if track is not None:
        if out_f32:
            raise DeqsciHipError("conv3x3_c64_split16: a measuring launch (track=) serves the sp16 output")
        with _dev(x.t):
            _check(load().deqsci_conv3x3_c64_split16(x.t.data_ptr(), wp.data_ptr(), _p(bias, "bias", True), None, n, H, W, 1 if relu else 0,
                                                     weights.sw, _rng(x.rng, n), x.exp, None, 0, _rng(track, n), 0, _stream(), None, None),
                   "conv3x3_c64_split16 (measuring)")
        return None
    if out_f32:
        o = out if out is not None else torch.empty((n, 64, H, W), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        ot = o
    else:
        o = out if out is not None else Sp16.empty(n, H, W, dev)
        o.rng, o.exp = out_rng, int(out_exp)
        ot = o.t
    ev = _hook_events("s16", n, H, W, events) or (None, None)
    with _dev(x.t):
        _check(load().deqsci_conv3x3_c64_split16(x.t.data_ptr(), wp.data_ptr(), _p(bias, "bias", True), ot.data_ptr(), n, H, W, 1 if relu else 0,
                                                 weights.sw, _rng(x.rng, n), x.exp, _rng(out_rng, n), int(out_exp), None, 1 if out_f32 else 0,
                                                 _stream(), ev[0], ev[1]), "conv3x3_c64_split16")
    return o


class Split16Stack:
    """A RUN of 64->64 layers for deqsci_conv3x3_c64_split16_stack: the device table of (weights, bias, w_exp, relu) per layer - three
    8-byte words each - the tensors it points to (kept alive here), and per launch shape the progress words of the tiles (zeroed once:
    they count on from launch to launch) and the two ping-pong buffers (kept: a captured hipGraph carries their addresses)."""
    __slots__ = ("table", "n_layers", "keep", "_state", "act")
    TILE = (16, 32)                                                   # block tile of the kernel: rows x columns of output pixels

    def __init__(self, layers, device):
        """layers: [(Split16Weights, bias tensor or None, relu), ...]"""
        rows, keep = [], []
        self.act = Sp16
        for w16, bias, relu in layers:
            if not isinstance(w16, self._weights_class()):
                raise DeqsciHipError(f"{type(self).__name__}: every layer needs {self._weights_class().__name__}")
            wp = w16.packed if w16.packed.device == torch.device(device) else w16.packed.to(device)
            b = None if bias is None else f32c(bias.detach().to(device))
            if b is not None and b.numel() < 64:
                raise DeqsciHipError("Split16Stack: bias must have 64 elements")
            keep += [wp, b]
            rows += [wp.data_ptr(), 0 if b is None else b.data_ptr(), (int(w16.sw) & 0xffffffff) | ((1 if relu else 0) << 32)]
        self.table = torch.tensor(rows, dtype=torch.int64).to(device)
        self.n_layers, self.keep, self._state = len(layers), keep, {}

    @staticmethod
    def _weights_class():
        return Split16Weights

    def state(self, n, H, W):
        """(Sp16, Sp16) ping-pong outputs of a batch of n images (kept: a captured hipGraph carries their addresses)."""
        st = self._state.get(("out", n, H, W))
        if st is None:
            dev = self.table.device
            for key in [k for k in self._state if k[0] == "out"]:      # one live batch shape at a time (2 x 256 bytes per pixel and image)
                del self._state[key]
            st = self._state[("out", n, H, W)] = (self.act.empty(n, H, W, dev), self.act.empty(n, H, W, dev))
        return st

    def head_buffer(self, n, H, W):
        """An Sp16 for the run's INPUT of a slice of n images (the engine runs first layer -> run -> last layer slice by slice, so that each
        hands its output to the next through the Infinity Cache); kept like the output buffers."""
        hb = self._state.get(("head", n, H, W))
        if hb is None:
            for key in [k for k in self._state if k[0] == "head"]:
                del self._state[key]
            hb = self._state[("head", n, H, W)] = self.act.empty(n, H, W, self.table.device)
        return hb

    def flags(self, n, H, W):
        """The progress words of a LAUNCH of n images: 32 (n_tiles + 1) words (a 128-byte line per tile + the time-out word), zeroed once.
        Launches of one shape share them (they run one after the other on a stream and each advances every word by n_layers)."""
        fl = self._state.get(("flags", n, H, W))
        if fl is None:
            n_tiles = n * (-(-H // self.TILE[0])) * (-(-W // self.TILE[1]))
            fl = self._state[("flags", n, H, W)] = torch.zeros(32 * (n_tiles + 1), dtype=torch.int32, device=self.table.device)
        return fl

    def timed_out(self):
        """(host sync) True if a wait of any launch since the last call timed out - the outputs since then are invalid; the words are
        rearmed."""
        bad = False
        for key, fl in self._state.items():
            if key[0] == "flags" and int(fl[-32]) != 0:
                fl.zero_()
                bad = True
        return bad


STACK_SLICE_BYTES = 128 << 20      # activation bytes of one stack launch: its two ping-pong buffers share the 256 MiB Infinity Cache


def split16_stack_per_launch(n, H, W, slice_bytes=STACK_SLICE_BYTES, cus=None, tile=(16, 32)):
    """Images per stack launch for a batch of n images of H x W: as many as keep one activation (256 bytes per pixel) within `slice_bytes`
    (32 images of 128 x 128) - rounded down to a whole number of 16 x 32 tiles per CU when the device's CU count is given (a launch of
    2.75 tiles per workgroup takes as long as one of 3): a batch goes out as full slices and a remainder (40 images: 32 + 8).  Measured
    on MI355X at 256 x 256 x 8, FFDNet, 8 measurements per call: slices of 32 / 16 images 148 / 147 frames/s, of 22 / 11 images (704 / 352
    tiles on 256 CUs) 140 / 122, of 40-64 images 138-141 (as with a launch per layer)."""
    per = max(1, int(slice_bytes) // (H * W * 256))
    if cus:
        tiles = (-(-H // tile[0])) * (-(-W // tile[1]))
        unit = int(cus) // math.gcd(tiles, int(cus))             # images whose tiles are a multiple of the CUs
        if per >= unit:
            per -= per % unit
    return min(per, n)


def conv3x3_c64_split16_stack(x, stack, ranges=None, events=None, per_launch=None, rng_offset=0, out_bufs=None, check=None):
    """x Sp16 -> the run of 64->64 layers `stack` (Split16Stack), each launch a whole run (csrc/conv_s16.hip, STACK: the persistent
    workgroups walk their tiles layer after layer, a tile waiting for the layer before of itself and its eight neighbours) over a slice
    of the batch - per_launch images (None: split16_stack_per_launch, slices that fit the Infinity Cache), one launch after the other;
    ranges: the (n_layers + 1, n) range slots of the run - its input's first - or None (fixed exponents: the input's, then 2^8).
    Returns the Sp16 the last layer wrote (one of the stack's two buffers of this shape).  stack.timed_out() afterwards tells whether a
    wait gave up (foreign work on the device's CUs): the result is invalid then.  x may itself be a slice of a larger batch: then
    `ranges` are the whole batch's (n_layers + 1, n_total) slots and rng_offset the slice's first image in them; out_bufs: two Sp16 of at
    least x.n images to write into instead of the stack's own."""
    if not isinstance(x, Sp16) or not isinstance(stack, Split16Stack) or not x.t.is_contiguous() or x.t.dtype != torch.float16 or not x.t.is_cuda:
        raise DeqsciHipError("conv3x3_c64_split16_stack: a contiguous Sp16 GPU activation and a Split16Stack are required")
    n, H, W = x.n, x.H, x.W
    if x.t.device != stack.table.device:
        raise DeqsciHipError("conv3x3_c64_split16_stack: the stack was built for another device")
    n_total = n if ranges is None or not isinstance(ranges, torch.Tensor) or ranges.dim() != 2 else ranges.shape[1]
    if ranges is not None and (not isinstance(ranges, torch.Tensor) or ranges.dtype != torch.float32 or ranges.dim() != 2
                               or ranges.shape[0] != stack.n_layers + 1 or rng_offset < 0 or rng_offset + n > n_total
                               or not ranges.is_contiguous() or ranges.device != x.t.device):
        raise DeqsciHipError(f"conv3x3_c64_split16_stack: ranges must be a contiguous fp32 ({stack.n_layers + 1}, >= {rng_offset + n}) tensor on the input's device")
    if (ranges is None) != (x.rng is None):
        raise DeqsciHipError("conv3x3_c64_split16_stack: the input's range and the run's ranges go together (both measured or both fixed)")
    per = (split16_stack_per_launch(n, H, W, cus=torch.cuda.get_device_properties(x.t.device).multi_processor_count) if per_launch is None
           else int(per_launch))
    if per <= 0:
        raise DeqsciHipError("conv3x3_c64_split16_stack: per_launch must be positive")
    bufs = stack.state(n, H, W) if out_bufs is None else out_bufs
    if len(bufs) != 2 or any(not isinstance(b, Sp16) or b.n < n or (b.H, b.W) != (H, W) or not b.t.is_contiguous() or b.t.device != x.t.device for b in bufs):
        raise DeqsciHipError("conv3x3_c64_split16_stack: out_bufs must be two contiguous Sp16 of the input's H x W with at least its images")
    with _dev(x.t):
        for a in range(0, n, per):
            m = min(per, n - a)
            ev = _hook_events("s16stack", m, H, W, events, layers=stack.n_layers) or (None, None)
            _check(load().deqsci_conv3x3_c64_split16_stack(x.t[a:a + m].data_ptr(), bufs[0].t[a:a + m].data_ptr(), bufs[1].t[a:a + m].data_ptr(),
                                                           stack.table.data_ptr(), stack.n_layers, m, H, W,
                                                           None if ranges is None else ranges.data_ptr() + 4 * (rng_offset + a), n_total, x.exp, SP16_DEFAULT_EXP,
                                                           stack.flags(m, H, W).data_ptr(), _stream(), ev[0], ev[1]), "conv3x3_c64_split16_stack")
    out = bufs[(stack.n_layers - 1) % 2]
    if out.n != n:
        out = Sp16(out.t[:n], n, H, W)
    out.rng, out.exp = (None if ranges is None else ranges[stack.n_layers][rng_offset:rng_offset + n]), SP16_DEFAULT_EXP
    if check is None:
        check = not torch.cuda.is_current_stream_capturing()
    if check and stack.timed_out():                             # (one host sync; DEQSCIEngine passes check=False and looks once per reconstruction)
        raise DeqsciHipError("a wait inside the stack launch timed out - its workgroups were not all resident (the device's CUs are shared with "
                             "other work): the output of this call is invalid; use one launch per layer")
    return out


# ----------------------------------------------------------------------------- split-fp16 Winograd F(2,3) x direct (csrc/conv_w16.hip)

class P32:
    """An activation (n,64,H,W) in the "p32" layout of csrc/conv_w16.hip: t is (n, 8, 2, H, ceil(W/64), 2, 32, 4) float32 =
    [8-channel block][half of 4][H][block of 64 columns][column parity][32][4 channels] holding 2^e x, e from the range (rng, exp) exactly
    as for Sp16 - the 16 planes of 16-byte pixels of sp16, unsplit, and inside every block of 64 columns the even columns first, then the
    odd ones (the Winograd kernel's lanes own column PAIRS: with the parities apart its loads and stores are contiguous).  Columns >= W of
    the last block are padding (never read as data).  Only exists between 64->64 layers."""
    __slots__ = ("t", "n", "H", "W", "rng", "exp")

    def __init__(self, t, n, H, W, rng=None, exp=SP16_DEFAULT_EXP):
        self.t, self.n, self.H, self.W, self.rng, self.exp = t, n, H, W, rng, exp

    is_cuda = property(lambda self: self.t.is_cuda)
    device = property(lambda self: self.t.device)
    shape = property(lambda self: (self.n, 64, self.H, self.W))

    @staticmethod
    def empty(n, H, W, device):
        return P32(torch.empty((n, 8, 2, H, -(-W // 64), 2, 32, 4), dtype=torch.float32, device=device), n, H, W)

    def exponents(self):
        return [self.exp] * self.n if self.rng is None else [act_exp(v) for v in self.rng.tolist()]

    def to_nchw(self):
        """(tests / tools) back to an fp32 (n,64,H,W) channels_last tensor: t / 2^e(image), exactly."""
        sc = torch.tensor([2.0 ** (-e) for e in self.exponents()], dtype=torch.float32, device=self.t.device).view(-1, 1, 1, 1)
        nb = self.t.shape[4]
        v = self.t.permute(0, 1, 2, 7, 3, 4, 6, 5).reshape(self.n, 64, self.H, nb * 64)[..., :self.W]     # (n, b8, j, k, H, blk, i, par) -> col = 64 blk + 2 i + par
        return (v * sc).contiguous(memory_format=torch.channels_last)

    @staticmethod
    def from_nchw(x, rng=None, exp=SP16_DEFAULT_EXP):
        """(tests / tools; torch ops) x (n,64,H,W) fp32 -> P32 holding 2^e x (the padding columns zero)."""
        n, c, H, W = x.shape
        o = P32.empty(n, H, W, x.device)
        o.rng, o.exp = rng, int(exp)
        sc = torch.tensor([2.0 ** e for e in o.exponents()], dtype=torch.float32, device=x.device).view(-1, 1, 1, 1)
        nb = o.t.shape[4]
        xp = torch.zeros((n, 64, H, nb * 64), dtype=torch.float32, device=x.device)
        xp[..., :W] = x * sc
        o.t.copy_(xp.reshape(n, 8, 2, 4, H, nb, 32, 2).permute(0, 1, 2, 4, 5, 7, 6, 3))
        return o


_W16_G = ((1.0, 0.0, 0.0), (0.5, 0.5, 0.5), (0.5, -0.5, 0.5), (0.0, 0.0, 1.0))      # G of Winograd F(2,3)


class Wino16Weights:
    """(64,64,3,3) fp32 conv weight for csrc/conv_w16.hip: U[dy][xi] = sum_dx G[xi][dx] w[:, :, dy, dx] formed in float64, rounded to fp32,
    as two fp16 pieces of 2^sw U (max |U| in [2^13, 2^14)) in the kernel's LDS order
    [cin chunk c (4)][xi half h (2)][xi' (2)][dy (3)][piece: hi, lo (2)][cout group g (2)][lane (64)][j (8)],
    xi = 2 h + xi', cout = 32 g + lane % 32, cin = 16 c + 8 (lane // 32) + j."""
    __slots__ = ("packed", "sw")

    def __init__(self, w):
        if tuple(w.shape) != (64, 64, 3, 3):
            raise DeqsciHipError(f"wino16 conv expects a (64,64,3,3) weight, got {tuple(w.shape)}")
        g = torch.tensor(_W16_G, dtype=torch.float64, device=w.device)
        u = torch.einsum('xk,ocyk->yxoc', g, w.detach().double()).float()           # [dy][xi][cout][cin]
        self.sw = _weight_exp(u)
        us = u * (2.0 ** self.sw)
        hi = us.half()
        lo = (us - hi.float()).half()
        if not bool(torch.isfinite(hi).all()):
            raise DeqsciHipError("wino16 weights overflow fp16")
        p = torch.stack((hi, lo), 0)                                   # (hl, dy, xi, cout, cin)
        p = p.reshape(2, 3, 2, 2, 2, 32, 4, 2, 8)                      # [hl][dy][h][xp][g][m][c][kb][j]
        self.packed = p.permute(6, 2, 3, 1, 0, 4, 7, 5, 8).contiguous()   # [c][h][xp][dy][hl][g][kb][m][j]  (lane = 32 kb + m)


def _act_check(x, what):
    if (not isinstance(x, P32) or tuple(x.t.shape) != (x.n, 8, 2, x.H, -(-x.W // 64), 2, 32, 4) or x.t.dtype != torch.float32 or not x.t.is_contiguous()
            or not x.t.is_cuda):
        raise DeqsciHipError(f"{what}: a contiguous P32 GPU activation is required")


def conv3x3_c64_wino16(x, weights, bias=None, relu=True, out=None, events=None, out_rng=None, out_exp=SP16_DEFAULT_EXP):
    """x P32 -> relu(conv3x3(x, w, pad=1) + bias) as a P32 with the range (out_rng, out_exp): the split-fp16 arithmetic under Winograd F(2,3)
    along x nested in the direct sum along y (csrc/conv_w16.hip).  `weights` = Wino16Weights(w)."""
    _act_check(x, "conv3x3_c64_wino16")
    if not isinstance(weights, Wino16Weights) or weights.packed.numel() != 4 * 2 * 2 * 3 * 2 * 2 * 64 * 8:
        raise DeqsciHipError("conv3x3_c64_wino16: weights must be a Wino16Weights")
    n, H, W, dev = x.n, x.H, x.W, x.t.device
    wp = weights.packed if weights.packed.device == dev else weights.packed.to(dev)
    o = out if out is not None else P32.empty(n, H, W, dev)
    if not isinstance(o, P32):
        raise DeqsciHipError("conv3x3_c64_wino16: out must be a P32")
    o.rng, o.exp = out_rng, int(out_exp)
    ev = _hook_events("w16", n, H, W, events) or (None, None)
    with _dev(x.t):
        _check(load().deqsci_conv3x3_c64_wino16(x.t.data_ptr(), wp.data_ptr(), _p(bias, "bias", True), o.t.data_ptr(), n, H, W, 1 if relu else 0,
                                                weights.sw, _rng(x.rng, n), x.exp, _rng(out_rng, n), int(out_exp), _stream(), ev[0], ev[1]),
               "conv3x3_c64_wino16")
    return o


def ffdnet_head_p32(x, weights, sigma, out=None, in_rng=None, in_exp=SP16_DEFAULT_EXP, out_rng=None, out_exp=SP16_DEFAULT_EXP):
    """ffdnet_head_split16 writing a P32 (the input of a run of conv3x3_c64_wino16 layers): same weights, same arithmetic, 2^e y unsplit."""
    n, c, H2, W2 = x.shape
    if c != 1 or H2 % 2 or W2 % 2 or not isinstance(weights, HeadSplit16Weights):
        raise DeqsciHipError(f"ffdnet_head_p32: (n,1,even,even) image and HeadSplit16Weights required, got {tuple(x.shape)}")
    if sigma.numel() not in (1, n) or sigma.dtype != torch.float32 or not sigma.is_cuda:
        raise DeqsciHipError("ffdnet_head_p32: sigma must be a fp32 GPU tensor with 1 or n elements")
    H, W = H2 // 2, W2 // 2
    o = out if out is not None else P32.empty(n, H, W, x.device)
    if not isinstance(o, P32) or (o.n, o.H, o.W) != (n, H, W):
        raise DeqsciHipError("ffdnet_head_p32: out must be a P32 of the output's shape")
    o.rng, o.exp = out_rng, int(out_exp)
    wp = weights.packed if weights.packed.device == x.device else weights.packed.to(x.device)
    with _dev(x):
        _check(load().deqsci_ffdnet_head_p32(_p(x, "x"), wp.data_ptr(), sigma.data_ptr(), 0 if sigma.numel() == 1 else sigma.stride(0),
                                             o.t.data_ptr(), n, H, W, weights.sw, _rng(in_rng, n), int(in_exp), _rng(out_rng, n), o.exp,
                                             _stream()), "ffdnet_head_p32")
    return o


def ffdnet_tail_p32(h, weights, out=None):
    """tail_split16 reading a P32: COUT = 4 (FFDNet's last layer + pixel shuffle) or 1 (a plain 64 -> 1 layer: SimpleCNN's last);
    `weights` = TailSplit16Weights(w)."""
    _act_check(h, "ffdnet_tail_p32")
    if not isinstance(weights, TailSplit16Weights) or weights.cout not in (1, 4):
        raise DeqsciHipError("ffdnet_tail_p32: TailSplit16Weights of a (4,64,3,3) or (1,64,3,3) weight are required")
    shape = (h.n, 1, 2 * h.H, 2 * h.W) if weights.cout == 4 else (h.n, 1, h.H, h.W)
    o = out if out is not None else torch.empty(shape, device=h.t.device, dtype=torch.float32)
    wp = weights.packed if weights.packed.device == h.t.device else weights.packed.to(h.t.device)
    fn = load().deqsci_ffdnet_tail_p32 if weights.cout == 4 else load().deqsci_conv3x3_c64_to_1_p32
    with _dev(h.t):
        _check(fn(h.t.data_ptr(), wp.data_ptr(), _p(o, "out"), h.n, h.H, h.W, weights.sw, _rng(h.rng, h.n), h.exp, _stream()), "tail_p32")
    return o


class Wino16Stack(Split16Stack):
    """A RUN of 64->64 layers for deqsci_conv3x3_c64_wino16_stack: Split16Stack with Wino16Weights, block tiles of 8 x 64 pixels and P32
    activations."""
    __slots__ = ()
    TILE = (8, 64)

    def __init__(self, layers, device):
        super().__init__(layers, device)
        self.act = P32

    @staticmethod
    def _weights_class():
        return Wino16Weights


def conv3x3_c64_wino16_stack(x, stack, ranges=None, events=None, per_launch=None, rng_offset=0, out_bufs=None, check=None):
    """conv3x3_c64_split16_stack on the Winograd kernel: x P32 -> the run of 64->64 layers `stack` (Wino16Stack), each launch a whole run
    over a slice of the batch; same arguments, same time-out contract."""
    _act_check(x, "conv3x3_c64_wino16_stack")
    if not isinstance(stack, Wino16Stack):
        raise DeqsciHipError("conv3x3_c64_wino16_stack: a Wino16Stack is required")
    n, H, W = x.n, x.H, x.W
    if x.t.device != stack.table.device:
        raise DeqsciHipError("conv3x3_c64_wino16_stack: the stack was built for another device")
    n_total = n if ranges is None or not isinstance(ranges, torch.Tensor) or ranges.dim() != 2 else ranges.shape[1]
    if ranges is not None and (not isinstance(ranges, torch.Tensor) or ranges.dtype != torch.float32 or ranges.dim() != 2
                               or ranges.shape[0] != stack.n_layers + 1 or rng_offset < 0 or rng_offset + n > n_total
                               or not ranges.is_contiguous() or ranges.device != x.t.device):
        raise DeqsciHipError(f"conv3x3_c64_wino16_stack: ranges must be a contiguous fp32 ({stack.n_layers + 1}, >= {rng_offset + n}) tensor on the input's device")
    if (ranges is None) != (x.rng is None):
        raise DeqsciHipError("conv3x3_c64_wino16_stack: the input's range and the run's ranges go together (both measured or both fixed)")
    per = (split16_stack_per_launch(n, H, W, cus=torch.cuda.get_device_properties(x.t.device).multi_processor_count, tile=stack.TILE)
           if per_launch is None else int(per_launch))
    if per <= 0:
        raise DeqsciHipError("conv3x3_c64_wino16_stack: per_launch must be positive")
    bufs = stack.state(n, H, W) if out_bufs is None else out_bufs
    if len(bufs) != 2 or any(type(b) is not stack.act or b.n < n or (b.H, b.W) != (H, W) or not b.t.is_contiguous() or b.t.device != x.t.device for b in bufs):
        raise DeqsciHipError("conv3x3_c64_wino16_stack: out_bufs must be two contiguous activations of the input's format and H x W with at least its images")
    with _dev(x.t):
        for a in range(0, n, per):
            m = min(per, n - a)
            ev = _hook_events("w16stack", m, H, W, events, layers=stack.n_layers) or (None, None)
            _check(load().deqsci_conv3x3_c64_wino16_stack(x.t[a:a + m].data_ptr(), bufs[0].t[a:a + m].data_ptr(), bufs[1].t[a:a + m].data_ptr(),
                                                          stack.table.data_ptr(), stack.n_layers, m, H, W,
                                                          None if ranges is None else ranges.data_ptr() + 4 * (rng_offset + a), n_total, x.exp, SP16_DEFAULT_EXP,
                                                          stack.flags(m, H, W).data_ptr(), _stream(), ev[0], ev[1]), "conv3x3_c64_wino16_stack")
    out = bufs[(stack.n_layers - 1) % 2]
    if out.n != n:
        out = stack.act(out.t[:n], n, H, W)
    out.rng, out.exp = (None if ranges is None else ranges[stack.n_layers][rng_offset:rng_offset + n]), SP16_DEFAULT_EXP
    if check is None:
        check = not torch.cuda.is_current_stream_capturing()
    if check and stack.timed_out():                             # (one host sync; DEQSCIEngine passes check=False and looks once per reconstruction)
        raise DeqsciHipError("a wait inside the stack launch timed out - its workgroups were not all resident (the device's CUs are shared with "
                             "other work): the output of this call is invalid; use one launch per layer")
    return out


class Conv64Weights:
    """The weights of one 64->64 layer for all three kernels.  The split-fp16 pack (a host sync: max |w|) is made here when `s16` is set
    - the engine does whenever its policy can pick that kernel, so that the pack never falls inside a hipGraph capture - or on first use."""
    __slots__ = ("f22", "f44", "_w", "_s16", "_w16")

    def __init__(self, w, s16=False):
        self.f22 = pack_winograd_weights(w)
        self.f44 = pack_winograd44_weights(w)
        self._w, self._s16, self._w16 = w.detach(), None, None
        if s16:
            self._s16 = Split16Weights(self._w)
            self._w16 = Wino16Weights(self._w)

    @property
    def w16(self):
        if self._w16 is None:
            self._w16 = Wino16Weights(self._w)
        return self._w16

    @property
    def s16(self):
        if self._s16 is None:
            self._s16 = Split16Weights(self._w)
        return self._s16


def pack_conv64_weights(w, s16=False):
    return Conv64Weights(w, s16=s16)


# launch time of one block tile per CU, us (tools/w44_check.py, tools/s16_check.py; profiles/r02_w44_shapes.jsonl, r03_s16_*): the F(4x4,3x3)
# and the split-fp16 kernels do 2x the pixels of F(2x2,3x3) per tile
_T_TILE_F22, _T_TILE_F44, _T_TILE_S16 = 21.5, 33.5, 29.5


W44_MAX_PIXELS = (0x80000000 - 4096 - 2048 - 16) // 256   # per image, width padded to 32 columns: the F(4x4,3x3) launcher's limit
S16_MAX_PIXELS = (0x80000000 - 4096 - 4096 - 16) // 256   # per image (H * W): the split-fp16 launcher's (csrc/conv_s16.hip: RAW_BIAS + 4096 + 16)
FORCE_CONV64 = None                                     # "f22" / "f44" / "s16": set by the tools for A/B runs (never by the package or the environment); overrides every policy below


_CUS = {}


def _cus(device):
    idx = torch.cuda.current_device() if device is None or getattr(device, "index", None) is None else device.index
    if idx not in _CUS:                                  # (get_device_properties costs tens of microseconds: not once per layer call)
        _CUS[idx] = torch.cuda.get_device_properties(idx).multi_processor_count
    return _CUS[idx]


def conv64_kernel_for(n, H, W, device=None, policy="fast"):
    """'s16', 'f44' or 'f22' for a 64->64 layer on n images of H x W.  All three kernels run one persistent workgroup per CU over
    block tiles of 16 x 16 (F(2x2,3x3)) / 16 x 32 (F(4x4,3x3), split-fp16) output pixels, so a launch takes (waves of block tiles) x
    (time of a tile).
      "fast"    the faster of the split-fp16 direct convolution (csrc/conv_s16.hip) and Winograd F(2x2,3x3) (csrc/winograd.hip): split-fp16
                from about one block tile per CU on (6 images of 128 x 128), F(2x2,3x3) below
      "fast32"  fp32 MFMA arithmetic only: the faster of F(4x4,3x3) (csrc/winograd44.hip) and F(2x2,3x3)
      "f22" / "f44" / "s16"   that kernel whatever the size.
    Rounding per layer against a float64 convolution on FFDNet's own data (tools/conv_error_real.py, profiles/r03_conv_error_real.json):
    split-fp16 1.6e-7, F(2x2,3x3) 2.0e-7, F(4x4,3x3) 2.2e-7 (5.6e-7 on the blocky first iterate), MIOpen's direct fp32 convolution 3.5e-7."""
    if FORCE_CONV64 in ("f22", "f44", "s16"):
        return FORCE_CONV64
    if policy in ("f22", "f44", "s16"):
        return policy
    if policy not in ("fast", "fast32"):
        raise DeqsciHipError(f"conv64 policy {policy!r}: expected 'fast', 'fast32', 'f22', 'f44' or 's16'")
    # beyond the 32-bit buffer offsets of the 16 x 32-tile kernel this policy would pick (its launcher refuses): F(2x2,3x3)
    if (H * (-(-W // 32)) * 32 > W44_MAX_PIXELS) if policy == "fast32" else (H * W > S16_MAX_PIXELS):
        return "f22"
    cus = _cus(device)
    t22 = -(-(n * (-(-H // 16)) * (-(-W // 16))) // cus) * _T_TILE_F22
    big = -(-(n * (-(-H // 16)) * (-(-W // 32))) // cus)
    if policy == "fast32":
        return "f44" if big * _T_TILE_F44 < t22 else "f22"
    return "s16" if big * _T_TILE_S16 < t22 else "f22"


def conv3x3_c64(x, weights, bias=None, relu=True, out=None, policy="fast", chain=False):
    """relu(conv3x3(x, w, pad=1) + bias) for a 64->64 layer on the kernel conv64_kernel_for(..., policy) names (`weights` =
    pack_conv64_weights(w)).  x: fp32 channels_last (n,64,H,W), or the previous layer's output in its kernel's own layout (Sp16 /
    Blk32: that kernel runs again).  chain=True leaves the result in the kernel's own layout for the next 64->64 layer (Sp16 for the
    split-fp16 kernel, Blk32 for F(4x4,3x3); F(2x2,3x3) has none); chain=False returns fp32 channels_last."""
    kind = ("s16" if isinstance(x, Sp16) else "f44" if isinstance(x, Blk32) else
            conv64_kernel_for(x.shape[0], x.shape[2], x.shape[3], x.device, policy))
    if kind == "s16":
        return conv3x3_c64_split16(x if isinstance(x, Sp16) else to_split16(x), weights.s16, bias, relu, out, out_f32=not chain)
    if kind == "f44":
        return conv3x3_c64_winograd44(x, weights.f44, bias, relu, out, out_blk=chain)
    return conv3x3_c64_winograd(x, weights.f22, bias, relu, out)


# ----------------------------------------------------------------------------- measurement helpers (bench.py)
class KernelTimer:
    """Per-launch kernel durations from the dispatch's own start/stop timestamps (the *_timed_f32 entry points record
    HIP events around the kernel on the stream it runs on).  All events are created up front (`capacity` launches) and
    destroyed by close(); launches beyond the capacity run untimed.  Read durations after synchronising the stream."""

    _plain_mix_gap = staticmethod(anderson_mix_gap)          # bound here: callers may monkey-patch the module names

    def __init__(self, capacity=0):
        self.events = []
        self.used = 0
        for _ in range(2 * capacity):
            h = _ptr()
            _check(load().deqsci_event_create(ctypes.byref(h)), "event_create")
            self.events.append(h)

    def _pair(self):
        if 2 * self.used + 1 >= len(self.events):
            return None
        e = self.events[2 * self.used], self.events[2 * self.used + 1]
        self.used += 1
        return e

    def reset(self):
        self.used = 0

    @property
    def full(self):
        return 2 * self.used + 1 >= len(self.events)

    def mix_gap(self, ws, beta, n, phi, y, phisum, x_out, z1, layout):
        ev = self._pair()
        if ev is None:
            return self._plain_mix_gap(ws, beta, n, phi, y, phisum, x_out, z1, layout)
        bsz, H, W, B = _dims(layout, z1.shape)
        with _dev(ws.F):
            _check(load().deqsci_anderson_mix_gap_timed_f32(
                _p(ws.F), _p(ws.G), _p(ws.alpha), float(beta), n, ws.m, _p(phi, "Phi"), _p(y, "y"), _p(phisum, "Phi_sum"),
                _p(x_out, "x_out"), _p(z1, "z1"), bsz, H, W, B, layout, _phi_shared(phi, bsz), _stream(), ev[0], ev[1]),
                "anderson_mix_gap_timed")

    def pair(self):
        """The next unused (start, stop) pair, or None when the capacity is used up (for CONV64_EVENT_HOOK)."""
        return self._pair()

    def durations_ms(self):
        out = []
        ms = _f32()
        for i in range(self.used):
            _check(load().deqsci_event_elapsed_ms(self.events[2 * i], self.events[2 * i + 1], ctypes.byref(ms)), "event_elapsed_ms")
            out.append(ms.value)
        return out

    def close(self):
        for e in self.events:
            load().deqsci_event_destroy(e)
        self.events, self.used = [], 0
