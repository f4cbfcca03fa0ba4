#!/usr/bin/env python3
"""Numerics of csrc/conv_w16.hip BEFORE the kernel: an emulation in torch (CPU is enough) of what it computes for one 64->64 layer -

    the 3x3 convolution as Winograd F(2,3) ALONG X nested in a direct sum ALONG Y (a 2D F(2x2,3x3) needs 16 accumulators per 4 outputs and
    its transforms do not hide on this machine; the nested form needs 4 per 2 and transforms rows, not patches):
        U[dy][xi] = G g[dy, :]           float64 -> fp32 -> x 2^sw (one power of two per layer, max |U| in [2^13, 2^14)) -> hi + lo fp16
        V[row][xi] = B^T d[row, 2t-1 .. 2t+2]      fp32, from the sp16 activation d = hi + lo of 2^e x; then hi + lo fp16
        M[xi] = sum_dy sum_cin U_hi V_hi + U_lo V_hi + U_hi V_lo       fp32 accumulation, ONE chain of 36 steps of K = 16
        y[2t], y[2t+1] = A^T M                       fp32, x 2^(e_out - e_in - sw), + bias, ReLU

on FFDNet's own weights and a real first iterate, layer by layer, against a float64 convolution, beside an emulation of the split-fp16
DIRECT convolution (two chains, as csrc/conv_s16.hip).  Prints one JSON line per layer and a summary."""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def split(t):
    hi = t.half().float()
    return hi, (t - hi).half().float()


def pow2(e):
    return 2.0 ** e


def act_exp(amax):
    return 11 - int(np.floor(np.log2(float(amax))))


def chain_k16(terms):
    """fp32 accumulation of a list of (A, B) products in steps of K = 16 (one rounding per step, as an MFMA does): A (M, K), B (K, N)."""
    acc = None
    for a, b in terms:
        for k in range(0, a.shape[-1], 16):
            p = (a[..., k:k + 16].double() @ b[..., k:k + 16, :].double())
            acc = p.float() if acc is None else (acc.double() + p).float()
    return acc


def wino16_layer(x, w, bias, relu):
    """x (n, 64, H, W) fp32 (even W), w (64, 64, 3, 3) -> the emulated kernel output, fp32."""
    n, C, H, W = x.shape
    U = torch.einsum('xk,ocyk->yxoc', G, w.double()).float()                        # [dy][xi][cout][cin]
    sw = 13 - int(np.floor(np.log2(float(U.abs().max()))))
    Uh, Ul = split(U * pow2(sw))
    out = torch.empty_like(x)
    for i in range(n):
        e = act_exp(x[i].abs().max())
        dh, dl = split(x[i] * pow2(e))
        d = dh + dl                                                                  # what the sp16 activation holds, exactly
        dp = F.pad(d, (1, 1, 1, 1))                                                  # (C, H + 2, W + 2)
        tiles = dp.unfold(2, 4, 2)                                                   # (C, H + 2, W / 2, 4)
        V = torch.einsum('xj,crtj->xrct', BT, tiles)                                 # [xi][row][cin][tile]   fp32 adds: exact? (rounded like the kernel's)
        Vh, Vl = split(V)
        M = []
        for xi in range(4):
            rows = []
            for r in range(H):
                terms = []
                for dy in range(3):
                    terms += [(Ul[dy, xi], Vh[xi, r + dy]), (Uh[dy, xi], Vl[xi, r + dy]), (Uh[dy, xi], Vh[xi, r + dy])]
                rows.append(chain_k16(terms))                                        # (cout, tiles)
            M.append(torch.stack(rows, 1))                                           # (cout, H, tiles)
        M = torch.stack(M, 0)                                                        # (xi, cout, H, tiles)
        y0 = (M[0] + M[1]) + M[2]
        y1 = (M[1] - M[2]) - M[3]
        y = torch.stack((y0, y1), -1).reshape(C, H, W) * pow2(-e - sw)
        if bias is not None:
            y = y + bias.view(-1, 1, 1)
        out[i] = torch.relu(y) if relu else y
    return out


def wino16_fixed_layer(x, w, bias, relu, grid_bits=10):
    """(round 6) the same layer with a CHEAPER input transform: the activation split on a FIXED grid per image - hi = the
    multiple of q = 2^(12 - grid_bits) nearest to 2^e x (at most grid_bits bits below the image's maximum, so that a difference of two hi pieces is
    exact in fp16), lo = fp16(2^e x - hi) - and B^T d formed on the pieces separately in fp16 (v_pk_add_f16: V_hi exact, V_lo rounded to fp16):
    16 packed additions per halo row and half-stage instead of 16 fp32 subtractions + 24 split instructions."""
    n, C, H, W = x.shape
    U = torch.einsum('xk,ocyk->yxoc', G, w.double()).float()
    sw = 13 - int(np.floor(np.log2(float(U.abs().max()))))
    Uh, Ul = split(U * pow2(sw))
    q = pow2(12 - grid_bits)
    out = torch.empty_like(x)
    for i in range(n):
        e = act_exp(x[i].abs().max())
        xs = x[i] * pow2(e)
        dh = torch.round(xs / q) * q                                                 # exact in fp16: |dh / q| <= 2^grid_bits
        dl = (xs - dh).half().float()
        tiles_h = F.pad(dh, (1, 1, 1, 1)).unfold(2, 4, 2)
        tiles_l = F.pad(dl, (1, 1, 1, 1)).unfold(2, 4, 2)
        Vh = torch.einsum('xj,crtj->xrct', BT, tiles_h)
        assert bool((Vh.half().float() == Vh).all())                                 # (the hi differences are fp16 numbers)
        Vl = torch.einsum('xj,crtj->xrct', BT.double(), tiles_l.double()).half().float()     # one fp16 rounding per packed addition
        M = []
        for xi in range(4):
            rows = []
            for r in range(H):
                terms = []
                for dy in range(3):
                    terms += [(Ul[dy, xi], Vh[xi, r + dy]), (Uh[dy, xi], Vl[xi, r + dy]), (Uh[dy, xi], Vh[xi, r + dy])]
                rows.append(chain_k16(terms))
            M.append(torch.stack(rows, 1))
        M = torch.stack(M, 0)
        y0 = (M[0] + M[1]) + M[2]
        y1 = (M[1] - M[2]) - M[3]
        y = torch.stack((y0, y1), -1).reshape(C, H, W) * pow2(-e - sw)
        if bias is not None:
            y = y + bias.view(-1, 1, 1)
        out[i] = torch.relu(y) if relu else y
    return out


def direct16_layer(x, w, bias, relu):
    """The split-fp16 direct convolution of csrc/conv_s16.hip, emulated the same way: two chains (hi.hi | cross), 144 / 288 steps."""
    n, C, H, W = x.shape
    sw = 13 - int(np.floor(np.log2(float(w.abs().max()))))
    wh, wl = split(w * pow2(sw))
    out = torch.empty_like(x)
    for i in range(n):
        e = act_exp(x[i].abs().max())
        dh, dl = split(x[i] * pow2(e))
        dph, dpl = F.pad(dh, (1, 1, 1, 1)), F.pad(dl, (1, 1, 1, 1))
        big, small = [], []
        for c in range(0, 64, 16):
            for dx in range(3):
                for dy in range(3):
                    ph = dph[c:c + 16, dy:dy + H, dx:dx + W].reshape(16, H * W)
                    pl = dpl[c:c + 16, dy:dy + H, dx:dx + W].reshape(16, H * W)
                    small += [(wl[:, c:c + 16, dy, dx], ph), (wh[:, c:c + 16, dy, dx], pl)]
                    big += [(wh[:, c:c + 16, dy, dx], ph)]
        y = (chain_k16(big) + chain_k16(small)).reshape(C, H, W) * pow2(-e - sw)
        if bias is not None:
            y = y + bias.view(-1, 1, 1)
        out[i] = torch.relu(y) if relu else y
    return out


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


def main():
    from deqsci_amd import checkpoint
    from deqsci_amd.networks import FFDNet
    torch.manual_seed(0)
    net = FFDNet(1, "ffdnet")
    net.load_state_dict(checkpoint.read_state_dict(checkpoint.shipped("ffdnet_gray"))[0])
    net.eval()
    seq = net.intermediate_dncnn.itermediate_dncnn
    crop = int(os.environ.get("WINO16_CROP", "48"))
    import scipy.io
    mat = scipy.io.loadmat(os.path.join(ROOT, "data", "test_gray", sorted(os.listdir(os.path.join(ROOT, "data", "test_gray")))[-1]))
    orig = torch.from_numpy(np.asarray(mat["orig"], dtype=np.float32) / 255.0)[:2 * crop, :2 * crop, :2].permute(2, 0, 1)[:, None]
    x = orig + 0.05 * torch.randn_like(orig)
    h = torch.relu(seq[0](net.concatenate_input_noise_map(x, torch.full((x.shape[0],), 50 / 255.0))))
    rows, li = [], 2
    with torch.no_grad():
        while li < len(seq) - 1:
            conv, bn = seq[li], seq[li + 1]
            s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            w = (conv.weight * s.view(-1, 1, 1, 1)).float()
            b = (bn.bias - bn.running_mean * s).float()
            ref = torch.relu(F.conv2d(h.double(), w.double(), b.double(), padding=1))
            row = {"layer": (li - 2) // 3 + 1,
                   "wino16 F(2,3)x nested (emulated)": rel(wino16_layer(h, w, b, True), ref),
                   "wino16, hi on a fixed 10-bit grid, packed fp16 transform (emulated)": rel(wino16_fixed_layer(h, w, b, True, 10), ref),
                   "... 9-bit grid": rel(wino16_fixed_layer(h, w, b, True, 9), ref),
                   "s16 direct two chains (emulated)": rel(direct16_layer(h, w, b, True), ref),
                   "fp32 conv2d (oneDNN)": rel(torch.relu(F.conv2d(h, w, b, padding=1)), ref)}
            rows.append(row)
            print(json.dumps(row), flush=True)
            h = ref.float()
            li += 3
    keys = [k for k in rows[0] if k != "layer"]
    print("SUMMARY", json.dumps({k: {"median": float(np.median([r[k] for r in rows])), "worst": max(r[k] for r in rows)} for k in keys}))


if __name__ == "__main__":
    main()
