#!/usr/bin/env python3
"""What comes after the three-product DIRECT convolution: the same split-fp16 arithmetic under a Winograd
transform - 2.25x (F(2x2,3x3)) or 4x (F(4x4,3x3)) fewer matrix-core flops per output.  Would it still be fp32-class?  Numerics only: an
EMULATION in torch of what such a kernel would compute, on FFDNet's own data, layer by layer, against a float64 convolution of the same
fp32 operands -

    U = G g G^T      in float64, rounded to fp32, scaled by a power of two per position so that max |U| sits in [2^13, 2^14), split into
                     hi + lo fp16 (as csrc/conv_s16.hip packs its weights)
    V = B^T d B      in fp32 from the fp32 activation, scaled by a power of two per image (2^e max |V| in [2^11, 2^12)), split into hi + lo
    M = U_hi V_hi + (U_lo V_hi + U_hi V_lo)      the three products of exactly representable fp16 values, accumulated in fp32 (here: fp32
                     GEMMs on the pieces cast back to fp32 - exact products, fp32 accumulation in rocBLAS's order instead of the MFMA's)
    Y = A^T M A      in fp32, + bias, ReLU

beside the kernels that exist (split-fp16 direct, fp32 Winograd F(2x2,3x3) / F(4x4,3x3), MIOpen).  One JSON line per (input, layer), a summary
at the end.  No timing - the emulation is slow."""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deqsci_amd  # noqa: E402
from deqsci_amd import _hip, checkpoint  # noqa: E402
from deqsci_amd.cli import build_pipeline  # noqa: E402
from deqsci_amd.engine import DEQSCIEngine, SIGMA0  # noqa: E402
from deqsci_amd.harness import SCITestDataset, as_clip  # noqa: E402

DATA = os.path.join(ROOT, "data", "test_gray")
FORMS = {
    "F(2x2,3x3)": dict(m=2, BT=[[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]],
                       G=[[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], AT=[[1, 1, 1, 0], [0, 1, -1, -1]]),
    "F(4x4,3x3)": dict(m=4, BT=[[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]],
                       G=[[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
                       AT=[[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]),
}


def split(t):
    """fp32 -> (hi, lo) as fp32 tensors holding exactly fp16-representable values (round to nearest even, as v_cvt_pk_f16_f32)."""
    hi = t.half().float()
    return hi, (t - hi).half().float()


def pow2_scale(amax, target):
    """2^e with 2^e amax in [2^target, 2^(target+1))."""
    return torch.exp2(target - torch.floor(torch.log2(amax.clamp_min(1e-30))))


def winograd_split16(x, w, bias, relu, form, split_operands=True):
    f = FORMS[form]
    m, a = f["m"], f["m"] + 2
    dev = x.device
    bt = torch.tensor(f["BT"], dtype=torch.float32, device=dev)
    at = torch.tensor(f["AT"], dtype=torch.float32, device=dev)
    g = torch.tensor(f["G"], dtype=torch.float64, device=dev)
    n, C, H, W = x.shape
    U = (g @ w.double() @ g.t()).float().permute(2, 3, 0, 1).reshape(a * a, 64, 64)          # [pos][cout][cin]
    xp = F.pad(x.contiguous(), (1, 1, 1, 1))
    d = xp.unfold(2, a, m).unfold(3, a, m)                                                      # (n, C, th, tw, a, a)
    th, tw = d.shape[2], d.shape[3]
    V = torch.einsum('ij,nctujk,lk->ilnctu', bt, d, bt).reshape(a * a, n, C, th * tw)          # fp32 input transform
    if split_operands:
        su = pow2_scale(U.abs().amax(dim=(1, 2), keepdim=True), 13)                              # per position
        sv = pow2_scale(V.abs().amax(dim=(0, 2, 3), keepdim=True), 11)                           # per image
        Uh, Ul = split(U * su)
        Vh, Vl = split(V * sv)
        M = torch.einsum('xoc,xnct->xnot', Uh, Vh) + (torch.einsum('xoc,xnct->xnot', Ul, Vh) + torch.einsum('xoc,xnct->xnot', Uh, Vl))
        M = M / (su.view(-1, 1, 1, 1) * sv.view(1, -1, 1, 1))                                  # exact: powers of two
    else:
        M = torch.einsum('xoc,xnct->xnot', U, V)
    M = M.reshape(a, a, n, 64, th, tw)
    Y = torch.einsum('ij,jknotu,lk->notiul', at, M, at).reshape(n, 64, th * m, tw * m)[:, :, :H, :W]
    if bias is not None:
        Y = Y + bias.view(1, -1, 1, 1)
    return torch.relu(Y) if relu else Y


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


def main():
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 180)[0].nonlinear_op
    eng = DEQSCIEngine(net, max_iter=30, use_graph=False)
    den = eng.den
    clip = [as_clip(c) for c in SCITestDataset(DATA)][-1]                # traffic
    Phi = clip["mask"].to("cuda")[None].contiguous()
    y = clip["meas"][..., 0].to("cuda")[None].contiguous()
    inputs = {"x0": deqsci_amd.initial_point(y, Phi, None, None), "iterate30": eng.reconstruct(y, Phi)}
    rows = []
    for name, z in inputs.items():
        x = z.permute(0, 3, 1, 2).reshape(8, 1, 256, 256).contiguous()
        sig = torch.full((1,), SIGMA0 * 0.971 ** (0 if name == "x0" else 30), device="cuda")
        h = _hip.ffdnet_head(x, den.head_w, sig)
        for li in range(1, len(den.fast) - 1):
            w, b, relu = den.fast[li]
            ref = F.conv2d(h.double(), w.double(), b.double(), padding=1)
            ref = torch.relu(ref) if relu else ref
            row = {"input": name, "layer": li}
            row["s16 direct (kernel)"] = rel(_hip.conv3x3_c64_split16(_hip.to_split16(h), _hip.Split16Weights(w), b, relu, out_f32=True), ref)
            row["fp32 F(2x2,3x3) (kernel)"] = rel(_hip.conv3x3_c64_winograd(h, den.wino[li].f22, b, relu), ref)
            row["fp32 F(4x4,3x3) (kernel)"] = rel(_hip.conv3x3_c64_winograd44(h, den.wino[li].f44, b, relu), ref)
            g = F.conv2d(h, w, b, padding=1)
            row["fp32 direct (MIOpen)"] = rel(torch.relu(g) if relu else g, ref)
            for form in FORMS:
                row[f"split-fp16 {form} (emulated)"] = rel(winograd_split16(h, w, b, relu, form), ref)
                row[f"fp32 {form} (emulated)"] = rel(winograd_split16(h, w, b, relu, form, split_operands=False), ref)
            rows.append(row)
            print(json.dumps({k: (round(v, 10) if isinstance(v, float) else v) for k, v in row.items()}), flush=True)
            h = _hip.conv3x3_c64_winograd(h, den.wino[li].f22, b, relu)
    keys = [k for k in rows[0] if k not in ("input", "layer")]
    summary = {k: {"median": float(torch.tensor([r[k] for r in rows]).median()), "worst": max(r[k] for r in rows)} for k in keys}
    print("SUMMARY", json.dumps(summary, indent=1))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"what": __doc__.split("\n\n")[0], "rows": rows, "summary": summary}, open(os.path.join(ROOT, "gpurun_out", "winograd_split16_numerics.json"), "w"), indent=1)


if __name__ == "__main__":
    with torch.no_grad():
        main()
