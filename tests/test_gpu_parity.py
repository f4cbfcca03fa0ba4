"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs, against the committed golden vectors generated from the reference, and - at full size -
through size-independent properties.  Tolerances: the path is fp32 floating point; north_star
asks <= 1e-4 relative L2 / 0.01 dB end to end; single kernels are held to ~1 ulp-level bounds."""
import json
import math
import os
import time

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import deqsci_amd
    from deqsci_amd import _hip, checkpoint
    from deqsci_amd.cli import build_pipeline
    from deqsci_amd.engine import DEQSCIEngine
    from oracle import deqsci_oracle as orc

DEV = "cuda"
HWB, BHW = 0, 1


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def make_case(bsz, H, W, B, seed, shared=False, binary=True):
    g = torch.Generator().manual_seed(seed)
    nb = 1 if shared else bsz
    Phi = (torch.rand(nb, H, W, B, generator=g) < 0.5).float() if binary else torch.rand(nb, H, W, B, generator=g)
    Phi[:, 0, :2, :] = 0
    x = torch.rand(bsz, H, W, B, generator=g)
    z = torch.randn(bsz, H, W, B, generator=g)
    Phie = Phi.expand(bsz, H, W, B)
    y = orc.sci_forward(x, Phie)
    return Phi, Phie, x, z, y, orc.phi_sum(Phi)


SHAPES = [(2, 16, 16, 8), (1, 32, 32, 16), (3, 8, 12, 4), (2, 5, 7, 5), (1, 6, 6, 32), (2, 9, 4, 12), (1, 64, 64, 8)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("shared", [False, True])
def test_ops_vs_oracle_all_layouts(shape, shared):
    bsz, H, W, B = shape
    Phi, Phie, x, z, y, Ps = make_case(bsz, H, W, B, seed=sum(shape), shared=shared)
    Pse = Ps.expand(bsz, H, W)
    want_Az, want_Aty, want_z1 = orc.sci_forward(z, Phie), orc.sci_adjoint(y, Phie), orc.gap_update(z, y, Phie, Pse)
    dPhi, dz, dy, dPs = G(Phi), G(z), G(y), G(Ps)
    planar = lambda t: t.permute(0, 3, 1, 2).contiguous()
    # HWB
    assert torch.equal(_hip.phi_sum(dPhi, HWB).cpu(), Ps)
    assert torch.equal(_hip.sci_adjoint(dy, dPhi, HWB).cpu(), want_Aty)
    torch.testing.assert_close(_hip.sci_forward(dz, dPhi, HWB).cpu(), want_Az, rtol=1e-6, atol=2e-6)
    torch.testing.assert_close(_hip.gap_update(dz, dPhi, dy, dPs, HWB, HWB).cpu(), want_z1, rtol=1e-5, atol=1e-5)
    # BHW
    pPhi, pz = planar(dPhi), planar(dz)
    assert torch.equal(_hip.phi_sum(pPhi, BHW).cpu(), Ps)
    assert torch.equal(_hip.sci_adjoint(dy, pPhi, BHW).cpu(), planar(want_Aty))
    torch.testing.assert_close(_hip.sci_forward(pz, pPhi, BHW).cpu(), want_Az, rtol=1e-6, atol=2e-6)
    torch.testing.assert_close(_hip.gap_update(pz, pPhi, dy, dPs, BHW, BHW).cpu(), planar(want_z1), rtol=1e-5, atol=1e-5)
    # HWB -> BHW fused transpose, and the transposes themselves
    torch.testing.assert_close(_hip.gap_update(dz, dPhi, dy, dPs, HWB, BHW).cpu(), planar(want_z1), rtol=1e-5, atol=1e-5)
    assert torch.equal(_hip.transpose(dz, BHW), pz)
    assert torch.equal(_hip.transpose(pz, HWB), dz)
    # in place
    zz = dz.clone()
    _hip.gap_update(zz, dPhi, dy, dPs, HWB, HWB, out=zz)
    torch.testing.assert_close(zz.cpu(), want_z1, rtol=1e-5, atol=1e-5)
    # residual_out
    n = torch.randn_like(pz)
    assert torch.equal(_hip.residual_out(pz, n, HWB), (pz - n).permute(0, 2, 3, 1).contiguous())
    assert torch.equal(_hip.residual_out(pz, n, BHW), pz - n)


def test_ops_reference_golden():
    g = np.load(os.path.join(GOLDEN, "ops.npz"))
    for c in ("c0_", "c1_", "c2_"):
        Phi, z, y, x = G(g[c + "Phi"]), G(g[c + "z"]), G(g[c + "y"]), G(g[c + "x"])
        Ps = deqsci_amd.phi_sum(Phi)
        assert np.array_equal(Ps.cpu().numpy(), g[c + "Phi_sum"])
        assert np.array_equal(deqsci_amd.At_torch_(y, Phi).cpu().numpy(), g[c + "Aty"])
        assert np.array_equal(deqsci_amd.initial_point(y, Phi, Ps, None).cpu().numpy(), g[c + "x0"])
        np.testing.assert_allclose(deqsci_amd.A_torch_(x, Phi).cpu().numpy(), g[c + "y"], rtol=1e-6, atol=2e-6)
        np.testing.assert_allclose(deqsci_amd.A_torch_(z, Phi).cpu().numpy(), g[c + "Az"], rtol=1e-6, atol=2e-6)
        from deqsci_amd.operators import gap_update
        np.testing.assert_allclose(gap_update(z, y, Phi, Ps).cpu().numpy(), g[c + "z1"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(gap_update(G(g["grey_z"]), G(g["grey_y"]), G(g["grey_Phi"]), G(g["grey_Phi_sum"])).cpu().numpy(),
                               g["grey_z1"], rtol=1e-5, atol=1e-5)


def test_error_conventions():
    with pytest.raises(_hip.DeqsciHipError):
        deqsci_amd.A_torch_(torch.zeros(1, 4, 4, 8), torch.zeros(1, 4, 4, 8))           # CPU tensors: no fallback
    with pytest.raises(_hip.DeqsciHipError):
        deqsci_amd.A_torch_(torch.zeros(1, 4, 4, 8, device=DEV), torch.zeros(1, 4, 5, 8, device=DEV))
    with pytest.raises(_hip.DeqsciHipError):
        _hip.AndersonWorkspace(1, 64, 9, DEV)                                          # m > DEQSCI_MAX_M
    base = torch.zeros(4 * 4 * 8 + 1, device=DEV)
    mis = base[1:].view(1, 4, 4, 8)
    lib = _hip.load()
    code = lib.deqsci_sci_forward_f32(mis.data_ptr(), mis.data_ptr(), base.data_ptr(), 1, 4, 4, 8, 0, 0, None)
    assert code == -3
    assert lib.deqsci_sci_forward_f32(None, mis.data_ptr(), base.data_ptr(), 1, 4, 4, 8, 0, 0, None) == -1
    assert lib.deqsci_sci_forward_f32(base.data_ptr(), base.data_ptr(), base.data_ptr(), 0, 4, 4, 8, 0, 0, None) == -2
    assert lib.deqsci_sci_forward_f32(base.data_ptr(), base.data_ptr(), base.data_ptr(), 1, 4, 4, 8, 7, 0, None) == -4


def test_adjointness_and_projection_full_size():
    """Size-independent properties at BASELINE sizes (256x256x8 batch 8, 512x512x16):
    <Phi x, y> = <x, Phi^T y>, and one GAP step lands on the data: Phi z1 = y where Phi_sum != 0."""
    for (bsz, H, W, B) in [(8, 256, 256, 8), (2, 512, 512, 16)]:
        g = torch.Generator(device=DEV).manual_seed(1234)
        Phi = (torch.rand(bsz, H, W, B, device=DEV, generator=g) < 0.5).float()
        x = torch.rand(bsz, H, W, B, device=DEV, generator=g)
        yv = torch.randn(bsz, H, W, device=DEV, generator=g)
        for layout in (HWB, BHW):
            P_, x_ = (Phi, x) if layout == HWB else (Phi.permute(0, 3, 1, 2).contiguous(), x.permute(0, 3, 1, 2).contiguous())
            lhs = (_hip.sci_forward(x_, P_, layout).double() * yv.double()).sum()
            rhs = (x_.double() * _hip.sci_adjoint(yv, P_, layout).double()).sum()
            assert abs(lhs - rhs) / abs(lhs) < 1e-6
            Ps = _hip.phi_sum(P_, layout)
            y = _hip.sci_forward(x_, P_, layout)
            z1 = _hip.gap_update(torch.randn_like(x_), P_, y, Ps, layout, layout)
            err = (_hip.sci_forward(z1, P_, layout) - y).abs()
            assert err[Ps_nonzero(P_, layout)].max() < 2e-5
            z2 = _hip.gap_update(z1, P_, y, Ps, layout, layout)          # projection is idempotent
            assert (z2 - z1).abs().max() < 2e-5


def Ps_nonzero(P_, layout):
    return (P_.sum(3 if layout == HWB else 1) != 0)


# ----------------------------------------------------------------------------- Anderson / Picard (generic f)
def _toy(a, c):
    return lambda z: a * z + 0.3 * torch.sin(z) + c


@pytest.mark.parametrize("bsz", [1, 3])
def test_anderson_generic_vs_reference_golden(bsz):
    g = np.load(os.path.join(GOLDEN, "anderson_toy.npz"))
    a, c, x0 = G(g[f"b{bsz}_a"]), G(g[f"b{bsz}_c"]), G(g[f"b{bsz}_x0"])
    for it in (3, 7, 12, 40):
        for aa in ("reference", "float64"):                  # (the default: the reference's fp32 Gram + fp32 LU; the exactly accumulated Gram)
            z, res = deqsci_amd.andersonexp(_toy(a, c), x0, m=5, lam=1e-2, max_iter=it, tol=1e-5, beta=1.0, anderson_arith=aa)
            assert rel_l2(z.cpu().numpy(), g[f"b{bsz}_it{it}_z"]) < 2e-5, (it, aa)
            assert abs(res - float(g[f"b{bsz}_it{it}_res"])) <= 2e-3 * float(g[f"b{bsz}_it{it}_res"]) + 1e-7
    with pytest.raises(ValueError):
        deqsci_amd.andersonexp(_toy(a, c), x0, anderson_arith="fp16")
    n = [0]

    def f2(z):
        n[0] += 1
        return _toy(a, c)(z)
    z, res = deqsci_amd.andersonexp(f2, x0, m=5, lam=1e-2, max_iter=40, tol=1e-3, beta=1.0)
    assert n[0] == int(g[f"b{bsz}_early_ncalls"])
    assert rel_l2(z.cpu().numpy(), g[f"b{bsz}_early_z"]) < 2e-5
    z, res = deqsci_amd.andersonexp(_toy(a, c), x0, m=3, lam=1e-3, max_iter=9, tol=1e-5, beta=0.7)
    assert rel_l2(z.cpu().numpy(), g[f"b{bsz}_m3beta_z"]) < 2e-5
    z, res = deqsci_amd.forward_iteration(_toy(a, c), x0, max_iter=15, tol=1e-5)
    assert rel_l2(z.cpu().numpy(), g[f"b{bsz}_picard_z"]) < 1e-6
    np.testing.assert_allclose(np.array(res), g[f"b{bsz}_picard_res"], rtol=1e-4)
    z, res = deqsci_amd.forward_iteration(_toy(a, c), x0, max_iter=60, tol=1e-3)
    assert len(res) == len(g[f"b{bsz}_picard_early_res"])
    with pytest.raises(UnboundLocalError):
        deqsci_amd.andersonexp(_toy(a, c), x0, m=5, lam=1e-2, max_iter=2)
    with pytest.raises(IndexError):
        deqsci_amd.andersonexp(_toy(a, c), x0, m=1, lam=1e-2, max_iter=5)


def test_anderson_kernels_vs_oracle_odd_sizes():
    """N not a multiple of 4 (scalar paths) and m = 8 (largest supported history)."""
    g = torch.Generator().manual_seed(3)
    a, c, x0 = torch.rand(2, 3, 7, 5, generator=g) * 0.5, torch.randn(2, 3, 7, 5, generator=g), torch.randn(2, 3, 7, 5, generator=g)
    for m, beta in ((8, 1.0), (4, 0.5)):
        want, wres = orc.andersonexp(_toy(a, c), x0, m=m, lam=1e-2, max_iter=14, tol=1e-9, beta=beta)
        got, gres = deqsci_amd.andersonexp(_toy(G(a), G(c)), G(x0), m=m, lam=1e-2, max_iter=14, tol=1e-9, beta=beta)
        assert rel_l2(got.cpu().numpy(), want.numpy()) < 5e-5
        assert abs(gres - wres) < 5e-2 * wres + 1e-7


def _history(ws, rows):
    """rows (bsz, m, N) -> the workspace's residual history G (x = 0: G = f), slot by slot as the loop stores it; returns the last slot."""
    bsz, m, N = rows.shape
    zero = torch.zeros(bsz, N, device=DEV)
    for k in range(m):
        _hip.residual_store(ws, rows[:, k].contiguous(), None, zero, k, k + 1, None)
    return m - 1


def _gram_rows(kind, bsz, m, N, gen):
    r = lambda *sh: torch.randn(*sh, device=DEV, generator=gen)
    if kind == "correlated":                                   # what the loop's history looks like late in the iteration
        return (r(bsz, 1, N) * (1 + 0.05 * torch.arange(m, device=DEV).view(1, m, 1)) + 0.3 * r(bsz, m, N)) * 1e-2
    if kind == "heavy":                                        # ... with its heavy tails: most products are absorbed by the running sum
        return (r(bsz, 1, N) ** 3 + 0.3 * r(bsz, m, N) ** 3) * 1e-3
    if kind == "orthogonal":                                   # off-diagonal sums wander around zero: nearly every block is walked
        return r(bsz, m, N)
    if kind == "negative":                                     # negative running sums
        sign = torch.tensor([1.0, -1.0] * m, device=DEV)[:m].view(1, m, 1)
        return sign * (r(bsz, 1, N) + 0.2 * r(bsz, m, N))
    if kind == "tiny":
        return (r(bsz, 1, N) + 0.3 * r(bsz, m, N)) * 1e-17
    if kind == "huge":
        return (r(bsz, 1, N) + 0.3 * r(bsz, m, N)) * 1e14
    if kind == "few_bits":                                     # small integers / 1024: every product lands on a grid point or a TIE
        return torch.randint(-40, 41, (bsz, m, N), device=DEV, generator=gen).float() / 1024
    if kind == "zero_row":
        x = r(bsz, m, N)
        x[:, m - 1] = 0
        return x
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["correlated", "heavy", "orthogonal", "negative", "tiny", "huge", "few_bits", "zero_row"])
@pytest.mark.parametrize("bsz,N,m", [(1, 1 << 19, 5), (3, 1 << 17, 5), (2, 20480, 3), (1, (1 << 19) + 28, 8), (2, 105, 5), (1, 2048, 2), (1, 1 << 22, 3), (9, 1 << 15, 5)])
def test_reference_gram_two_pass_form_is_bit_equal_to_the_serial_chains(kind, bsz, N, m):
    """anderson_arith = "reference": the 16 FMA chains per Gram entry (csrc/anderson.hip) as they are written - N / 16 dependent FMAs,
    gram_row_chain16_kernel - and in the two-pass form that ships (gram_round_kernel + gram_chain_apply_kernel: inside a binade of the running sum
    a chain step is S + RN_ulp(p), an integer sum in any order; crossings, ties and oversized terms are walked) give THE SAME BITS, on every kind
    of history: the loop's (correlated, heavy-tailed), sums that wander around zero or run negative, scales near the ends of fp32, operands with
    few bits (ties), an all-zero row; ragged N, N % 4 != 0 (both forms fall back to the serial kernel), m = 8, more blocks than the record window of
    the apply kernel (N = 2^22: 2048 blocks, eight windows), an odd batch."""
    gen = torch.Generator(device=DEV).manual_seed(11)
    rows = _gram_rows(kind, bsz, m, N, gen).float().contiguous()
    ws = _hip.AndersonWorkspace(bsz, N, m, DEV)
    slot = _history(ws, rows)
    two = _hip.gram_row_chain16(ws, slot, m, serial=False)[:, :m].clone()
    walked = ws.chains_walked()[:, :m].clone()
    ser = _hip.gram_row_chain16(ws, slot, m, serial=True)[:, :m].clone()
    assert torch.isfinite(ser).all()
    assert torch.equal(two, ser), (kind, int((two != ser).sum()))
    if N % 4 == 0:
        assert (walked >= 0).all()                             # (the two-pass form did run)
        if kind in ("correlated", "heavy") and N >= 1 << 17:
            assert walked.float().mean() < 0.25 * (N // 2048)  # ... and passed most blocks without walking them
    # a second call on another slot of the same workspace (records, term slots and counters are reused)
    two2 = _hip.gram_row_chain16(ws, 0, m, serial=False)[:, :m].clone()
    ser2 = _hip.gram_row_chain16(ws, 0, m, serial=True)[:, :m].clone()
    assert torch.equal(two2, ser2)
    # round 6: the first pass fused into K4 (deqsci_residual_store_ref_f32: the blocks tell one another where the chains stand, a ticket
    # orders them) - the same F / G / block sums as K4 alone, and the same chain sums again, slot after slot through the loop's own calls
    wf = _hip.AndersonWorkspace(bsz, N, m, DEV)
    assert wf.ref_fusable == (N % 2048 == 0 and N // 2048 <= 256)
    zero = torch.zeros(bsz, N, device=DEV)
    for k in range(m):
        _hip.residual_store(wf, rows[:, k].contiguous(), None, zero, k, k + 1, None, ref=True)
        assert (wf._rounded == (k, k + 1)) == wf.ref_fusable
        _hip.anderson_solve(wf, k, k + 1, k + 1 if k else 0, 1e-2, 1e-5, ref=True)
        assert wf._rounded is None
    assert torch.equal(wf.G, ws.G) and torch.equal(wf.F, ws.F)
    assert torch.equal(wf.chain_sums()[:, :m], ser), (kind, int((wf.chain_sums()[:, :m] != ser).sum()))
    if wf.ref_fusable:
        wk = wf.chains_walked()[:, :m]
        assert (wk >= 0).all()
        if kind in ("correlated", "heavy") and N >= 1 << 17:
            assert wk.float().mean() < 0.25 * (N // 2048)      # (the look-back's predictions are as good as the separate pass's)
    # ... and once more into slot 0 of the same state (the ticket counter runs on: call number 2 of that state)
    _hip.residual_store(wf, rows[:, 0].contiguous(), None, zero, 0, m, None, ref=True)
    _hip.anderson_solve(wf, 0, m, m, 1e-2, 1e-5, ref=True)
    assert torch.equal(wf.chain_sums()[:, :m], ser2)


@pytest.mark.parametrize("fused", [False, True])
def test_reference_gram_kernels_vs_oracle_and_the_references_bmm(fused):
    """The chain sums of the kernels against the CPU restatement of the reference's summation order (oracle.gram_chain16: bit-equal), and the
    folded Gram entries the engine solves with against what torch.bmm itself returned for the same rows on the CPU behind tests/golden
    (gram_bmm_cpu.npz: within one ulp, and the same bias of the diagonal).  fused: the first pass inside K4 (what the engine runs)."""
    g = np.load(os.path.join(GOLDEN, "gram_bmm_cpu.npz"))
    rows = orc.heavy_tailed_rows(int(g["seed"]), int(g["n"]), int(g["N"]))
    want, chains = orc.gram_chain16(rows)
    m, N = rows.shape
    ws = _hip.AndersonWorkspace(1, N, m, DEV)
    R = G(torch.from_numpy(rows))[None]
    zero = torch.zeros(1, N, device=DEV)
    for k in range(m):                                         # every slot's row / column, as the loop fills them
        _hip.residual_store(ws, R[:, k].contiguous(), None, zero, k, k + 1, None, ref=fused)
        assert (ws._rounded is not None) == fused
        _hip.anderson_solve(ws, k, k + 1, k + 1 if k else 0, 1e-2, 1e-5, ref=True)
        assert np.array_equal(ws.chain_sums()[0, :k + 1].cpu().numpy(), chains[k, :k + 1]), k
    got = ws.gram32_state()[0, :m, :m].cpu().numpy()
    assert np.array_equal(got, want)
    assert (np.abs(got.astype(np.float64) - g["bmm"]) <= np.spacing(np.abs(g["bmm"]))).all()
    err = np.diag((got.astype(np.float64) - g["exact"]) / g["exact"])
    assert (err < -1e-5).all()


# ----------------------------------------------------------------------------- f-map and the DEQ loop
def _pipeline(kind, iters, iterator="anderson"):
    solver, deq = build_pipeline(kind, checkpoint.shipped("ffdnet_gray" if kind == "ffdnet" else "cnn"), iters)
    if iterator == "picard":
        deq = deqsci_amd.DEQFixedPoint(solver, deqsci_amd.forward_iteration, max_iter=iters, tol=1e-5)
    return solver, deq


@pytest.mark.parametrize("kind", ["SimpleCNN", "ffdnet"])
def test_teacher_forced_f_vs_reference_trace(kind):
    """Every one of the reference's 12 f-calls (64x64 crop, and_maxiters=10): same input -> same output."""
    g = np.load(os.path.join(GOLDEN, f"trace_{kind}.npz"))
    solver, _ = _pipeline(kind, 10)
    Phi, y, Ps = G(g["Phi"]), G(g["y"]), G(g["Phi_sum"])
    with torch.no_grad():
        for i in range(12):
            out = solver(G(g["fed"][i]), y, Phi, Ps)
            assert rel_l2(out.cpu().numpy(), g["ret"][i]) < 1e-5, i
            if kind == "ffdnet":
                assert np.array_equal(solver.noise_sigma.cpu().numpy(), g["sigma"][i])


@pytest.mark.parametrize("kind", ["SimpleCNN", "ffdnet"])
@pytest.mark.parametrize("use_engine", [True, False])
def test_deq_loop_vs_reference_trace(kind, use_engine):
    g = np.load(os.path.join(GOLDEN, f"trace_{kind}.npz"))
    _, deq = _pipeline(kind, 10)
    deq.use_engine = use_engine
    Phi, y, Ps, x0 = G(g["Phi"]), G(g["y"]), G(g["Phi_sum"]), G(g["x0"])
    rec = deq.forward(y, Phi, Ps, initial_point=x0, train_flag=False)
    assert rel_l2(rec.cpu().numpy(), g["rec"]) < 1e-4
    assert abs(deq.forward_res - float(g["res"])) < 1e-2 * float(g["res"])


def _golden_meta(tag):
    with open(os.path.join(GOLDEN, f"e2e_{tag}.json")) as fh:
        return json.load(fh)


def _clip(name):
    d = orc.load_clip(os.path.join(orc.DATA_DIR, name))
    return {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in d.items()}


E2E = [("SimpleCNN", "anderson", 180, "traffic_cacti.mat", 0, "traffic_m0", 1e-4),
       ("SimpleCNN", "anderson", 180, "drop8_cacti.mat", 0, "drop8_m0", 1e-4),
       ("SimpleCNN", "anderson", 180, "runner8_cacti.mat", 0, "runner8_m0", 1e-4),
       ("SimpleCNN", "anderson", 10, "traffic_cacti.mat", 3, "traffic_m3", 1e-4),
       ("ffdnet", "anderson", 10, "traffic_cacti.mat", 0, "traffic_m0", 1e-4),
       ("ffdnet", "anderson", 10, "drop8_cacti.mat", 0, "drop8_m0", 1e-4),
       ("ffdnet", "anderson", 30, "traffic_cacti.mat", 0, "traffic_m0", 1e-4),
       ("ffdnet", "picard", 180, "traffic_cacti.mat", 0, "traffic_m0", 1e-4)]


@pytest.mark.parametrize("kind,iterator,iters,clip,fi,key,tol", E2E)
def test_end_to_end_vs_reference_rec(kind, iterator, iters, clip, fi, key, tol):
    """Full 256x256x8 reconstructions against the reference's own output tensors: <= 1e-4 relative L2
    and <= 0.01 dB (north_star).  FFDNet+Anderson is only gated up to 30 iterations because the
    reference itself moves by 4e-2 under a 1-ulp input change at 180; the 180-iteration
    FFDNet pin is the Picard run."""
    tag = f"{kind}_{iterator}_{iters}" + ("_first" if iterator == "picard" else "")
    recs = np.load(os.path.join(GOLDEN, f"e2e_{tag}_rec.npz"))
    meta = [m for m in _golden_meta(tag)["measurements"] if m["id"] == f"{clip}:{fi}"][0]
    d = _clip(clip)
    Phi = d["mask"][None].to(DEV)
    y = d["meas"][None, ..., fi].contiguous().to(DEV)
    gt = d["gt"][None, ..., fi * 8:(fi + 1) * 8]
    _, deq = _pipeline(kind, iters, iterator)
    Ps = deqsci_amd.phi_sum(Phi)
    rec = deq.forward(y, Phi, Ps, initial_point=deqsci_amd.initial_point(y, Phi, Ps, None), train_flag=False).cpu().numpy()
    assert rel_l2(rec, recs[key]) < tol
    assert abs(orc.psnr(rec, gt.numpy()) - meta["psnr"]) < 0.01
    res = deq.forward_res[-1] if isinstance(deq.forward_res, list) else deq.forward_res
    assert abs(res - meta["res"]) < 2e-2 * meta["res"]


@pytest.mark.parametrize("kind,iters", [("SimpleCNN", 10), ("ffdnet", 10)])
def test_harness_all_clips_vs_reference(kind, iters):
    """The build's own test_solver_sci over data/test_gray against the reference harness run:
    measurement order, slicing rules, per-measurement PSNR, clip means, grand mean, PNG payload count."""
    from deqsci_amd.harness import SCITestDataset, test_solver_sci
    meta = _golden_meta(f"{kind}_anderson_{iters}")
    _, deq = _pipeline(kind, iters)
    loader = torch.utils.data.DataLoader(dataset=SCITestDataset(orc.DATA_DIR), batch_size=1, shuffle=False, drop_last=True)
    records = []
    avg, images = test_solver_sci(deq, test_dataloader=loader, save_img_path="", verbose=False, save_image=False, records=records)
    assert [r["id"] for r in records] == [m["id"] for m in meta["measurements"]]
    for r, m in zip(records, meta["measurements"]):
        assert abs(r["psnr"] - m["psnr"]) < 0.01, r["id"]
    assert abs(avg - meta["avg_psnr"]) < 0.01
    assert len(images) == meta["n_png_payloads"] == 64


def test_harness_clip_batched_equals_sequential():
    from deqsci_amd.harness import SCITestDataset, test_solver_sci
    _, deq = _pipeline("SimpleCNN", 10)
    loader = torch.utils.data.DataLoader(dataset=SCITestDataset(orc.DATA_DIR), batch_size=1, shuffle=False, drop_last=True)
    r1, r2 = [], []
    a1, _ = test_solver_sci(deq, test_dataloader=loader, save_img_path="", verbose=False, save_image=False, records=r1,
                            batch_measurements=False)
    a2, im = test_solver_sci(deq, test_dataloader=loader, save_img_path="", verbose=False, save_image=False, records=r2,
                             batch_measurements=True)
    assert len(im) == 64 and abs(a1 - a2) < 1e-3
    for x, y_ in zip(r1, r2):
        assert x["id"] == y_["id"] and rel_l2(y_["rec"].numpy(), x["rec"].numpy()) < 2e-5
    # round 6: the measurements of ALL clips of one frame size as one engine batch, each with its own clip's mask (the three shipped clips:
    # one call of eight) - same order, same PSNRs, same PNG payloads
    r3 = []
    a3, im3 = test_solver_sci(deq, test_dataloader=loader, save_img_path="", verbose=False, save_image=False, records=r3, batch_measurements="all")
    assert len(im3) == 64 and sorted(im3) == sorted(im) and abs(a1 - a3) < 1e-3 and [x["id"] for x in r3] == [x["id"] for x in r1]
    for x, y_ in zip(r1, r3):
        assert rel_l2(y_["rec"].numpy(), x["rec"].numpy()) < 2e-5 and abs(x["psnr"] - y_["psnr"]) < 1e-3
    # ... and for FFDNet's stack path, where a measurement's result does not depend on its batch at all: bit-identical to the clip batches
    _, dq = _pipeline("ffdnet", 6)
    from deqsci_amd.harness import evaluate
    clips = list(SCITestDataset(orc.DATA_DIR))
    _, by_clip = evaluate(dq, clips, batch=True)
    _, together = evaluate(dq, clips, batch="all")
    assert [r.name for r in together] == [r.name for r in by_clip]
    for p_, q_ in zip(by_clip, together):
        assert torch.equal(p_.rec, q_.rec) and p_.psnr == q_.psnr and q_.info["batched"] == "all" and p_.frames == q_.frames


def test_engine_batch_equals_single_and_shared_mask():
    """Batched measurements are independent problems: bsz=3 (shared mask) == three bsz=1 runs."""
    d = _clip("traffic_cacti.mat")
    Phi = d["mask"][None].to(DEV)
    ys = d["meas"].permute(2, 0, 1)[:3].contiguous().to(DEV)
    net = build_pipeline("SimpleCNN", checkpoint.shipped("cnn"), 12)[0].nonlinear_op
    eng = DEQSCIEngine(net, max_iter=12)
    batched = eng.reconstruct(ys, Phi)
    assert eng.last_info["f_calls"] == 13
    per = eng.reconstruct(ys, Phi.expand(3, -1, -1, -1).contiguous())
    assert rel_l2(per.cpu().numpy(), batched.cpu().numpy()) < 1e-6
    for i in range(3):
        one = eng.reconstruct(ys[i:i + 1], Phi)
        assert rel_l2(one.cpu().numpy(), batched[i:i + 1].cpu().numpy()) < 2e-5


class _Contract(torch.nn.Module):
    tag = "conv2d"

    def forward(self, x):
        return 0.25 * x + 0.1


def test_engine_early_stop_matches_oracle_semantics():
    """tol fires: the engine must return f(X_k) for the same k as the reference algorithm, although it
    polls the residual one iteration late."""
    Phi, Phie, x, z, y, Ps = make_case(1, 16, 16, 8, seed=11)
    calls = [0]

    def f_cpu(zz):
        calls[0] += 1
        z1 = orc.gap_update(zz, y, Phie, Ps)
        return 0.25 * z1 + 0.1
    x0 = orc.initial_point(y, Phie)
    for iterator, fn, kw in (("anderson", orc.andersonexp, dict(m=5, lam=1e-2, beta=1.0)), ("picard", orc.forward_iteration, {})):
        calls[0] = 0
        zs, res = fn(f_cpu, x0, max_iter=50, tol=1e-4, **kw)
        n_loop = calls[0]
        want = f_cpu(zs)
        eng = DEQSCIEngine(_Contract(), iterator=iterator, max_iter=50, tol=1e-4)
        got = eng.reconstruct(G(y), G(Phi))
        assert n_loop < 40
        assert eng.last_info["f_calls"] == n_loop + 1, (iterator, eng.last_info, n_loop)
        assert rel_l2(got.cpu().numpy(), want.numpy()) < 1e-5


# ----------------------------------------------------------------------------- denoiser epilogue + engine variants
@pytest.mark.parametrize("shape", [(3, 64, 8, 12), (2, 4, 6, 6), (5, 8, 16, 4)])
def test_bias_relu_epilogue_vs_torch(shape):
    g = torch.Generator(device=DEV).manual_seed(7)
    h = torch.randn(shape, device=DEV, generator=g)
    b = torch.randn(shape[1], device=DEV, generator=g)
    want = torch.relu(h + b.view(1, -1, 1, 1))
    assert torch.equal(_hip.bias_relu_(h.clone(), b), want)
    assert torch.equal(_hip.bias_relu_(h.clone(), b, relu=False), h + b.view(1, -1, 1, 1))
    hc = h.clone().contiguous(memory_format=torch.channels_last)
    out = _hip.bias_relu_(hc, b)
    assert out.is_contiguous(memory_format=torch.channels_last) and torch.equal(out, want)


@pytest.mark.parametrize("channels_last,fused,edges", [(False, True, True), (True, True, True), (True, True, False),
                                                       (True, False, True), (False, False, False)])
def test_engine_ffdnet_variants_vs_reference(channels_last, fused, edges):
    """Every denoiser fast-path variant of the engine holds the same end-to-end gates:
    FFDNet Anderson@30 and the 12-call trace, <= 1e-4."""
    recs = np.load(os.path.join(GOLDEN, "e2e_ffdnet_anderson_30_rec.npz"))
    d = _clip("traffic_cacti.mat")
    Phi = d["mask"][None].to(DEV)
    y = d["meas"][None, ..., 0].contiguous().to(DEV)
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 30)[0].nonlinear_op
    eng = DEQSCIEngine(net, max_iter=30, channels_last=channels_last, fused_epilogue=fused, fused_edges=edges)
    rec = eng.reconstruct(y, Phi).cpu().numpy()
    assert rel_l2(rec, recs["traffic_m0"]) < 1e-4
    g = np.load(os.path.join(GOLDEN, "trace_ffdnet.npz"))
    eng10 = DEQSCIEngine(net, max_iter=10, channels_last=channels_last, fused_epilogue=fused, fused_edges=edges)
    rec = eng10.reconstruct(G(g["y"]), G(g["Phi"]), G(g["Phi_sum"]), initial_point=G(g["x0"]))
    assert rel_l2(rec.cpu().numpy(), g["rec"]) < 1e-4


def test_engine_ffdnet_picard_180_channels_last():
    recs = np.load(os.path.join(GOLDEN, "e2e_ffdnet_picard_180_first_rec.npz"))
    d = _clip("traffic_cacti.mat")
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 180)[0].nonlinear_op
    eng = DEQSCIEngine(net, iterator="picard", max_iter=180, channels_last=True)
    rec = eng.reconstruct(d["meas"][None, ..., 0].contiguous().to(DEV), d["mask"][None].to(DEV)).cpu().numpy()
    assert rel_l2(rec, recs["traffic_m0"]) < 1e-4


def test_engine_512x512x16_vs_oracle():
    """BASELINE config 4 shape (compression ratio 16, B=16 kernel instantiations) against the CPU oracle:
    synthetic clip, SimpleCNN (cnn.ckpt), and_maxiters=6.  At N = 4.2M the reference's own fp32 bmm Gram is
    only good to ~2.6e-4 (see oracle.andersonexp docstring), so the tight gate is against the oracle with an
    exactly accumulated Gram and the reference-exact oracle is held to the looser bound its own rounding allows."""
    g = torch.Generator().manual_seed(4)
    H = W = 512
    B = 16
    Phi = (torch.rand(1, H, W, B, generator=g) < 0.5).float()
    x = torch.rand(1, H, W, B, generator=g)
    y = orc.sci_forward(x, Phi)
    Ps = orc.phi_sum(Phi)
    kw = dict(m=5, beta=1.0, lam=1e-2, max_iter=6, tol=1e-5)
    want32, _ = orc.deq_forward(orc.ProxGradSCI("SimpleCNN"), orc.andersonexp, y, Phi, Ps, orc.initial_point(y, Phi), **kw)
    want64, wres = orc.deq_forward(orc.ProxGradSCI("SimpleCNN"), orc.andersonexp, y, Phi, Ps, orc.initial_point(y, Phi),
                                   gram_dtype=torch.float64, **kw)
    net = build_pipeline("SimpleCNN", checkpoint.shipped("cnn"), 6)[0].nonlinear_op
    for cl in (True, False):
        eng = DEQSCIEngine(net, max_iter=6, channels_last=cl, anderson_arith="float64")
        got = eng.reconstruct(G(y), G(Phi)).cpu().numpy()
        assert rel_l2(got, want64.numpy()) < 2e-5
        assert rel_l2(got, want32.numpy()) < 1e-3
        assert abs(eng.last_info["res"] - wres) < 2e-3 * wres and eng.last_info["f_calls"] == 7
    # the default arithmetic - the reference's fp32 Gram in the summation order of the BUILD HOST's torch.bmm (chains of 2^18 steps here: a
    # diagonal 2.7e-4 too small, tests/test_oracle_golden.py::test_gram_chain16_holds_at_other_shapes) - is as far from the exact Gram as the
    # oracle's as-it-is run is; the two need not agree with each other more closely than that: `want32` is torch.bmm on the CPU of the machine
    # this test runs on, which may order the sum differently (the GPU boxes' CPUs do: DESIGN section 5, item 5)
    eng = DEQSCIEngine(net, max_iter=6)
    assert eng.anderson_arith == "reference"
    got_ref = eng.reconstruct(G(y), G(Phi)).cpu().numpy()
    print("512x512x16 SimpleCNN@6: reference arithmetic vs the oracle as it is %.2e, vs the exact-Gram oracle %.2e (exact-Gram engine vs the oracle as it is: %.2e)"
          % (rel_l2(got_ref, want32.numpy()), rel_l2(got_ref, want64.numpy()), rel_l2(got, want32.numpy())))
    assert rel_l2(got_ref, want32.numpy()) < 1e-3 and 2e-5 < rel_l2(got_ref, want64.numpy()) < 1e-3


def test_engine_is_deterministic_and_options():
    """Reductions are two-stage with fixed order (no atomics on data): two runs are bit-identical; the
    optional dead f-call and disabling residual polling do not change the result."""
    d = _clip("traffic_cacti.mat")
    Phi = d["mask"][None].to(DEV)
    y = d["meas"].permute(2, 0, 1)[:2].contiguous().to(DEV)
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 14)[0].nonlinear_op
    eng = DEQSCIEngine(net, max_iter=14)
    a = eng.reconstruct(y, Phi).clone()
    info = dict(eng.last_info)
    b = eng.reconstruct(y, Phi)
    assert torch.equal(a, b) and eng.last_info["res"] == info["res"] and info["f_calls"] == 15
    eng2 = DEQSCIEngine(net, max_iter=14, extra_call=True, poll_residual=False)
    c = eng2.reconstruct(y, Phi)
    assert torch.equal(a, c) and eng2.last_info["f_calls"] == 16 and eng2.last_info["res"] == info["res"]
    assert len(info["res_per_sample"]) == 2 and all(r > 0 for r in info["res_per_sample"])


@pytest.mark.parametrize("shape", [(3, 16, 32), (2, 13, 37), (1, 128, 128), (5, 8, 64)])
def test_ffdnet_tail_kernel_vs_torch(shape):
    """conv3x3(64->4) + pixel_shuffle(2) as one HIP stencil vs F.conv2d + F.pixel_shuffle (incl. ragged tiles)."""
    import torch.nn.functional as Fn
    n, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(3)
    h = torch.randn(n, 64, H, W, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.randn(4, 64, 3, 3, device=DEV, generator=g) * 0.05
    want = Fn.pixel_shuffle(Fn.conv2d(h.double(), w.double(), padding=1), 2)
    got = _hip.ffdnet_tail(h, _hip.pack_tail_weights(w))
    assert got.shape == (n, 1, 2 * H, 2 * W)
    ref32 = Fn.pixel_shuffle(Fn.conv2d(h, w, padding=1), 2)
    e_got = float((got.double() - want).norm() / want.norm())
    e_ref = float((ref32.double() - want).norm() / want.norm())
    assert e_got < 1e-6 and e_got < 4 * e_ref + 1e-7
    # with the previous layer's bias + ReLU folded into the read
    b = torch.randn(64, device=DEV, generator=g)
    hb = torch.relu(h + b.view(1, -1, 1, 1))
    want_b = Fn.pixel_shuffle(Fn.conv2d(hb.double(), w.double(), padding=1), 2)
    got_b = _hip.ffdnet_tail(h, _hip.pack_tail_weights(w), in_bias=b)
    assert float((got_b.double() - want_b).norm() / want_b.norm()) < 1e-6
    # the matrix-core form of the layer on an sp16 input (taps in the N dimension, split-fp16 products): what the engine runs behind
    # a stack of split-fp16 layers - with the fixed 2^8 scale, and with the scale following the data (a measured range slot)
    got_mm = _hip.tail_split16(_hip.to_split16(h), _hip.TailSplit16Weights(w))
    e_mm = float((got_mm.double() - want).norm() / want.norm())
    assert got_mm.shape == got.shape and e_mm < 3e-7 and e_mm < 2 * e_ref + 1e-7, (e_mm, e_ref)
    for scale in (1.0, 1e-5, 3e3):
        slot = torch.zeros(n, device=DEV)                      # one range per image
        hs = (h * scale).contiguous(memory_format=torch.channels_last)
        _hip.absmax(hs, slot)
        assert torch.equal(slot, hs.abs().amax(dim=(1, 2, 3)))
        got_s = _hip.tail_split16(_hip.to_split16(hs, rng=slot), _hip.TailSplit16Weights(w))
        want_s = Fn.pixel_shuffle(Fn.conv2d(hs.double(), w.double(), padding=1), 2)
        e_s = float((got_s.double() - want_s).norm() / want_s.norm())
        assert e_s < 3e-7 and e_s < 2 * e_ref + 1e-7, (scale, e_s, e_ref)


@pytest.mark.parametrize("shape", [(3, 32, 64), (2, 26, 38), (1, 256, 256), (4, 16, 96), (130, 128, 128), (140, 100, 124), (64, 256, 256)])   # last three: the matrix-core variant (>= 512 tiles of 32 x 32), ragged and not
def test_ffdnet_head_kernel_vs_torch(shape):
    """sigma map + pixel_unshuffle + conv3x3(5->64) + ReLU as one HIP kernel vs the torch ops (incl. ragged
    tiles, per-image sigma and the zero-padded sigma ring at the border); small launches run the vector-ALU stencil,
    large ones the MFMA formulation of the same sum."""
    import torch.nn.functional as Fn
    n, H2, W2 = shape
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(n, 1, H2, W2, device=DEV, generator=g)
    w = torch.randn(64, 5, 3, 3, device=DEV, generator=g) * 0.1
    for sig in (torch.rand(n, device=DEV, generator=g), torch.rand(1, device=DEV, generator=g)):
        inp = torch.cat((sig.expand(n).view(n, 1, 1, 1).expand(n, 1, H2 // 2, W2 // 2), Fn.pixel_unshuffle(x, 2)), 1)
        want = torch.relu(Fn.conv2d(inp.double(), w.double(), padding=1))
        ref32 = torch.relu(Fn.conv2d(inp, w, padding=1))
        got = _hip.ffdnet_head(x, _hip.pack_head_weights(w), sig)
        assert got.shape == (n, 64, H2 // 2, W2 // 2) and got.is_contiguous(memory_format=torch.channels_last)
        e_got = float((got.double() - want).norm() / want.norm())
        e_ref = float((ref32.double() - want).norm() / want.norm())
        assert e_got < 1e-6 and e_got < 4 * e_ref + 1e-7
        # the matrix-core form on the f16 pipes writing the sp16 layout (input taps split into hi + lo on the fly): what the engine runs
        # in front of split-fp16 layers; every plane fully written
        sp = _hip.Sp16.empty(n, H2 // 2, W2 // 2, DEV)
        sp.t.fill_(float("nan"))
        got_mm = _hip.ffdnet_head_split16(x, _hip.HeadSplit16Weights(w), sig, out=sp)
        assert got_mm is sp and bool(torch.isfinite(sp.t).all())
        e_mm = float((sp.to_nchw().double() - want).norm() / want.norm())
        assert e_mm < 3e-7 and e_mm < 2 * e_ref + 1e-7, (e_mm, e_ref)
        # ... with the scales following the data: the image's range measured by absmax, the output's by a measuring launch
        if n <= 4:
            for scale in (1.0, 1e-4):
                xs, ss = x * scale, sig * scale
                rng = torch.zeros(2, n, device=DEV)               # [image range | output range] x one slot per image
                _hip.absmax(xs, rng[0])
                _hip.ffdnet_head_split16(xs, _hip.HeadSplit16Weights(w), ss, out=sp, in_rng=rng[0], out_exp=0, track=rng[1])
                inp_s = torch.cat((ss.expand(n).view(n, 1, 1, 1).expand(n, 1, H2 // 2, W2 // 2), Fn.pixel_unshuffle(xs, 2)), 1)
                want_s = torch.relu(Fn.conv2d(inp_s.double(), w.double(), padding=1))
                assert float((rng[1] / want_s.abs().amax(dim=(1, 2, 3)).float() - 1).abs().max()) < 1e-6
                sp.t.fill_(float("nan"))
                _hip.ffdnet_head_split16(xs, _hip.HeadSplit16Weights(w), ss, out=sp, in_rng=rng[0], out_rng=rng[1])
                assert sp.exponents() == [_hip.act_exp(v) for v in rng[1].tolist()]
                top = sp.t[:, :, 0].float().abs().amax(dim=(1, 2, 3, 4, 5))                # every image fills its own [2^11, 2^12)
                assert bool((top >= 2048).all()) and bool((top <= 4096).all())
                e_s = float((sp.to_nchw().double() - want_s).norm() / want_s.norm())
                assert e_s < 3e-7 and e_s < 2 * e_ref + 1e-7, (scale, e_s, e_ref)


class _Affine(torch.nn.Module):
    """Deterministic toy plugin with tag 'denoiser' (predicts noise): used to drive the engine kernels with
    settings the shipped denoisers never use (m != 5, beta != 1, B = 4 / 16, shared vs per-sample masks)."""
    tag = "denoiser"

    def forward(self, x):
        return 0.4 * x - 0.05


@pytest.mark.parametrize("cfg", [dict(B=8, m=3, beta=0.7, lam=1e-3, bsz=2, shared=False),
                                 dict(B=4, m=8, beta=1.0, lam=1e-2, bsz=1, shared=False),
                                 dict(B=16, m=5, beta=0.5, lam=1e-2, bsz=3, shared=True),
                                 dict(B=5, m=4, beta=0.9, lam=1e-2, bsz=2, shared=False)])
def test_engine_anderson_settings_vs_oracle(cfg):
    """mix_gap with the (1-beta) G term, history depths 3..8, B in {4,5,8,16} (B=5: unfused generic kernels)."""
    B, bsz = cfg["B"], cfg["bsz"]
    Phi, Phie, x, z, y, Ps = make_case(bsz, 24, 20, B, seed=B + cfg["m"], shared=cfg["shared"])
    Pse = Ps.expand(bsz, 24, 20)

    def f_cpu(zz):
        z1 = orc.gap_update(zz, y, Phie, Pse)
        return z1 - (0.4 * z1 - 0.05)
    zs, res = orc.andersonexp(f_cpu, orc.initial_point(y, Phie), m=cfg["m"], lam=cfg["lam"], max_iter=11, tol=1e-9, beta=cfg["beta"])
    want = f_cpu(zs)
    eng = DEQSCIEngine(_Affine(), max_iter=11, m=cfg["m"], beta=cfg["beta"], lam=cfg["lam"], tol=1e-9)
    got = eng.reconstruct(G(y), G(Phi))
    assert rel_l2(got.cpu().numpy(), want.numpy()) < 2e-5
    assert abs(eng.last_info["res"] - res) < 2e-2 * res + 1e-9


def test_cli_end_to_end(tmp_path):
    """The reference's test_*.sh usage through the build's CLI: loads the shipped archive, runs the harness over
    data/test_gray, prints PSNRs and writes the 64 PNGs."""
    from deqsci_amd.cli import main as cli_main
    avg = cli_main(["--denoiser", "SimpleCNN", "--testpath", orc.DATA_DIR + "/", "--savepath", str(tmp_path) + "/",
                    "--and_maxiters", "10", "--inference", "True"])
    assert abs(avg - _golden_meta("SimpleCNN_anderson_10")["avg_psnr"]) < 0.01
    pngs = sorted(os.listdir(tmp_path))
    assert len(pngs) == 64 and "traffic_cacti.mat_reconstruction_47.png" in pngs
    from PIL import Image
    im = np.asarray(Image.open(os.path.join(tmp_path, "drop8_cacti.mat_reconstruction_0.png")))
    assert im.shape == (256, 256) and im.dtype == np.uint8 and im.std() > 5


def test_cli_end_to_end_ffdnet_from_a_reference_pickle(tmp_path):
    """The literal test_ffdnet.sh form (test_ffdnet.sh:1-7, video_sci_proxgrad.py:210-227): `--denoiser ffdnet --loadpath <pickle>` where
    the pickle is the reference's training-checkpoint format - {'solver_state_dict', 'epoch', ...} with DataParallel's `module.`
    prefix on every `nonlinear_op.*` key - written here from the shipped tensors.  Through the CLI: checkpoint load, harness over
    data/test_gray, PSNR printout, 64 PNGs; the harness average equals the reference's own run (golden) to 0.01 dB."""
    from deqsci_amd.cli import main as cli_main
    ff, _ = checkpoint.read_state_dict(checkpoint.shipped("ffdnet_gray"))
    path = str(tmp_path / "ffdnet.ckpt")
    torch.save({"solver_state_dict": {"module.nonlinear_op." + k: v for k, v in ff.items()}, "epoch": 11,
                "optimizer_state_dict": {"state": {}, "param_groups": []}, "scheduler_state_dict": {"last_epoch": 11}}, path)
    out = tmp_path / "png"
    out.mkdir()
    avg = cli_main(["--denoiser", "ffdnet", "--loadpath", path, "--testpath", orc.DATA_DIR + "/", "--savepath", str(out) + "/",
                    "--and_maxiters", "10", "--inference", "True", "--gpu_ids", "0"])
    assert abs(avg - _golden_meta("ffdnet_anderson_10")["avg_psnr"]) < 0.01
    assert len(os.listdir(out)) == 64
    with pytest.raises(FileNotFoundError):                     # the reference would run on random weights here (:211)
        cli_main(["--denoiser", "ffdnet", "--loadpath", str(tmp_path / "missing.ckpt"), "--testpath", orc.DATA_DIR + "/", "--savepath",
                  str(out) + "/", "--and_maxiters", "10"])


class _ToyClean(torch.nn.Module):
    conv3d = False

    def forward(self, x):
        return 0.8 * x + 0.02 * torch.tanh(x)


def test_admm_variant_vs_reference_golden():
    """EquilibriumADMMSCI / admmexp / DEQFixedPointADMM / initial_point_admm against the reference."""
    g = np.load(os.path.join(GOLDEN, "admm_toy.npz"))
    Phi, y, Ps = G(g["Phi"]), G(g["y"]), G(g["Phi_sum"])
    f = deqsci_amd.EquilibriumADMMSCI(A=deqsci_amd.A_torch_, At=deqsci_amd.At_torch_, nonlinear_operator=_ToyClean(), eta=0.2)
    init = deqsci_amd.initial_point_admm(y, Phi, Ps, None)
    assert np.array_equal(init[0].cpu().numpy(), g["x0"]) and float(init[1].abs().max()) == 0.0
    z1, u1 = f(init[0], init[1], y, Phi, Ps)
    np.testing.assert_allclose(z1.cpu().numpy(), g["step_z"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(u1.cpu().numpy(), g["step_u"], rtol=1e-5, atol=1e-5)
    for it, tol in ((8, 1e-9), (40, 5e-2)):
        deq = deqsci_amd.DEQFixedPointADMM(f, deqsci_amd.admmexp, None, max_iter=it, tol=tol)
        z = deq.forward(y, Phi, Ps, initial_point=[init[0].clone(), init[1].clone()])
        assert rel_l2(z.cpu().numpy(), g[f"it{it}_z"]) < 1e-5
        assert abs(deq.forward_res - float(g[f"it{it}_res"])) < 1e-3 * float(g[f"it{it}_res"]) + 1e-6   # 4.5e-8 at it=8: round-off level


@pytest.mark.parametrize("shape", [(2, 16, 16), (1, 128, 128), (3, 20, 36), (1, 18, 14), (2, 17, 23), (1, 1, 1), (70, 64, 80), (33, 50, 46), (300, 16, 16)])
def test_winograd_conv64_vs_torch(shape):
    """Winograd F(2x2,3x3) MFMA conv 64->64 (+bias+ReLU) vs conv2d in fp64: transpose-detecting (random asymmetric
    weights), ragged tile edges, with and without the fused epilogue.  The last three shapes have more block tiles than
    persistent workgroups (runs of several tiles per workgroup, ragged per-XCD ranges, the pipeline crossing tile and
    image boundaries)."""
    import torch.nn.functional as Fn
    n, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(n, 64, H, W, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.randn(64, 64, 3, 3, device=DEV, generator=g) * 0.05
    b = torch.randn(64, device=DEV, generator=g)
    U = _hip.pack_winograd_weights(w)
    ref = Fn.conv2d(x.double(), w.double(), padding=1)
    ref32 = Fn.conv2d(x, w, padding=1)
    got = _hip.conv3x3_c64_winograd(x, U, None, relu=False)
    assert got.is_contiguous(memory_format=torch.channels_last) and got.shape == x.shape
    e_got = float((got.double() - ref).norm() / ref.norm())
    e_ref = float((ref32.double() - ref).norm() / ref.norm())
    assert e_got < 2e-6, (e_got, e_ref)
    got2 = _hip.conv3x3_c64_winograd(x, U, b, relu=True)
    want2 = torch.relu(ref + b.double().view(1, -1, 1, 1))
    assert float((got2.double() - want2).norm() / want2.norm()) < 2e-6


@pytest.mark.parametrize("shape", [(1, 16, 32), (8, 128, 128), (3, 40, 56), (1, 18, 14), (2, 17, 23), (1, 1, 1), (70, 64, 80), (33, 50, 46),
                                   (300, 16, 16), (2, 250, 130)])
def test_winograd44_conv64_vs_torch(shape):
    """Winograd F(4x4,3x3) MFMA conv 64->64 (+bias+ReLU) vs conv2d in fp64: random asymmetric weights, block tiles that
    stick out of the image on every side (tiles are 16 x 32 outputs), single block tiles and runs of many per workgroup
    crossing image boundaries, with and without bias/ReLU.  Bound on RANDOM data: 2.5e-6 (measured 1.2-1.7e-6; F(2x2,3x3): 2e-7, direct
    fp32: 3e-7) - the network's own data are another matter: test_conv64_rounding_on_the_networks_own_data."""
    import torch.nn.functional as Fn
    n, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(12)
    x = torch.randn(n, 64, H, W, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.randn(64, 64, 3, 3, device=DEV, generator=g) * 0.05
    b = torch.randn(64, device=DEV, generator=g)
    U = _hip.pack_winograd44_weights(w)
    ref = Fn.conv2d(x.double(), w.double(), padding=1)
    out = torch.full_like(x, float("nan"))
    got = _hip.conv3x3_c64_winograd44(x, U, None, relu=False, out=out)
    assert got.is_contiguous(memory_format=torch.channels_last) and got.shape == x.shape
    assert float((got.double() - ref).norm() / ref.norm()) < 2.5e-6
    got2 = _hip.conv3x3_c64_winograd44(x, U, b, relu=True)
    want2 = torch.relu(ref + b.double().view(1, -1, 1, 1))
    assert float((got2.double() - want2).norm() / want2.norm()) < 2.5e-6
    # repeated launches on one stream (the persistent pipeline leaves nothing behind) and the selecting front end
    got3 = _hip.conv3x3_c64(x, _hip.pack_conv64_weights(w), b, relu=True, policy="fast32")
    assert float((got3.double() - want2).norm() / want2.norm()) < 2.5e-6
    with pytest.raises(_hip.DeqsciHipError):                                   # the other kernel's pack would be read out of bounds
        _hip.conv3x3_c64_winograd44(x, _hip.pack_winograd_weights(w), b, relu=True)
    with pytest.raises(_hip.DeqsciHipError):
        _hip.conv3x3_c64_winograd(x, U, b, relu=True)
    assert torch.equal(_hip.conv3x3_c64_winograd44(x, U, b, relu=True), got2)


@pytest.mark.parametrize("shape", [(1, 16, 32), (8, 128, 128), (3, 40, 56), (2, 17, 23), (70, 64, 80), (2, 250, 130)])
def test_winograd44_blk32_layouts_vs_torch(shape):
    """The F(4x4,3x3) kernel with its own activation layout (blk32) on the input side, the output side, and both, and a chain of
    three layers NHWC -> blk32 -> blk32 -> NHWC as the engine runs a denoiser.  The padding columns of a blk32 buffer (W not a
    multiple of 32) are NaN on the way in and must still be NaN on the way out: never read, never written."""
    import torch.nn.functional as Fn
    n, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(13)
    x = torch.randn(n, 64, H, W, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    ws = [torch.randn(64, 64, 3, 3, device=DEV, generator=g) * 0.05 for _ in range(3)]
    bs = [torch.randn(64, device=DEV, generator=g) * 0.1 for _ in range(3)]
    Us = [_hip.pack_winograd44_weights(w) for w in ws]
    ref = torch.relu(Fn.conv2d(x.double(), ws[0].double(), bs[0].double(), padding=1))

    def err(a, b):
        return float((a.double() - b).norm() / b.norm())
    xb = _hip.Blk32.from_nchw(x)
    assert torch.equal(xb.to_nchw(), x)                                     # the host-side converters are inverses
    got_in = _hip.conv3x3_c64_winograd44(xb, Us[0], bs[0], True)           # blk32 -> NHWC
    assert err(got_in, ref) < 2.5e-6
    ob = _hip.Blk32.empty(n, H, W, DEV)
    ob.t.fill_(float("nan"))
    got_out = _hip.conv3x3_c64_winograd44(x, Us[0], bs[0], True, out=ob, out_blk=True)   # NHWC -> blk32
    assert err(got_out.to_nchw(), ref) < 2.5e-6
    if W % 32:
        pos = _hip.Blk32._pos().to(DEV)
        pad = got_out.t[:, :, :, -1][:, :, :, pos][:, :, :, W % 32:]      # last block, columns >= W
        assert bool(torch.isnan(pad).all())
    got_both = _hip.conv3x3_c64_winograd44(xb, Us[0], bs[0], True, out_blk=True)          # blk32 -> blk32
    assert err(got_both.to_nchw(), ref) < 2.5e-6
    # a stack of three layers, as the engine chains them
    h = x
    want = x.double()
    for i in range(3):
        h = _hip.conv3x3_c64_winograd44(h, Us[i], bs[i], True, out_blk=(i < 2))
        want = torch.relu(Fn.conv2d(want, ws[i].double(), bs[i].double(), padding=1))
    assert err(h, want) < 5e-6


def test_conv64_front_end_picks_the_faster_kernel():
    """One block tile per CU and more -> the 16 x 32-tile kernel (split-fp16 under "fast", F(4x4,3x3) under "fast32"); below ->
    F(2x2,3x3) (measured: profiles/r02_w44_shapes.jsonl, profiles/r03_s16_time.jsonl); and the front end runs what it names."""
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert cus == 256
    assert _hip.conv64_kernel_for(8, 128, 128) == "s16" and _hip.conv64_kernel_for(64, 128, 128) == "s16"
    assert _hip.conv64_kernel_for(1, 128, 128) == "f22" and _hip.conv64_kernel_for(4, 128, 128) == "f22"
    assert _hip.conv64_kernel_for(2, 256, 256) == "s16" and _hip.conv64_kernel_for(1, 256, 256) == "f22"
    assert _hip.conv64_kernel_for(8, 128, 128, policy="fast32") == "f44" and _hip.conv64_kernel_for(4, 128, 128, policy="fast32") == "f22"
    import torch.nn.functional as Fn
    g = torch.Generator(device=DEV).manual_seed(21)
    x = torch.randn(8, 64, 128, 128, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.randn(64, 64, 3, 3, device=DEV, generator=g) * 0.05
    b = torch.randn(64, device=DEV, generator=g)
    Wts = _hip.pack_conv64_weights(w)
    want = torch.relu(Fn.conv2d(x.double(), w.double(), b.double(), padding=1))
    seen = []
    _hip.CONV64_EVENT_HOOK = lambda kind, n, H, W: seen.append(kind)           # returns None: untimed launches, but the kind is reported
    try:
        for pol, kind, tol in (("fast", "s16", 4e-7), ("fast32", "f44", 2.5e-6), ("f22", "f22", 4e-7)):
            got = _hip.conv3x3_c64(x, Wts, b, True, policy=pol)
            assert seen[-1] == kind and got.is_contiguous(memory_format=torch.channels_last)
            assert float((got.double() - want).norm() / want.norm()) < tol, pol
        mid = _hip.conv3x3_c64(x, Wts, b, True, policy="fast", chain=True)       # the kernel's own layout between layers ...
        assert isinstance(mid, _hip.Sp16) and isinstance(_hip.conv3x3_c64(x, Wts, b, True, policy="fast32", chain=True), _hip.Blk32)
        out2 = _hip.conv3x3_c64(mid, Wts, b, True, policy="f22")                  # ... which pins the next layer's kernel
        assert seen[-1] == "s16" and out2.shape == x.shape
    finally:
        _hip.CONV64_EVENT_HOOK = None


@pytest.mark.parametrize("shape", [(1, 16, 32), (8, 128, 128), (3, 40, 56), (1, 18, 14), (2, 17, 23), (1, 1, 1), (70, 64, 80), (33, 50, 46),
                                   (300, 16, 16), (2, 250, 130)])
def test_split16_conv64_vs_torch(shape):
    """Split-fp16 direct convolution 64->64 (+bias+ReLU) on the f16 matrix cores vs conv2d in fp64: random asymmetric weights, block tiles
    that stick out of the image on every side (16 x 32 outputs), single tiles and runs of many per workgroup, both output forms
    (sp16 for the next layer / fp32 channels_last), a three-layer chain, repeated launches.  Bound on random data: 2.5e-7 (measured
    1.2-1.4e-7; MIOpen's direct fp32 convolution on the same data: 1.6-3.6e-7; F(2x2,3x3) 2e-7; F(4x4,3x3) 1.2-1.7e-6).
    The sp16 round trip itself (fp32 -> hi + lo -> fp32) is exact to 2^-22."""
    import torch.nn.functional as Fn
    n, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(12)
    x = torch.randn(n, 64, H, W, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    ws = [torch.randn(64, 64, 3, 3, device=DEV, generator=g) * 0.05 for _ in range(3)]
    bs = [torch.randn(64, device=DEV, generator=g) * 0.3 for _ in range(3)]
    Ws = [_hip.Split16Weights(w) for w in ws]

    def err(a, b):
        return float((a.double() - b).norm() / b.norm())
    xs = _hip.to_split16(x)
    assert err(xs.to_nchw(), x.double()) < 2.0 ** -22
    ref = Fn.conv2d(x.double(), ws[0].double(), padding=1)
    got = _hip.conv3x3_c64_split16(xs, Ws[0], None, relu=False, out_f32=True)
    assert got.is_contiguous(memory_format=torch.channels_last) and got.shape == x.shape and err(got, ref) < 2.5e-7
    want = torch.relu(ref + bs[0].double().view(1, -1, 1, 1))
    o_sp = _hip.Sp16.empty(n, H, W, DEV)
    o_sp.t.fill_(float("nan"))
    got_sp = _hip.conv3x3_c64_split16(xs, Ws[0], bs[0], True, out=o_sp)
    assert got_sp is o_sp and bool(torch.isfinite(o_sp.t).all()) and err(o_sp.to_nchw(), want) < 2.5e-7
    got_f = _hip.conv3x3_c64_split16(xs, Ws[0], bs[0], True, out_f32=True)
    assert err(got_f, want) < 2.5e-7
    assert torch.equal(_hip.conv3x3_c64_split16(xs, Ws[0], bs[0], True, out_f32=True), got_f)      # repeated launches: nothing left behind
    h, wd = xs, x.double()
    for i in range(3):                                                       # sp16 -> sp16 -> fp32, as the engine chains a denoiser
        h = _hip.conv3x3_c64_split16(h, Ws[i], bs[i], True, out_f32=(i == 2))
        wd = torch.relu(Fn.conv2d(wd, ws[i].double(), bs[i].double(), padding=1))
    assert err(h, wd) < 5e-7
    with pytest.raises(_hip.DeqsciHipError):
        _hip.conv3x3_c64_split16(x, Ws[0], bs[0], True)                       # an fp32 tensor is not an Sp16
    with pytest.raises(_hip.DeqsciHipError):
        _hip.conv3x3_c64_split16(xs, _hip.pack_winograd_weights(ws[0]), bs[0], True)


@pytest.mark.parametrize("kind,weights", [("ffdnet", "ffdnet_gray"), ("SimpleCNN", "cnn")])
def test_engine_split16_on_ragged_sizes(kind, weights):
    """The whole split-fp16 path of the engine (head -> sp16 -> 64->64 layers -> matrix-core tail) forced at sizes where every kernel's
    tiles stick out of the image (frames of 100 x 76: FFDNet's half-resolution planes are 50 x 38, neither a multiple of 16 nor 32),
    against the same engine on the fp32 Winograd F(2x2,3x3) kernels and against the CPU oracle."""
    g = torch.Generator().manual_seed(31)
    H, W, B = 100, 76, 8
    Phi = (torch.rand(2, H, W, B, generator=g) < 0.5).float()
    x = torch.rand(2, H, W, B, generator=g)
    y = orc.sci_forward(x, Phi)
    net = build_pipeline(kind, checkpoint.shipped(weights), 6)[0].nonlinear_op
    seen = []
    _hip.CONV64_EVENT_HOOK = lambda k, n, h, w: seen.append((k, n, h, w))
    try:
        a = DEQSCIEngine(net, max_iter=6, use_graph=False, conv64="s16").reconstruct(G(y), G(Phi))
        # (conv64="s16" means the DIRECT kernel - ADVICE r5: until round 6 SimpleCNN's two middle layers took the Winograd form behind the measuring
        #  f-call even then; FFDNet's run at this size is below a stack launch's worth and goes out per layer)
        assert {k for k, *_ in seen} == {"s16"} and seen[0][1:] == ((16, 50, 38) if kind == "ffdnet" else (16, 100, 76))
        del seen[:]
        a16 = DEQSCIEngine(net, max_iter=6, use_graph=False, conv64="s16", stack_kernel="s16").reconstruct(G(y), G(Phi))
        assert {k for k, *_ in seen} == {"s16"} and rel_l2(a.cpu().numpy(), a16.cpu().numpy()) < 2e-6
        del seen[:]
        b = DEQSCIEngine(net, max_iter=6, use_graph=False, conv64="f22").reconstruct(G(y), G(Phi))
        assert {k for k, *_ in seen} == {"f22"}
    finally:
        _hip.CONV64_EVENT_HOOK = None
    assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 2e-5
    Ps = orc.phi_sum(Phi)
    want, _ = orc.deq_forward(orc.ProxGradSCI(kind), orc.andersonexp, y, Phi, Ps, orc.initial_point(y, Phi), m=5, beta=1.0, lam=1e-2, max_iter=6, tol=1e-5)
    assert rel_l2(a.cpu().numpy(), want.numpy()) < 1e-4


def test_engine_warns_when_split16_overflows():
    """The engine end of the same promise, with the scales pinned at 2^8 (act_range="fixed", the round-3 arithmetic): measurements 255x too
    large (the classic forgotten /255) push FFDNet's activations beyond fp16's range inside the split-fp16 layers; the reconstruction comes
    back non-finite and the engine says why - conv64='fast32' has no limit, and the default policy falls back to it by itself.  With the
    scales following the data (the default) the same input is simply reconstructed: no warning, the fp32 kernels' result to 1e-5."""
    d = _clip("traffic_cacti.mat")
    Phi = d["mask"][None].to(DEV)
    y = (d["meas"][None, ..., 0] * 2000.0).contiguous().to(DEV)
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 6)[0].nonlinear_op
    import functools
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        rec_data = DEQSCIEngine(net, max_iter=6, use_graph=False).reconstruct(y, Phi)
        rec32_ = DEQSCIEngine(net, max_iter=6, use_graph=False, conv64="fast32").reconstruct(y, Phi)
    assert bool(torch.isfinite(rec_data).all()) and rel_l2(rec_data.cpu().numpy(), rec32_.cpu().numpy()) < 1e-5
    Fixed = functools.partial(DEQSCIEngine, act_range="fixed")
    eng = Fixed(net, max_iter=6, use_graph=False, conv64="fast")          # the explicit policy: told, not rescued
    with pytest.warns(RuntimeWarning, match="fp16's range"):
        rec = eng.reconstruct(y, Phi)
    assert not bool(torch.isfinite(rec).all()) and eng.conv64_policy == "fast"
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        rec32 = Fixed(net, max_iter=6, use_graph=False, conv64="fast32").reconstruct(y, Phi)
    assert bool(torch.isfinite(rec32).all()) and torch.equal(rec32, rec32_)
    # the default ("auto") keeps the reference's fp32 range: it says what happened and redoes THAT call on the fp32 MFMA kernels - the
    # engine's policy, and what it does with the next (sane) input, are left alone (not sticky, visible in last_info)
    auto = Fixed(net, max_iter=6, use_graph=False)
    y_ok = d["meas"][None, ..., 0].contiguous().to(DEV)
    ok_before = auto.reconstruct(y_ok, Phi)
    assert auto.last_info["conv64_fallback"] is None
    with pytest.warns(RuntimeWarning, match="fp16's range"):
        rec_auto = auto.reconstruct(y, Phi)
    assert torch.equal(rec_auto, rec32) and auto.conv64_policy == "fast" and math.isfinite(auto.last_info["res"])
    assert auto.last_info["conv64_fallback"] == "fast32"
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert torch.equal(auto.reconstruct(y_ok, Phi), ok_before) and auto.last_info["conv64_fallback"] is None
    # the same through the hipGraph path (one measurement per call, the reference's usage): the captured graph survives a fallback call
    g = Fixed(net, max_iter=6, use_graph=True)
    for _ in range(3):
        assert torch.equal(g.reconstruct(y_ok, Phi), ok_before)
    assert g.last_info["graph"] is True
    with pytest.warns(RuntimeWarning, match="fp16's range"):
        assert torch.equal(g.reconstruct(y, Phi), rec32) and g.last_info["conv64_fallback"] == "fast32" and g.last_info["graph"] is False
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert torch.equal(g.reconstruct(y_ok, Phi), ok_before) and g.last_info["graph"] is True
    # a Picard run of zero iterations still reports the residual of its one f-call (it used to leave the pinned row at its initial inf)
    p0 = DEQSCIEngine(net, iterator="picard", max_iter=0, use_graph=False)
    p0.reconstruct(y_ok, Phi)
    assert math.isfinite(p0.last_info["res"]) and p0.last_info["conv64_fallback"] is None


def test_split16_overflow_is_loud():
    """Activations beyond fp16's range (2^8 |y| >= 65504) must surface as inf / NaN at the END of a stack of layers, never as a wrong
    finite number: the overflowing layer writes inf pieces, the next layer's accumulators become inf / NaN, and the ReLU of this
    kernel is the NaN-propagating maximum (fmaxf would turn NaN into 0 and hide it)."""
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.rand(1, 64, 16, 32, device=DEV, generator=g).contiguous(memory_format=torch.channels_last) * 40.0
    w = torch.ones(64, 64, 3, 3, device=DEV) * 0.5                          # y ~ 64 * 9 * 20 * 0.5 = 5760 >> 255
    Wb = _hip.Split16Weights(w)
    out = _hip.conv3x3_c64_split16(_hip.to_split16(x), Wb, None, True)
    assert not bool(torch.isfinite(out.t).all())
    ok = _hip.conv3x3_c64_split16(_hip.to_split16(x), Wb, None, True, out_f32=True)   # the fp32 output form has no such limit
    want = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    assert float((ok.double() - want).norm() / want.norm()) < 2.5e-7
    # two more layers with mixed-sign weights and ReLU: still not finite at the end
    w2 = torch.randn(64, 64, 3, 3, device=DEV, generator=g) * 0.05
    h = out
    for last in (False, True):
        h = _hip.conv3x3_c64_split16(h, _hip.Split16Weights(w2), torch.zeros(64, device=DEV), True, out_f32=last)
    assert not bool(torch.isfinite(h).all())


@pytest.mark.parametrize("scale", [1.0, 1e-2, 1e-4, 1e-6, 1e3])
def test_split16_conv64_follows_the_data_scale(scale):
    """VERDICT r3 #1 (the dynamic-range hole).  fp32 - the reference's arithmetic, solvers/equilibrium_solvers_yaping.py:397-420 - is
    scale-free; the fp16 pieces of the split-fp16 layers are not, so their power-of-two scales follow the data: every activation's
    range is measured on the device (a measuring launch of the layer folds max |output| into a range slot) and the writer and the reader
    of the activation derive the same exponent from that word.  Per layer and as a three-layer chain, on ReLU-like activations scaled by
    1 ... 1e-6 (and 1e3, beyond the fixed scale's overflow): the error against float64 stays at the unscaled level (<= 2.5e-7) and no worse
    than the fp32 direct convolution of the same operands (MIOpen).  With the scale pinned at 2^8 the small inputs lose their low pieces."""
    import torch.nn.functional as Fn
    n, H, W = 8, 64, 96
    g = torch.Generator(device=DEV).manual_seed(21)
    x = (torch.relu(torch.randn(n, 64, H, W, device=DEV, generator=g)) * scale).contiguous(memory_format=torch.channels_last)
    ws = [torch.randn(64, 64, 3, 3, device=DEV, generator=g) * 0.05 for _ in range(3)]
    bs = [torch.randn(64, device=DEV, generator=g) * 0.3 * scale for _ in range(3)]
    Ws = [_hip.Split16Weights(w) for w in ws]

    def err(a, b):
        return float((a.double() - b).norm() / b.norm())
    x = x * torch.logspace(0, -3, n, device=DEV).view(n, 1, 1, 1)          # ... and the images of the batch three decades apart: ranges are per image
    x = x.contiguous(memory_format=torch.channels_last)
    rng = torch.zeros(4, n, device=DEV)

    def full(t):
        top = t.t[:, :, 0].float().abs().amax(dim=(1, 2, 3, 4, 5))
        return bool((top >= 2048).all()) and bool((top <= 4096).all())    # (a maximum just under 2^12 rounds to 4096 in fp16)

    def rel_img(a, b):                                                     # worst image: every one is held to the bound, not the batch norm
        a, b = a.double(), b.double()
        return float(((a - b).flatten(1).norm(dim=1) / b.flatten(1).norm(dim=1)).max())
    _hip.absmax(x, rng[0])
    h, wd, w32 = _hip.to_split16(x, rng=rng[0]), x.double(), x
    assert err(h.to_nchw(), wd) < 2.0 ** -22 and full(h)
    for i in range(3):
        wd = torch.relu(Fn.conv2d(wd, ws[i].double(), bs[i].double(), padding=1))
        w32 = torch.relu(Fn.conv2d(w32, ws[i], bs[i], padding=1))
        assert _hip.conv3x3_c64_split16(h, Ws[i], bs[i], True, track=rng[i + 1]) is None                   # measure ...
        assert float((rng[i + 1] / wd.abs().amax(dim=(1, 2, 3)).float() - 1).abs().max()) < 1e-5
        nxt = _hip.Sp16.empty(n, H, W, DEV)
        nxt.t.fill_(float("nan"))
        h = _hip.conv3x3_c64_split16(h, Ws[i], bs[i], True, out=nxt, out_rng=rng[i + 1])                    # ... then write with those ranges
        assert h is nxt and bool(torch.isfinite(h.t).all()) and full(h)
        e_s16, e_f32 = rel_img(h.to_nchw(), wd), rel_img(w32, wd)
        assert e_s16 < 2.5e-7 * (i + 1) and e_s16 < 1.1 * e_f32 + 5e-8, (scale, i, e_s16, e_f32)
    f = _hip.conv3x3_c64_split16(_hip.to_split16(x, rng=rng[0]), Ws[0], bs[0], True, out_f32=True)     # the fp32 output form reads the ranges too
    assert err(f, torch.relu(Fn.conv2d(x.double(), ws[0].double(), bs[0].double(), padding=1))) < 2.5e-7
    # the hole this closes: the same data through the fixed 2^8 scale
    fixed = _hip.conv3x3_c64_split16(_hip.to_split16(x), Ws[0], bs[0], True, out_f32=True)
    e_fixed = err(fixed, torch.relu(Fn.conv2d(x.double(), ws[0].double(), bs[0].double(), padding=1)))
    if scale == 1.0:
        assert e_fixed < 2.5e-7
    elif scale <= 1e-4:
        assert e_fixed > (3e-7 if scale == 1e-4 else 1e-5), e_fixed
    elif scale == 1e3:
        assert not math.isfinite(e_fixed)


@pytest.mark.parametrize("n,H,W,n_layers,data_ranges", [(8, 128, 128, 13, True),      # the reference's usage: one measurement, FFDNet's 13 layers, ONE tile per CU
                                                        (3, 128, 128, 5, True),       # 96 tiles: every image's tiles straddle XCDs
                                                        (5, 64, 96, 4, False),        # 60 tiles (not a multiple of 8), fixed exponents
                                                        (1, 100, 76, 2, True),        # ragged edges, 21 tiles, SimpleCNN's two layers
                                                        (7, 16, 32, 3, True),         # one tile per image: no neighbours at all
                                                        (32, 128, 128, 13, True),     # a slice of a batch: FOUR tiles per workgroup
                                                        (19, 128, 128, 4, False),     # 608 tiles: workgroups with two and with three tiles
                                                        (2, 256, 512, 3, True),       # 512 tiles, a workgroup's own tiles are vertical neighbours
                                                        (4, 64, 64, 13, True),        # 32 tiles, 1 MB per activation: EVERYTHING stays in the L2s for 13 layers,
                                                        (6, 48, 80, 12, False)])      # 54 tiles dealt round-robin over the XCDs: a stale line would be read
def test_split16_stack_is_bit_identical_to_single_launches(n, H, W, n_layers, data_ranges):
    """A run of 64->64 layers as ONE launch (deqsci_conv3x3_c64_split16_stack: the persistent workgroups walk their tiles layer after
    layer, a tile waiting for the progress words of the tiles it reads; activations written through and fetched at agent scope) against
    the same layers as single launches:
    the same bits, launch after launch (the progress words count on: five launches on the same words, through the 32-bit wrap), with images that straddle XCDs,
    ragged edges, no-bias and no-ReLU layers, measured and fixed ranges, and batches small enough to live in the L2s of the eight XCDs for
    the whole launch (the L2s do not snoop one another: a tile's neighbours on another XCD are read with agent-scope loads - a stale line
    of the buffer as it was two layers ago would show here).  Both ping-pong buffers are poisoned before every launch - a tile that ran
    ahead of a neighbour would read the poison."""
    g = torch.Generator(device=DEV).manual_seed(100 * n + n_layers)
    x = (torch.relu(torch.randn(n, 64, H, W, device=DEV, generator=g)) * torch.logspace(0, -2, n, device=DEV).view(n, 1, 1, 1))
    x = x.contiguous(memory_format=torch.channels_last)
    ws = [torch.randn(64, 64, 3, 3, device=DEV, generator=g) * 0.06 for _ in range(n_layers)]
    bs = [None if i == 1 else torch.randn(64, device=DEV, generator=g) * 0.1 for i in range(n_layers)]
    relus = [i != n_layers - 2 for i in range(n_layers)]
    Ws = [_hip.Split16Weights(w) for w in ws]
    rng = torch.zeros(n_layers + 1, n, device=DEV) if data_ranges else None
    if data_ranges:
        _hip.absmax(x, rng[0])
    h0 = _hip.to_split16(x, rng=None if rng is None else rng[0])
    h = h0
    for i in range(n_layers):                                   # the single launches (measuring first, where the ranges follow the data)
        if data_ranges:
            _hip.conv3x3_c64_split16(h, Ws[i], bs[i], relus[i], track=rng[i + 1])
        h = _hip.conv3x3_c64_split16(h, Ws[i], bs[i], relus[i], out_rng=None if rng is None else rng[i + 1])
    want = h.t.clone()
    assert bool(torch.isfinite(want).all())
    stack = _hip.Split16Stack(list(zip(Ws, bs, relus)), DEV)
    flags, bufs = stack.flags(n, H, W), stack.state(n, H, W)
    flags.view(-1, 32)[:-1, 0] = -20                            # the words count on for ever: start them 20 below the 32-bit wrap
    for rep in range(5):
        for b in bufs:
            b.t.fill_(float("nan"))
        out = _hip.conv3x3_c64_split16_stack(h0, stack, rng, per_launch=n)
        assert out is bufs[(n_layers - 1) % 2]
        assert torch.equal(out.t, want), (rep, float((out.t.float() - want.float()).abs().max()))
        assert out.exponents() == h.exponents()
    fl = flags.cpu().view(-1, 32)[:, 0]
    assert int(fl[-1]) == 0 and bool((fl[:-1] == 5 * n_layers - 20).all())      # every tile: five launches of n_layers layers, no time-out
    assert not stack.timed_out()


@pytest.mark.parametrize("n,H,W,per", [(16, 128, 128, 8), (20, 128, 128, 8), (64, 128, 128, None), (40, 128, 128, None), (7, 100, 76, 3)])
def test_split16_stack_slices_a_batch(n, H, W, per):
    """A batch goes out as slices, one stack launch after the other - 2 x 8, 8 + 8 + 4 (two launch shapes, each with its own progress
    words), and with the default policy 2 x 32 and 32 + 8 (slices that fit the Infinity Cache, whole tiles per CU:
    _hip.split16_stack_per_launch): the
    same bits as the per-layer launches over the whole batch, per-image ranges read through the slice's offset into the slot table."""
    n_layers = 4
    g = torch.Generator(device=DEV).manual_seed(n)
    x = (torch.relu(torch.randn(n, 64, H, W, device=DEV, generator=g)) * torch.logspace(0, -3, n, device=DEV).view(n, 1, 1, 1))
    x = x.contiguous(memory_format=torch.channels_last)
    Ws = [_hip.Split16Weights(torch.randn(64, 64, 3, 3, device=DEV, generator=g) * 0.06) for _ in range(n_layers)]
    bs = [torch.randn(64, device=DEV, generator=g) * 0.1 for _ in range(n_layers)]
    rng = torch.zeros(n_layers + 1, n, device=DEV)
    _hip.absmax(x, rng[0])
    h0 = _hip.to_split16(x, rng=rng[0])
    h = h0
    for i in range(n_layers):
        _hip.conv3x3_c64_split16(h, Ws[i], bs[i], True, track=rng[i + 1])
        h = _hip.conv3x3_c64_split16(h, Ws[i], bs[i], True, out_rng=rng[i + 1])
    stack = _hip.Split16Stack([(w, b, True) for w, b in zip(Ws, bs)], DEV)
    assert _hip.split16_stack_per_launch(64, 128, 128) == 32 and _hip.split16_stack_per_launch(40, 128, 128, cus=256) == 32
    assert _hip.split16_stack_per_launch(8, 128, 128) == 8 and _hip.split16_stack_per_launch(8, 256, 256) == 8 and _hip.split16_stack_per_launch(3, 2048, 2048) == 1
    assert _hip.split16_stack_per_launch(64, 128, 96, cus=256) == 32          # 42 images fit 128 MiB; 32 x 24 tiles are 3 per CU
    assert _hip.split16_stack_per_launch(64, 100, 76, cus=256) == 64          # 21 tiles per image: no multiple within the 68 that fit
    for rep in range(3):
        for b in stack.state(n, H, W):
            b.t.fill_(float("nan"))
        out = _hip.conv3x3_c64_split16_stack(h0, stack, rng, per_launch=per)
        assert torch.equal(out.t, h.t), rep
    assert not stack.timed_out()


def test_split16_stack_errors():
    """Aliased buffers, a missing odd buffer, missing words, ranges of the wrong shape or stride."""
    g = torch.Generator(device=DEV).manual_seed(3)
    Ws = [_hip.Split16Weights(torch.randn(64, 64, 3, 3, device=DEV, generator=g) * 0.05) for _ in range(2)]
    stack = _hip.Split16Stack([(w, None, True) for w in Ws], DEV)
    small = _hip.Sp16.empty(2, 16, 32, DEV)
    small.t.zero_()
    with pytest.raises(_hip.DeqsciHipError, match="ranges"):
        _hip.conv3x3_c64_split16_stack(small, stack, torch.zeros(2, 2, device=DEV))
    with pytest.raises(_hip.DeqsciHipError, match="per_launch"):
        _hip.conv3x3_c64_split16_stack(small, stack, per_launch=0)
    lib = _hip.load()
    flags, bufs = stack.flags(2, 16, 32), stack.state(2, 16, 32)
    rng = torch.zeros(3, 2, device=DEV)
    args = lambda x, y0, y1, fl, r=None, rs=2: (x, y0, y1, stack.table.data_ptr(), 2, 2, 16, 32, r, rs, 8, 8, fl, None, None, None)   # noqa: E731
    assert lib.deqsci_conv3x3_c64_split16_stack(*args(small.t.data_ptr(), small.t.data_ptr(), bufs[1].t.data_ptr(), flags.data_ptr())) == -4
    assert lib.deqsci_conv3x3_c64_split16_stack(*args(small.t.data_ptr(), bufs[0].t.data_ptr(), None, flags.data_ptr())) == -1
    assert lib.deqsci_conv3x3_c64_split16_stack(*args(small.t.data_ptr(), bufs[0].t.data_ptr(), bufs[1].t.data_ptr(), None)) == -1
    assert lib.deqsci_conv3x3_c64_split16_stack(*args(small.t.data_ptr(), bufs[0].t.data_ptr(), bufs[1].t.data_ptr(), flags.data_ptr(), rng.data_ptr(), 1)) == -2


def _same(got, want, stack_kernel):
    """The stack launch on the split-fp16 DIRECT kernel ("s16") is the per-layer launches' arithmetic: the same bits.  On the Winograd
    kernel ("w16", the default) it is another arithmetic of the same accuracy (1.2-1.5e-7 per layer against float64, like the direct
    kernel's 1.6e-7): equal to a few 1e-6 over the <= 10 iterations these tests run."""
    if stack_kernel == "s16":
        return bool(torch.equal(got, want))
    return rel_l2(got.cpu().numpy(), want.cpu().numpy()) < 2e-5


@pytest.mark.parametrize("stack_kernel", ["w16", "s16"])
def test_engine_two_measurements_in_one_stack_launch(stack_kernel):
    """Two measurements per call = 16 images of 128 x 128 = two tiles per workgroup in one stack launch per f-call: the per-layer
    launches' result (s16: bit for bit) and - ranges are per image - BIT-identical to the two measurements reconstructed alone."""
    d = _clip("traffic_cacti.mat")
    Phi = d["mask"][None].to(DEV)
    y = d["meas"].permute(2, 0, 1)[1:3].contiguous().to(DEV)
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 8)[0].nonlinear_op
    want = DEQSCIEngine(net, max_iter=8, use_graph=False, stack=False).reconstruct(y, Phi)
    eng = DEQSCIEngine(net, max_iter=8, use_graph=False, stack_kernel=stack_kernel)
    got = eng.reconstruct(y, Phi)
    assert eng.last_info["stack_launches"] == eng.last_info["f_calls"] - 1 and eng.last_info["stack_timeouts"] == 0
    assert _same(got, want, stack_kernel)
    one = DEQSCIEngine(net, max_iter=8, use_graph=False, stack_kernel=stack_kernel)
    assert torch.equal(one.reconstruct(y[1:2], Phi), got[1:2])


@pytest.mark.parametrize("stack_kernel", ["w16", "s16"])
@pytest.mark.parametrize("bsz", [1, 3])
def test_split16_stack_timeout_is_reported_and_the_engine_falls_back(bsz, stack_kernel):
    """A stack launch waits for its own workgroups only - all resident when the device is ours.  When they are not (another process on the
    device's CUs) a wait gives up after a quarter of a second instead of hanging: the launch says so in its words, later launches on the
    same words do not wait at all, and the engine redoes the call with a launch per layer and stays there.  Simulated by setting one
    tile's progress word back: its neighbours can never see it reach their target.  One measurement per call (a tile per workgroup: the
    wait behind every layer) and three (three tiles per workgroup: the poll in the shadow of stage 2, then the same wait)."""
    d = _clip("traffic_cacti.mat")
    Phi, y = d["mask"][None].to(DEV), d["meas"].permute(2, 0, 1)[2:2 + bsz].contiguous().to(DEV)
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 4)[0].nonlinear_op
    want = DEQSCIEngine(net, max_iter=4, use_graph=False, stack=False).reconstruct(y, Phi)
    eng = DEQSCIEngine(net, max_iter=4, use_graph=False, stack_kernel=stack_kernel)
    ordinary = eng.reconstruct(y, Phi).clone()
    assert _same(ordinary, want, stack_kernel) and eng.last_info["stack_launches"] == eng.last_info["f_calls"] - 1 > 0
    assert eng.last_info["stack_timeouts"] == 0 and eng.last_info["denoiser_path"] == stack_kernel + " stack launch"
    # (round 6) behind a time-out the engine keeps the KERNEL and only drops the stack launch: one launch per layer of the same
    # Winograd kernel is bit-identical to its stack launch, so the calls behind a time-out return the bits of ordinary calls (under
    # stack_kernel="s16" that was always so: `want`)
    want = ordinary
    stack = (eng.den._wstacks if stack_kernel == "w16" else eng.den._stacks)[1]
    flags = stack.flags(8 * bsz, 128, 128)
    flags[32 * 5] -= 1000                                       # tile 5: the FIRST tile of its workgroup, which takes the words' common base from
                                                                # it (any other tile's word is simply rewritten when the tile is finished)
    t0 = time.time()
    with pytest.warns(RuntimeWarning, match="stack launch timed out"):
        got = eng.reconstruct(y, Phi)
    assert time.time() - t0 < 30.0
    assert torch.equal(got, want) and eng.den.stack is False and eng.last_info["stack_launches"] == 0 and eng.last_info["stack_timeouts"] == 1
    assert not stack.timed_out()                                # read and rearmed by the engine
    assert eng.last_info["denoiser_path"] == ("w16 per layer (behind a stack time-out)" if stack_kernel == "w16" else "per layer")
    assert torch.equal(eng.reconstruct(y, Phi), want) and eng.last_info["stack_timeouts"] == 0      # ... and stays on per-layer launches for a while,
    eng._stack_off_for = 1                                      # ... then tries the stack launch again (STACK_RETRY_CALLS calls later)
    again = eng.reconstruct(y, Phi)
    assert eng.den.stack is True and eng.last_info["stack_launches"] == eng.last_info["f_calls"] - 1 and eng.stack_timeouts_total == 1
    assert torch.equal(again, want) and eng.last_info["denoiser_path"] == stack_kernel + " stack launch"
    # the launchers' own API says so too: called directly (not from the engine, not under capture) a timed-out launch RAISES
    n = 8 * bsz
    x = torch.relu(torch.randn(n, 64, 128, 128, device=DEV)).contiguous(memory_format=torch.channels_last)
    if stack_kernel == "w16":
        xin, launch = _hip.P32.from_nchw(x), _hip.conv3x3_c64_wino16_stack
    else:
        xin, launch = _hip.to_split16(x), _hip.conv3x3_c64_split16_stack
    launch(xin, stack)                                          # fine
    stack.flags(n, 128, 128)[32 * 5] -= 1000
    with pytest.raises(_hip.DeqsciHipError, match="timed out"):
        launch(xin, stack)
    launch(xin, stack)                                          # (rearmed by the check)


@pytest.mark.parametrize("kind", ["ffdnet", "ffdnet-s16", "SimpleCNN", "SimpleCNN-128"])
def test_engine_stack_launch_matches_per_layer_launches(kind):
    """One measurement per call - the reference's usage (test_ffdnet.sh: batch 1) - takes the stack launch from the second f-call on (the
    first measures the ranges layer by layer): the reconstruction is bit-identical to the engine with stack=False, eagerly and as a
    replayed hipGraph, and last_info says how many stack launches ran.  SimpleCNN works at full resolution: 8 frames of 256 x 256 are four
    tiles per workgroup, a 128 x 128 crop of the measurement is one (its two 64->64 layers as one launch either way)."""
    d = _clip("traffic_cacti.mat")
    Phi, y = d["mask"][None].to(DEV), d["meas"][None, ..., 1].contiguous().to(DEV)
    sk = "w16" if kind == "ffdnet" else "s16"                   # (SimpleCNN under "w16" runs its two layers as Winograd launches and never as a stack:
                                                                #  its stack launch is the direct kernel's - stack_kernel="s16")
    if kind.startswith("ffdnet"):
        net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 10)[0].nonlinear_op
    else:
        net = build_pipeline("SimpleCNN", checkpoint.shipped("cnn"), 10)[0].nonlinear_op
    if kind == "SimpleCNN-128":
        Phi, y = Phi[:, 64:192, 100:228].contiguous(), y[:, 64:192, 100:228].contiguous()
    ref = DEQSCIEngine(net, max_iter=10, use_graph=False, stack=False, stack_kernel=sk)
    want = ref.reconstruct(y, Phi)
    assert ref.last_info["stack_launches"] == 0
    eng = DEQSCIEngine(net, max_iter=10, use_graph=False, stack_kernel=sk)
    got = eng.reconstruct(y, Phi)
    assert eng.last_info["stack_launches"] == (eng.last_info["f_calls"] - 1 if kind.startswith("ffdnet") else 0)     # (SimpleCNN's run is two layers: below STACK_MIN_LAYERS)
    assert eng.last_info["stack_timeouts"] == 0
    assert _same(got, want, "s16" if kind != "ffdnet" else "w16")
    want = got                                                  # (what the graph replay has to reproduce bit for bit)
    if not kind.startswith("ffdnet"):                           # ... but its two layers as one launch are the same bits too
        eng.den.STACK_MIN_LAYERS = 2
        eng.den.prepare(16, DEV)
        assert torch.equal(eng.reconstruct(y, Phi), want) and eng.last_info["stack_launches"] == eng.last_info["f_calls"] - 1
    gr = DEQSCIEngine(net, max_iter=10, use_graph=True, stack_kernel=sk)
    for _ in range(3):                                          # eager warm-up, capture, replay
        out = gr.reconstruct(y, Phi)
    assert gr.last_info["graph"] and torch.equal(out, want)


def _denoiser_error_vs_float64(net, z1, call=3):
    """rel-L2 error of one f-call of the engine's denoiser against the SAME folded layers evaluated in float64 (as
    test_denoiser_rounding_along_the_loop does): {"default": the shipped path, measuring its ranges on this input, "f22": the all-fp32
    Winograd F(2x2,3x3) path, "miopen": the folded layers on MIOpen's fp32 convolutions}."""
    import torch.nn.functional as Fn
    eng = DEQSCIEngine(net, max_iter=8, use_graph=False)
    den = eng.den
    den.prepare(16, DEV)
    bsz, B, H, W = z1.shape
    x = z1.view(bsz * B, 1, H, W)
    if den.tag == "ffdnet":
        sig = den.sigma_table[call:call + 1]
        h = torch.cat((sig.double().view(1, 1, 1, 1).expand(bsz * B, 1, H // 2, W // 2), Fn.pixel_unshuffle(x.double(), 2)), 1)
    else:
        h = x.double()
    for w, b, relu in den.fast:
        h = Fn.conv2d(h, w.double(), None if b is None else b.double(), padding=1)
        h = torch.relu(h) if relu else h
    ref = (Fn.pixel_shuffle(h, 2) if den.tag == "ffdnet" else h).reshape(z1.shape)
    e = lambda t: float((t.double() - ref).norm() / ref.norm())   # noqa: E731
    seen = []
    _hip.CONV64_EVENT_HOOK = lambda k, n, hh, ww: seen.append(k)
    try:
        got = den.run(z1, call, calibrate=True)[0]
        got2 = den.run(z1, call)[0]                           # the ranges stay: a second call without measuring is the same call
    finally:
        _hip.CONV64_EVENT_HOOK = None
    # (the measuring call runs the direct kernel layer by layer; FFDNet's later calls take the Winograd stack launch - the hook here does
    # not -: the same call to within either's own distance from the float64 network, bit for bit where the run keeps its per-layer launches)
    # (SimpleCNN: the measuring call on the direct kernel, later calls on Winograd launches; a plugin stack keeps the direct kernel)
    assert set(seen) == ({"s16", "w16"} if den.tag == "denoiser" and den.plain_head_w is not None else {"s16"})
    assert torch.equal(got, got2) if "w16" not in seen and den.tag != "ffdnet" else (rel_l2(got.cpu().numpy(), got2.cpu().numpy()) < 2.0 * e(got) and e(got2) < 1.25 * e(got))
    err = {"default": max(e(got), e(got2))}
    den.conv64 = den._policy = "f22"
    err["f22"] = e(den.run(z1, call)[0])
    mi = DEQSCIEngine(net, max_iter=8, use_graph=False, winograd=False).den
    mi.prepare(16, DEV)
    err["miopen"] = e(mi.run(z1, call)[0])
    print({k: "%.2e" % v for k, v in err.items()})
    return err


def test_config3_synthetic_batch_vs_oracle():
    """BASELINE config 3 - what bench.py times: the synthetic batch of SURVEY 8(d) C3 (Bernoulli(0.5) masks, x ~ U[0,1), y = Phi x, one mask
    per measurement, seed (1234, i)), 256 x 256 x 8, FFDNet - checked directly: measurements 0..2 as ONE engine batch (24 images of 128 x 128:
    the split-fp16 kernel with its ranges measured over the batch), and_maxiters=10 (the chaos-free horizon), against the CPU
    oracle run measurement by measurement; and the batch result equals the one-by-one results BIT FOR BIT: nothing couples the
    measurements of a batch (alpha, residual and - the ranges of the split-fp16 activations being per image - the denoiser's scales are
    all per measurement), so a measurement's reconstruction does not depend on how a batch is composed or sharded over GPUs."""
    import bench
    y, Phi, _ = bench.make_batch(0, 3, 256, 256, 8, 1234, torch.device(DEV))
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 10)[0].nonlinear_op
    seen = {}

    def hook(k, n, h, w, layers=1):                            # (with the `layers` keyword: stack launches are reported too)
        seen[(k, n, layers)] = seen.get((k, n, layers), 0) + 1
    _hip.CONV64_EVENT_HOOK = hook
    try:
        eng = DEQSCIEngine(net, max_iter=10, use_graph=False)
        rec = eng.reconstruct(y, Phi)
    finally:
        _hip.CONV64_EVENT_HOOK = None
    # f-call 0 runs the direct kernel layer by layer (13 writing launches; its measuring launches are not reported); f-calls 1 .. 10 are ONE stack launch each
    assert seen == {("s16", 24, 1): 13, ("w16stack", 24, 13): 10}, seen
    assert eng.last_info["stack_launches"] == 10 and eng.last_info["stack_timeouts"] == 0
    assert eng.last_info["f_calls"] == 11 and 0 < eng.last_info["res"] < 1
    yc, Pc = y.cpu(), Phi.cpu()
    for i in range(3):
        Ps = orc.phi_sum(Pc[i:i + 1])
        want, wres = orc.deq_forward(orc.ProxGradSCI("ffdnet"), orc.andersonexp, yc[i:i + 1], Pc[i:i + 1], Ps, orc.initial_point(yc[i:i + 1], Pc[i:i + 1]),
                                     m=5, beta=1.0, lam=1e-2, max_iter=10, tol=1e-5)
        assert rel_l2(rec[i:i + 1].cpu().numpy(), want.numpy()) < 1e-4
        assert abs(eng.last_info["res_per_sample"][i] / wres - 1) < 1e-3
        one = DEQSCIEngine(net, max_iter=10, use_graph=False).reconstruct(y[i:i + 1], Phi[i:i + 1])
        assert torch.equal(one, rec[i:i + 1])
    # ... also when the measurements of a batch are decades apart in scale (each keeps its own ranges)
    sc = torch.tensor([1.0, 1e-3, 30.0], device=DEV).view(3, 1, 1)
    mixed = DEQSCIEngine(net, max_iter=10, use_graph=False).reconstruct(y * sc, Phi)
    for i in range(3):
        one = DEQSCIEngine(net, max_iter=10, use_graph=False).reconstruct((y * sc)[i:i + 1].contiguous(), Phi[i:i + 1])
        assert torch.equal(one, mixed[i:i + 1])


@pytest.mark.parametrize("anderson_arith,bsz,iterator", [("reference", 8, "anderson"), ("float64", 12, "anderson"), ("reference", 8, "picard")])
def test_grouped_reconstruction_is_bit_identical_to_one_stream(anderson_arith, bsz, iterator):
    """DEQSCIEngine(groups=2) - an option, not the default: it measured between -4 % and +1.3 % - reconstructs a batch of at least two stack
    slices (8 measurements of 256 x 256 x 8 = 2 x 32 images) as two half batches on two streams, issued alternately up to every stack
    launch, their stack launches chained by events (_reconstruct_grouped): what the device overlaps changes, no result does - every
    kernel of the path is per measurement.  Held here:
    the reconstruction, the residual of the whole batch (new_equilibrium_utils_yaping.py:184: folded from the samples' float64 norms in
    sample order, as K6's last block does) and the per-sample residuals are BIT-identical to groups=1; a ragged split (12 measurements =
    3 slices: 8 + 4) likewise; extra_call rides along; a second call through the same engine reuses the halves; and a batch of less
    than two slices is not grouped."""
    import bench
    y, Phi, _ = bench.make_batch(0, bsz, 256, 256, 8, 1234, torch.device(DEV))
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 9)[0].nonlinear_op
    kw = dict(max_iter=9, use_graph=False, anderson_arith=anderson_arith, iterator=iterator, extra_call=(bsz == 12))
    one = DEQSCIEngine(net, groups=1, **kw)
    want = one.reconstruct(y, Phi)
    assert "groups" not in one.last_info
    assert DEQSCIEngine(net, **kw).groups == 1
    two = DEQSCIEngine(net, groups=2, **kw)
    for rep in range(2):
        got = two.reconstruct(y, Phi)
        assert two.last_info["groups"] == ([[0, 4], [4, 8]] if bsz == 8 else [[0, 8], [8, 12]])
        assert torch.equal(got, want)
        for k in ("res", "res_per_sample", "iterations", "f_calls", "stack_timeouts"):
            assert two.last_info[k] == one.last_info[k], (k, two.last_info[k], one.last_info[k])
        assert two.last_info["stack_launches"] == one.last_info["stack_launches"] > 0
        assert two.last_info["act_ranges"] == one.last_info["act_ranges"]
    small = two.reconstruct(y[:5], Phi[:5])                    # 40 images: one slice of 32 + a remainder - not two slices
    assert "groups" not in two.last_info and torch.equal(small, want[:5])
    # a shared mask and a caller's Phi_sum / initial point go to both halves
    if bsz == 8 and iterator == "anderson":
        Ps = deqsci_amd.phi_sum(Phi[:1])
        x0 = deqsci_amd.initial_point(y, Phi[:1].expand(bsz, -1, -1, -1).contiguous(), None, None)
        a = two.reconstruct(y, Phi[:1], Phi_sum=Ps, initial_point=x0)
        assert two.last_info["groups"] == [[0, 4], [4, 8]]
        assert torch.equal(a, one.reconstruct(y, Phi[:1], Phi_sum=Ps, initial_point=x0))


def test_ranges_are_measured_by_the_first_split16_call():
    """The ranges of the split-fp16 activations are measured by the first f-call that takes that path - f-call 0, unless the policy runs
    its first K f-calls on another kernel (conv64_f22_calls=K): then f-call K measures (a round-4 bug: it used unmeasured slots and the
    head's scale, taken from sigma alone, overflowed on the image).  Either way the result is finite, equals the all-default run to
    rounding, and last_info reports one range per layer of the stack."""
    d = _clip("traffic_cacti.mat")
    Phi, y = d["mask"][None].to(DEV), d["meas"][None, ..., 0].contiguous().to(DEV)
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 8)[0].nonlinear_op
    a = DEQSCIEngine(net, max_iter=8, use_graph=False)
    ra = a.reconstruct(y, Phi)
    rng_a = a.last_info["act_ranges"]
    assert len(rng_a) == 16 and all(v > 0 for v in rng_a[:15]) and 0.5 < rng_a[0] < 2.0          # image, 14 activations (the tail's output is fp32)
    assert all(_hip.act_exp(v) in range(4, 14) for v in rng_a[:15])
    b = DEQSCIEngine(net, max_iter=8, use_graph=False, conv64_f22_calls=3)
    rb = b.reconstruct(y, Phi)
    rng_b = b.last_info["act_ranges"]
    assert bool(torch.isfinite(rb).all()) and all(v > 0 for v in rng_b[:15])
    assert rel_l2(rb.cpu().numpy(), ra.cpu().numpy()) < 2e-5
    fixed = DEQSCIEngine(net, max_iter=8, use_graph=False, act_range="fixed")
    rf = fixed.reconstruct(y, Phi)
    assert rel_l2(rf.cpu().numpy(), ra.cpu().numpy()) < 2e-5 and fixed.last_info["act_ranges"] is None
    assert fixed.last_info["stack_launches"] == fixed.last_info["f_calls"]     # (nothing to measure: every f-call takes the stack launch)
    rf_lay = DEQSCIEngine(net, max_iter=8, use_graph=False, act_range="fixed", stack=False).reconstruct(y, Phi)
    assert rel_l2(rf_lay.cpu().numpy(), rf.cpu().numpy()) < 2e-5                 # (the stack launch runs the Winograd kernel: another arithmetic)
    assert torch.equal(DEQSCIEngine(net, max_iter=8, use_graph=False, act_range="fixed", stack_kernel="s16").reconstruct(y, Phi), rf_lay)
    # a second input through the same engine is measured afresh: 1000 x larger measurements, 1000 x larger image range
    a.reconstruct(y * 1000.0, Phi)
    assert 500 < a.last_info["act_ranges"][0] / rng_a[0] < 2000


def _dncnn17(scale):
    from deqsci_amd.networks import DnCNN
    torch.manual_seed(7)
    net = DnCNN(1, num_of_layers=17, lip=0.0, no_bn=False, tag="denoiser")
    for mod in net.dncnn:
        if isinstance(mod, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(mod.weight, nonlinearity="relu")
        elif isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.uniform_(0.7, 1.3)
            mod.bias.data.normal_(0, 0.05).mul_(scale)          # (the affine parts scale with the input: the stack stays homogeneous)
            mod.running_mean.normal_(0, 0.05).mul_(scale)
            mod.running_var.uniform_(0.8, 1.2)
    return net.to(DEV).eval()


@pytest.mark.parametrize("scale", [1e-2, 1e-4, 1e-6, 30.0])
def test_plugin_stack_dncnn17_follows_the_data_scale(scale):
    """A user plugin on the default engine: a DnCNN-17-style stack (conv 1->64 + ReLU, 15 x [conv 64->64 + BatchNorm + ReLU], conv 64->1;
    networks/provable/model/SimpleCNN_models.py:6-61 with num_of_layers=17), seeded weights, whose 64->64 layers _Denoiser sends through
    the split-fp16 kernels.  (a) SCALE-FREE like fp32: the f-call's error against the float64 stack is the same whether the input (and
    the stack's affine terms) are scaled by 1, 1e-2 ... 1e-6 or 30; (b) fp32-CLASS: within 1.25 x of the same stack on the fp32 kernels
    (Winograd F(2x2,3x3) / MIOpen's direct convolution; measured 1.13-1.15 x on this random stack, 0.5-0.85 x on FFDNet's own data)."""
    g = torch.Generator(device=DEV).manual_seed(2)
    z = torch.rand(2, 8, 128, 128, device=DEV, generator=g)
    base = _denoiser_error_vs_float64(_dncnn17(1.0), z)
    err = _denoiser_error_vs_float64(_dncnn17(scale), z * scale)
    assert err["default"] < 1.05 * base["default"], (scale, err, base)
    assert err["default"] < 1.25 * max(err["f22"], err["miopen"]) and base["default"] < 1.25 * max(base["f22"], base["miopen"]), (scale, err, base)


@pytest.mark.parametrize("kind,weights", [("ffdnet", "ffdnet_gray"), ("SimpleCNN", "cnn")])
@pytest.mark.parametrize("scale", [1e-3, 1e-6])
def test_shipped_denoisers_follow_the_data_scale(kind, weights, scale):
    """The same for the shipped denoisers' f-call on a scaled iterate (traffic measurement 0's x0 / 4, times `scale`).  SimpleCNN (no bias:
    homogeneous) is scale-free, (a) and (b) as above; FFDNet is not a homogeneous map (its sigma plane stays 60/255 whatever the image, so
    a small image is a small difference of sigma-sized activations - for the reference's fp32 too): (b) only."""
    d = _clip("traffic_cacti.mat")
    Phi, y = d["mask"][None].to(DEV), d["meas"][None, ..., 0].contiguous().to(DEV)
    z = (_hip.transpose(deqsci_amd.initial_point(y, Phi, None, None), _hip.LAYOUT_BHW) / 4.0).contiguous()
    net = build_pipeline(kind, checkpoint.shipped(weights), 8)[0].nonlinear_op
    base = _denoiser_error_vs_float64(net, z)
    err = _denoiser_error_vs_float64(net, (z * scale).contiguous())
    if kind == "SimpleCNN":
        assert err["default"] < 1.05 * base["default"], (scale, err, base)
    assert err["default"] < 1.25 * max(err["f22"], err["miopen"]) and base["default"] < 1.25 * max(base["f22"], base["miopen"]), (kind, scale, err, base)


@pytest.mark.parametrize("kind,weights,iters,crop", [("ffdnet", "ffdnet_gray", 30, 256), ("SimpleCNN", "cnn", 180, 128)])
@pytest.mark.parametrize("scale", [1e-2, 1e-4, 1e-6, 1e2])
def test_engine_scaled_measurements_vs_reference_golden(kind, weights, iters, crop, scale):
    """VERDICT r3 #1, end to end: the reference's own reconstructions of traffic measurement 0 with the measurement multiplied by
    1e-2 / 1e-4 / 1e-6 / 1e2 (tests/golden/make_golden.py g12: FFDNet and_maxiters=30 on the full frames, SimpleCNN and_maxiters=180 on
    the 128 x 128 crop) against the DEFAULT engine - split-fp16 64->64 layers, scales following the data.
      SimpleCNN (homogeneous: every activation shrinks with the input - the case the fixed 2^8 scale of round 3 lost: 1.4e-4 at 1e-6,
    profiles/r04_scaled_measurements_by_policy.txt): <= 1e-5 rel-L2 at every scale (measured 1.4e-7 ... 9e-7).
      FFDNet's sigma plane does not shrink with the image, so its activations never leave the fixed scale's range either; what the small
    scales show instead is CONDITIONING: the reconstruction is a small difference of sigma-sized quantities (res ~ 0.05, not converging),
    and every fp32 implementation - Winograd F(2x2,3x3), F(4x4,3x3), MIOpen's direct convolution - sits 1.6-3.6e-4 from the reference's
    run and as far from one another, after 10 iterations already.  The gate there is "no further from the reference than the all-fp32
    F(2x2,3x3) engine" and < 5e-4.  At 100 x the Gram matrix dwarfs lam = 1e-2 and the reference's fp32 bmm + sgesv lose digits (the
    large-N effect of DESIGN section 5): all of the build's policies agree to 4e-5 and sit 2e-3 from the reference's run together."""
    gold = np.load(os.path.join(GOLDEN, "e2e_scaled_measurements.npz"))
    d = _clip("traffic_cacti.mat")
    Phi = d["mask"][None, :crop, :crop].contiguous().to(DEV)
    y = (d["meas"][None, :crop, :crop, 0] * np.float32(scale)).contiguous().to(DEV)
    net = build_pipeline(kind, checkpoint.shipped(weights), iters)[0].nonlinear_op
    eng = DEQSCIEngine(net, max_iter=iters, use_graph=False)
    seen = set()
    _hip.CONV64_EVENT_HOOK = lambda k, n, h, w: seen.add(k)
    try:
        rec = eng.reconstruct(y, Phi)
    finally:
        _hip.CONV64_EVENT_HOOK = None
    assert seen == ({"s16"} if kind == "ffdnet" else {"s16", "w16"}) and eng.conv64_policy == "fast" and eng.last_info["conv64_fallback"] is None
    want = gold[f"{kind}_{iters}_s{scale:g}_rec"]
    e_def = rel_l2(rec.cpu().numpy(), want)
    if kind == "SimpleCNN":
        assert e_def < 1e-5
        assert abs(eng.last_info["res"] / float(gold[f"{kind}_{iters}_s{scale:g}_res"]) - 1) < 1e-2
        return
    rec22 = DEQSCIEngine(net, max_iter=iters, use_graph=False, conv64="f22").reconstruct(y, Phi)
    e_f22 = rel_l2(rec22.cpu().numpy(), want)
    print(kind, scale, "default vs reference %.2e, F(2x2,3x3) vs reference %.2e" % (e_def, e_f22))
    if scale < 1:
        assert e_def < 1.1 * e_f22 + 1e-5 and e_def < 5e-4
    else:
        # (round 6: the default arithmetic of alpha is the reference's - fp32 Gram, fp32 LU: at this scale IT loses the digits, and two
        #  denoiser arithmetics end 2e-3 apart like the reference from either; with the exact Gram they agree to 4e-5 as before)
        assert e_def < 3e-3 and e_f22 < 5e-3
        x64 = [DEQSCIEngine(net, max_iter=iters, use_graph=False, anderson_arith="float64", conv64=c).reconstruct(y, Phi).cpu().numpy() for c in ("auto", "f22")]
        assert rel_l2(x64[0], x64[1]) < 1e-4 and rel_l2(x64[0], want) < 3e-3
    assert abs(eng.last_info["res"] / float(gold[f"{kind}_{iters}_s{scale:g}_res"]) - 1) < 2e-2


@pytest.mark.parametrize("shape", [(2, 40, 24), (1, 256, 256), (3, 33, 70)])
def test_plain_edge_kernels_vs_torch(shape):
    """conv3x3 1->64 (+ReLU) planar -> channels_last and conv3x3 64->1 channels_last -> planar (SimpleCNN edges)."""
    import torch.nn.functional as Fn
    n, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(13)
    x = torch.randn(n, 1, H, W, device=DEV, generator=g)
    w1 = torch.randn(64, 1, 3, 3, device=DEV, generator=g) * 0.2
    want = torch.relu(Fn.conv2d(x.double(), w1.double(), padding=1))
    got = _hip.conv3x3_c1_to_64(x, _hip.pack_c1_to_64_weights(w1), relu=True)
    assert got.is_contiguous(memory_format=torch.channels_last)
    assert float((got.double() - want).norm() / want.norm()) < 1e-6
    got_lin = _hip.conv3x3_c1_to_64(x, _hip.pack_c1_to_64_weights(w1), relu=False)
    want_lin = Fn.conv2d(x.double(), w1.double(), padding=1)
    assert float((got_lin.double() - want_lin).norm() / want_lin.norm()) < 1e-6
    h = torch.randn(n, 64, H, W, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    w2 = torch.randn(1, 64, 3, 3, device=DEV, generator=g) * 0.05
    want2 = Fn.conv2d(h.double(), w2.double(), padding=1)
    got2 = _hip.conv3x3_c64_to_1(h, _hip.pack_c64_to_1_weights(w2))
    assert got2.shape == (n, 1, H, W) and float((got2.double() - want2).norm() / want2.norm()) < 1e-6
    b = torch.randn(64, device=DEV, generator=g)
    want3 = Fn.conv2d(torch.relu(h.double() + b.double().view(1, -1, 1, 1)), w2.double(), padding=1)
    got3 = _hip.conv3x3_c64_to_1(h, _hip.pack_c64_to_1_weights(w2), in_bias=b)
    assert float((got3.double() - want3).norm() / want3.norm()) < 1e-6
    # sp16 on the 64-channel side: the head writing it (fixed 2^8 scale; and following the data: measured, then written), the matrix-core tail
    sp = _hip.conv3x3_c1_to_64(x, _hip.pack_c1_to_64_weights(w1), relu=True, sp16=True)
    assert isinstance(sp, _hip.Sp16) and float((sp.to_nchw() - got).norm() / got.norm()) < 1e-7
    for scale in (1.0, 1e-6):
        slot = torch.zeros(n, device=DEV)
        _hip.conv3x3_c1_to_64(x * scale, _hip.pack_c1_to_64_weights(w1), relu=True, sp16=True, out_exp=0, track=slot)
        assert float((slot / (scale * want.abs().amax(dim=(1, 2, 3)).float()) - 1).abs().max()) < 1e-5
        sp = _hip.conv3x3_c1_to_64(x * scale, _hip.pack_c1_to_64_weights(w1), relu=True, sp16=True, out_rng=slot)
        assert float((sp.to_nchw().double() - scale * want).norm() / (scale * want).norm()) < 3e-7
    got2_mm = _hip.tail_split16(_hip.to_split16(h), _hip.TailSplit16Weights(w2))       # matrix-core form, COUT = 1
    assert got2_mm.shape == (n, 1, H, W) and float((got2_mm.double() - want2).norm() / want2.norm()) < 3e-7


@pytest.mark.parametrize("kind", ["SimpleCNN", "ffdnet"])
def test_training_backward_vs_reference_golden(kind):
    """DEQFixedPoint with a tape = the reference's training forward + implicit-differentiation backward hook
    (solve without tape, taped f, g = J^T g + grad solved with the same Anderson settings).  Reconstruction, loss, both
    residuals, the FFDNet sigma state and the gradient of every denoiser weight against the reference's own run
    (tests/golden/backward*.npz);
    the linear operators' backward passes are the HIP kernels (deqsci_amd/autograd.py)."""
    g = np.load(os.path.join(GOLDEN, "backward.npz" if kind == "SimpleCNN" else "backward_ffdnet.npz"))
    solver, _ = build_pipeline(kind, checkpoint.shipped("cnn" if kind == "SimpleCNN" else "ffdnet_gray"), 12)
    deq = deqsci_amd.DEQFixedPoint(solver, deqsci_amd.andersonexp, m=5, beta=1.0, lam=1e-2, max_iter=12, tol=1e-9)
    Phi, y, Ps, gt = G(g["Phi"]), G(g["y"]), G(g["Phi_sum"]), G(g["gt"])
    rec = deq(y, Phi, Ps, initial_point=deqsci_amd.initial_point(y, Phi, Ps, gt))
    assert rec.requires_grad
    loss = torch.nn.functional.mse_loss(rec, gt)
    solver.zero_grad()
    loss.backward()
    assert rel_l2(rec.detach().cpu().numpy(), g["rec"]) < 1e-4
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5 * float(g["loss"])
    assert abs(deq.forward_res - float(g["forward_res"])) < 1e-2 * float(g["forward_res"])
    assert abs(deq.backward_res - float(g["backward_res"])) < 1e-2 * float(g["backward_res"])
    for name, p in solver.named_parameters():
        assert rel_l2(p.grad.cpu().numpy(), g["grad." + name]) < (1e-4 if kind == "SimpleCNN" else 5e-4), name
    if kind == "ffdnet":
        assert np.array_equal(solver.noise_sigma.cpu().numpy(), g["sigma_after"])
    # the inference switch: no tape, engine path, same reconstruction
    rec2 = deq(y, Phi, Ps, initial_point=deqsci_amd.initial_point(y, Phi, Ps, gt), train_flag=False)
    assert not rec2.requires_grad and rel_l2(rec2.cpu().numpy(), g["rec"]) < 1e-4


def test_autograd_ops_are_adjoint_consistent():
    """backward of A is At, of At is A, of the GAP projection is the projection with y = 0 (autograd.gradcheck-style dot test)."""
    from deqsci_amd import autograd as ag
    gen = torch.Generator(device=DEV).manual_seed(3)
    Phi = (torch.rand(2, 20, 12, 8, device=DEV, generator=gen) < 0.5).float()
    Ps = deqsci_amd.phi_sum(Phi)
    x = torch.randn(2, 20, 12, 8, device=DEV, generator=gen, requires_grad=True)
    yv = torch.randn(2, 20, 12, device=DEV, generator=gen, requires_grad=True)
    u = torch.randn(2, 20, 12, device=DEV, generator=gen)
    v = torch.randn(2, 20, 12, 8, device=DEV, generator=gen)
    (gx,) = torch.autograd.grad(deqsci_amd.A_torch_(x, Phi), x, u)
    assert rel_l2(gx.cpu().numpy(), (u[..., None] * Phi).cpu().numpy()) < 1e-6
    (gy,) = torch.autograd.grad(deqsci_amd.At_torch_(yv, Phi), yv, v)
    assert rel_l2(gy.cpu().numpy(), (v * Phi).sum(3).cpu().numpy()) < 1e-6
    z1 = ag.gap_update(x, yv, Phi, Ps)
    gz, gyy = torch.autograd.grad(z1, (x, yv), v)
    xr = x.detach().clone().requires_grad_()
    yr = yv.detach().clone().requires_grad_()
    ref = xr + ((yr - (xr * Phi).sum(3)) / Ps)[..., None] * Phi
    rz, ry = torch.autograd.grad(ref, (xr, yr), v)
    assert rel_l2(z1.detach().cpu().numpy(), ref.detach().cpu().numpy()) < 1e-6
    assert rel_l2(gz.cpu().numpy(), rz.cpu().numpy()) < 1e-5 and rel_l2(gyy.cpu().numpy(), ry.cpu().numpy()) < 1e-5


# ----------------------------------------------------------------------------- round 2: remaining (a) rows and gates
def test_sci_operator_vs_oracle():
    """the LinearOperator surface (operators/operator.py:3-14) - forward, adjoint, gramian = adjoint(forward)."""
    Phi, Phie, x, z, y, Ps = make_case(2, 24, 20, 8, seed=41)
    op = deqsci_amd.SCIOperator(G(Phi))
    assert isinstance(op, deqsci_amd.LinearOperator)
    torch.testing.assert_close(op.forward(G(z)).cpu(), orc.sci_forward(z, Phie), rtol=1e-6, atol=2e-6)
    assert torch.equal(op.adjoint(G(y)).cpu(), orc.sci_adjoint(y, Phie))
    want = orc.sci_adjoint(orc.sci_forward(z, Phie), Phie)
    torch.testing.assert_close(op.gramian(G(z)).cpu(), want, rtol=1e-6, atol=2e-6)
    # <Phi x, y> == <x, Phi^T y>
    lhs = float((op.forward(G(x)).double() * G(y).double()).sum())
    rhs = float((G(x).double() * op.adjoint(G(y)).double()).sum())
    assert abs(lhs - rhs) < 1e-6 * abs(lhs)
    assert "Phi" in dict(op.named_buffers())


class _ToyPlugin(torch.nn.Module):
    """tests/golden/make_golden.py g11's toy plugin: 0.5 tanh(conv(x, w)), 2-D or 3-D by the weight's rank."""

    def __init__(self, tag, weight):
        super().__init__()
        self.tag = tag
        self.weight = torch.nn.Parameter(weight, requires_grad=False)

    def forward(self, x):
        conv = torch.nn.functional.conv3d if self.weight.dim() == 5 else torch.nn.functional.conv2d
        return 0.5 * torch.tanh(conv(x, self.weight, padding=1))


@pytest.mark.parametrize("tag", ["conv2d", "conv3d", "3d_denoiser"])
def test_plugin_tags_vs_reference_golden(tag):
    """the tags without shipped weights (equilibrium_solvers_yaping.py:402-407,421-423) through the
    solver's forward, the generic DEQ path and the engine, against the reference's own run on the same toy plugin."""
    g = np.load(os.path.join(GOLDEN, "plugin_tags.npz"))
    Phi, y, Ps = G(g["Phi"]), G(g["y"]), G(g["Phi_sum"])
    net = _ToyPlugin(tag, G(g[tag + "_w"])).to(DEV)
    solver = deqsci_amd.EquilibriumProxGradSCI(A=deqsci_amd.A_torch_, At=deqsci_amd.At_torch_, nonlinear_operator=net, eta=0.2)
    x0 = deqsci_amd.initial_point(y, Phi, Ps, None)
    with torch.no_grad():
        assert rel_l2(solver(x0, y, Phi, Ps).cpu().numpy(), g[tag + "_f_x0"]) < 1e-5
    for use_engine in (True, False):
        deq = deqsci_amd.DEQFixedPoint(solver, deqsci_amd.andersonexp, m=5, beta=1.0, lam=1e-2, max_iter=9, tol=1e-9)
        deq.use_engine = use_engine
        assert (deq._engine_for() is not None) == use_engine
        rec = deq.forward(y, Phi, Ps, initial_point=x0, train_flag=False)
        assert rel_l2(rec.cpu().numpy(), g[tag + "_rec"]) < 1e-4, use_engine
        assert abs(deq.forward_res - float(g[tag + "_res"])) < 1e-2 * float(g[tag + "_res"])


def test_realsn_simplecnn_vs_reference():
    """RealSN_SimpleCNN / rsn_cnn.ckpt (video_sci_proxgrad.py:181-183, test_rsn_cnn.sh) over all 8 shipped
    measurements, on the HIP conv kernels (1->64 stencil, two Winograd 64->64 layers, 64->1 stencil)."""
    from deqsci_amd.harness import SCITestDataset, test_solver_sci
    meta = _golden_meta("RealSN_SimpleCNN_anderson_10")
    _, deq = build_pipeline("RealSN_SimpleCNN", checkpoint.shipped("rsn_cnn"), 10)
    records = []
    avg, images = test_solver_sci(deq, test_dataloader=SCITestDataset(orc.DATA_DIR), save_img_path="", verbose=False,
                                  save_image=False, records=records)
    den = deq._engine_for().den
    assert den.fast is not None and den.plain_head_w is not None and den.plain_tail_w is not None
    assert [w is not None for w in den.wino] == [False, True, True, False]
    assert [r["id"] for r in records] == [m["id"] for m in meta["measurements"]]
    for r, m in zip(records, meta["measurements"]):
        assert abs(r["psnr"] - m["psnr"]) < 0.01, r["id"]
        assert abs(r["res"] - m["res"]) < 2e-2 * m["res"], r["id"]
    assert abs(avg - meta["avg_psnr"]) < 0.01
    recs = np.load(os.path.join(GOLDEN, "e2e_RealSN_SimpleCNN_anderson_10_rec.npz"))
    by_id = {r["id"]: r["rec"].numpy() for r in records}
    assert rel_l2(by_id["traffic_cacti.mat:0"], recs["traffic_m0"]) < 1e-4
    assert rel_l2(by_id["drop8_cacti.mat:0"], recs["drop8_m0"]) < 1e-4


def test_realsn_simplecnn_100_iters_script_default():
    """test_rsn_cnn.sh leaves --and_maxiters at its default of 100: traffic measurement 0 against the reference's run
    (SURVEY f-4 recorded 22.6831 dB / res 6.34e-4 for it)."""
    fn = os.path.join(GOLDEN, "e2e_RealSN_SimpleCNN_anderson_100.json")
    assert os.path.exists(fn), "committed golden missing (tests/golden/make_golden.py g9): this test must not skip"
    meta = [m for m in _golden_meta("RealSN_SimpleCNN_anderson_100")["measurements"] if m["id"] == "traffic_cacti.mat:0"][0]
    assert abs(meta["psnr"] - 22.6831) < 1e-3
    d = _clip("traffic_cacti.mat")
    Phi = d["mask"][None].to(DEV)
    y = d["meas"][None, ..., 0].contiguous().to(DEV)
    _, deq = build_pipeline("RealSN_SimpleCNN", checkpoint.shipped("rsn_cnn"), 100)
    Ps = deqsci_amd.phi_sum(Phi)
    rec = deq.forward(y, Phi, Ps, initial_point=deqsci_amd.initial_point(y, Phi, Ps, None), train_flag=False).cpu().numpy()
    want = np.load(os.path.join(GOLDEN, "e2e_RealSN_SimpleCNN_anderson_100_rec.npz"))["traffic_m0"]
    assert rel_l2(rec, want) < 1e-4
    assert abs(orc.psnr(rec, d["gt"][None, ..., :8].numpy()) - meta["psnr"]) < 0.01
    assert abs(deq.forward_res - meta["res"]) < 2e-2 * meta["res"]


@pytest.mark.parametrize("kind", ["SimpleCNN", "RealSN_SimpleCNN"])
def test_png_payloads_vs_reference(kind):
    """the float images the reference hands to cv2.imwrite (tensor_to_np, sci_equilibrium_training.py:19-21,
    185-187), first and last exported frame of every clip, to 1e-4 * 255; names and count as the reference's.  The uint8
    rounding inside cv2.imwrite stays unpinned (cv2 is not installed in the build container)."""
    from deqsci_amd.harness import SCITestDataset, test_solver_sci
    tag = f"{kind}_anderson_10"
    meta = _golden_meta(tag)
    png = np.load(os.path.join(GOLDEN, f"e2e_{tag}_png.npz"))
    _, deq = build_pipeline(kind, checkpoint.shipped("cnn" if kind == "SimpleCNN" else "rsn_cnn"), 10)
    _, images = test_solver_sci(deq, test_dataloader=SCITestDataset(orc.DATA_DIR), save_img_path="", verbose=False, save_image=False)
    assert len(images) == meta["n_png_payloads"] == 64
    assert sorted(png.files) == sorted(meta["png_payload_keys"])
    for k in png.files:
        assert images[k].shape == png[k].shape == (256, 256, 1)
        assert np.abs(images[k] - png[k]).max() < 1e-4 * 255, k


def _config2_reference():
    """The two reference ensembles of make_golden g10 (FFDNet, Anderson @180): the reference as it is (fp32 torch.bmm Gram) and
    with the Gram matrix of :178 computed exactly (25 runs per `traffic` measurement each, 9-10 on drop8 / runner8)."""
    with open(os.path.join(GOLDEN, "e2e_ffdnet_anderson_180_spread.json")) as fh:
        a = json.load(fh)
    with open(os.path.join(GOLDEN, "e2e_ffdnet_anderson_180_spread_gram64.json")) as fh:
        b = json.load(fh)
    assert len(a["measurements"]) == 8 and "avg_psnr_min" in a and len(b["measurements"]) == 8 and "avg_psnr_min" in b
    return a, b


def _widened(vals, pad):
    lo, hi = min(vals), max(vals)
    return lo - 0.25 * (hi - lo) - pad, hi + 0.25 * (hi - lo) + pad          # total factor 1.5, + the north_star tolerance


CONFIG2_SEEDS = 25          # runs per measurement on the build side (0 = unperturbed, 1..24 = x0 (1 + 1e-7 randn)); the reference ensembles have 9-10


def _se(v):
    return float(np.std(v, ddof=1) / np.sqrt(len(v)))


CONFIG2_GRAM_SHIFT = 0.020   # dB: what the reference's fp32 torch.bmm Gram adds to the six-measurement mean (profiles/r04_config2_anderson_arith_*:
                             # +0.018 / +0.022 +- 0.005 between fp32 and fp64 Gram, same code path, same denoiser, same seeds)


def _config2_ensembles(eng, chaotic_only=False):
    """-> [(measurement id, build PSNRs, build residuals, reference-as-it-is PSNRs, reference-exact-Gram PSNRs, reference residuals)], unperturbed PSNR by clip"""
    from deqsci_amd.harness import SCITestDataset, as_clip, psnr, scored_measurements
    a, b = _config2_reference()
    report, base_by_clip = [], {}
    for clip in (as_clip(c) for c in SCITestDataset(orc.DATA_DIR)):
        if chaotic_only and "traffic" not in clip["file"]:
            continue
        Phi = clip["mask"].to(DEV)[None].contiguous()
        for fi in scored_measurements(clip["file"], clip["meas"].shape[-1]):
            mid = f"{clip['file']}:{fi}"
            y = clip["meas"][..., fi].to(DEV)[None].contiguous()
            gt = clip["gt"][..., 8 * fi:8 * fi + 8].numpy()
            x0 = deqsci_amd.initial_point(y, Phi, None, None)
            ps, rs = [], []
            for seed in range(CONFIG2_SEEDS):                                 # 0 = unperturbed
                xs = x0 if seed == 0 else x0 * (1 + 1e-7 * torch.randn(x0.shape, generator=torch.Generator().manual_seed(seed))).to(DEV)
                rec = eng.reconstruct(y, Phi, initial_point=xs)
                assert eng.last_info["f_calls"] == 181
                ps.append(psnr(rec.clamp(0, 1).cpu().numpy()[0], gt))
                rs.append(eng.last_info["res"])
            base_by_clip.setdefault(clip["file"], []).append(ps[0])
            # (the as-is file carries one extra row, "gram_fp64": the unperturbed start with an exact Gram - it belongs to the other ensemble)
            va = {k: v for k, v in a["measurements"][mid]["variants"].items() if k != "gram_fp64"}
            vb = dict(b["measurements"][mid]["variants"])
            if "gram_fp64" in a["measurements"][mid]["variants"] and "g64_base" not in vb:
                vb["gram_fp64"] = a["measurements"][mid]["variants"]["gram_fp64"]
            ra = [v["psnr"] for v in va.values()]
            rb = [v["psnr"] for v in vb.values()]
            ea = [v["res"] for v in va.values()] + [v["res"] for v in vb.values()]
            report.append((mid, ps, rs, ra, rb, ea))
    return report, base_by_clip, (a, b)


def _pooled(chaotic, k):
    mean = float(np.mean([np.mean(c[k]) for c in chaotic]))                   # mean of the per-measurement ensemble means
    se = float(np.sqrt(sum(_se(c[k]) ** 2 for c in chaotic)) / len(chaotic))
    return mean, se


def test_config2_ffdnet_anderson_180_all_measurements():
    """BASELINE config 2 as stated (test_ffdnet.sh:1-7): FFDNet, Anderson, and_maxiters=180, every shipped measurement, on the DEFAULT
    engine.  The map is chaotic on the `traffic` clip (the reference moves by 4e-2 rel-L2 / 0.1-0.2 dB under a 1e-7 perturbation of x0), so
    one run against one run says nothing; the gate compares ENSEMBLES built by the same recipe - the unperturbed x0 and x0 (1 + 1e-7 randn),
    seeds 1..24 - with the reference's own (make_golden g10: the reference as it is, and with the Gram matrix of :178 in float64).

    What four rounds of measurements established ("Config 2"; profiles/r04_config2_*):
      * per measurement the ensemble mean is a property of the (measurement, ARITHMETIC) pair, far beyond its ensemble error: the reference's
        own two variants differ by up to 0.22 dB on one measurement (RMS 0.12 over the six), every implementation of the build likewise;
      * the six-measurement mean is tight within a family of equivalent arithmetics (12 members, SD 0.003 = sampling error) and does NOT move
        with the denoiser's rounding (float64 denoiser: 21.414; split-fp16 21.417; F(2x2,3x3) 21.417; MIOpen 21.410);
      * it DOES move with the precision of the Gram matrix: the reference's Anderson step emulated on the GPU around the same f - fp32
        torch.bmm Gram, fp32 LU - gives 21.430 +- 0.004 and the reference's per-measurement pattern (m2 at 21.53), the same code with a
        float64 Gram 21.408 / 21.412.  The reference's fp32 Gram error (~5e-6 at N = 2^19) is worth +0.02 dB at 180 iterations.
    `anderson_arith="float64"` (until round 6 DEQSCIEngine's own default) keeps the exact Gram (deviation 3): around round 4's direct split-fp16 kernels its pooled mean sat CONFIG2_GRAM_SHIFT
    below the reference's (21.417), around round 5's Winograd kernels it is 21.434 - the allowance stays, it is no longer used up
    (DESIGN section 5, item 6); the
    engine's `anderson_arith="reference"` (the next test) reproduces the reference's.  Criteria, all computed from the ensembles:
      * six-measurement mean: no more than 3 SE above the reference's (either variant), no more than CONFIG2_GRAM_SHIFT + 3 SE below;
      * per chaotic measurement: median PSNR / residual inside the hull of both reference ensembles widened by 1.5 (+ 0.01 dB / 1 %), no run
        further than 1.5 hull widths outside (25 runs each side; single measurements are arithmetic-dependent, see above);
      * the RMS over the six of (build mean - exact-Gram reference mean) no larger than the same RMS for the reference as it is;
      * well-conditioned measurements (drop8, runner8: reference bands of 2 and 25 mdB): ensemble mean within 0.01 dB (+ 3 SE); every run
        within HALF a reference hull width (+ 0.01 dB) of the reference's 9-10 run hull - the expected range of 25 draws is 1.3 x that
        of 9, i.e. 0.16 widths per side, and the 9-run hull itself is uncertain by about as much; residuals within 3 % of the reference's
        band (the residual of a run that has not converged to 1e-5 moves by 1-2 % under a 1e-7 perturbation on either side);
      * the harness average of the unperturbed run inside the hull of the two reference average bands."""
    solver, _ = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 180)
    assert DEQSCIEngine(solver.nonlinear_op).anderson_arith == "reference"       # (one default for every entry point, round 6: the next test's)
    eng = DEQSCIEngine(solver.nonlinear_op, iterator="anderson", m=5, beta=1.0, lam=1e-2, max_iter=180, tol=1e-5, anderson_arith="float64")
    assert eng.conv64 == "auto" and eng.conv64_f22_calls is None and eng.conv64_policy == "fast" and eng.anderson_arith == "float64"
    assert _hip.conv64_kernel_for(8, 128, 128, policy=eng.conv64_policy) == "s16"      # one measurement per call = 8 images of 128 x 128
    report, base_by_clip, (a, b) = _config2_ensembles(eng)
    chaotic = []
    for mid, ps, rs, ra, rb, ea in report:
        lo, hi = _widened(ra + rb, 0.01)
        rlo, rhi = _widened(ea, 0.0)
        se = float(np.hypot(_se(ps), _se(rb)))
        print("%-22s build mean %.4f +- %.4f median %.4f [%.4f, %.4f] | reference exact-Gram mean %.4f +- %.4f, fp32 mean %.4f, widened hull [%.4f, %.4f]"
              % (mid, np.mean(ps), _se(ps), np.median(ps), min(ps), max(ps), np.mean(rb), _se(rb), np.mean(ra), lo, hi))
        w = max(ra + rb) - min(ra + rb)
        if w < 0.05:
            assert abs(np.mean(ps) - np.mean(rb)) <= 0.01 + 3 * se, (mid, np.mean(ps), np.mean(rb), se)
            assert min(ra + rb) - 0.5 * w - 0.01 <= min(ps) and max(ps) <= max(ra + rb) + 0.5 * w + 0.01, (mid, min(ps), max(ps), min(ra + rb), max(ra + rb))
            assert all(rlo * 0.97 <= r <= rhi * 1.03 for r in rs), (mid, rs, rlo, rhi)
            continue
        chaotic.append((ps, ra, rb))
        assert lo <= np.median(ps) <= hi, (mid, np.median(ps), lo, hi)
        assert rlo * 0.99 <= np.median(rs) <= rhi * 1.01, (mid, np.median(rs), rlo, rhi)
        assert min(ra + rb) - 1.5 * w <= min(ps) and max(ps) <= max(ra + rb) + 1.5 * w, (mid, min(ps), max(ps), min(ra + rb), max(ra + rb))
    assert len(chaotic) == 6
    (mb, sb), (ma_, sa), (mx, sx) = _pooled(chaotic, 0), _pooled(chaotic, 1), _pooled(chaotic, 2)
    print("six chaotic measurements: build %.4f +- %.4f | reference exact Gram %.4f +- %.4f (diff %+.4f) | as it is %.4f +- %.4f (diff %+.4f); "
          "expected shift of the exact Gram -%.3f" % (mb, sb, mx, sx, mb - mx, ma_, sa, mb - ma_, CONFIG2_GRAM_SHIFT))
    for ref, sr in ((mx, sx), (ma_, sa)):
        se = float(np.hypot(sb, sr))
        assert -(CONFIG2_GRAM_SHIFT + 3 * se) <= mb - ref <= 3 * se, (mb, ref, se)
    rms = lambda k: float(np.sqrt(np.mean([(np.mean(c[k]) - np.mean(c[2])) ** 2 for c in chaotic])))   # noqa: E731
    print("per-measurement offsets from the exact-Gram reference, RMS: build %.4f dB, reference as it is %.4f dB" % (rms(0), rms(1)))
    assert rms(0) <= rms(1) + sb, (rms(0), rms(1))
    avg = float(np.mean([np.mean(v) for v in base_by_clip.values()]))         # test_solver_sci's average: mean over clips of the clip mean
    alo, ahi = _widened([a["avg_psnr_min"], a["avg_psnr_max"], b["avg_psnr_min"], b["avg_psnr_max"]], 0.01)
    print("harness average of the unperturbed run %.4f in [%.4f, %.4f]" % (avg, alo, ahi))
    assert alo <= avg <= ahi, (avg, alo, ahi)


def test_config2_with_the_references_anderson_arithmetic():
    """The other half of the config-2 account: with the reference's OWN arithmetic for alpha - `anderson_arith="reference"`, what the drop-in
    DEQFixedPoint runs: G G^T in fp32 in the summation ORDER of the reference's torch.bmm (16 interleaved FMA chains per entry, formed by
    csrc/anderson.hip's own kernels; its 2^15-step chains absorb the small products of the heavy-tailed residuals, so the diagonal comes out
    3-7e-6 too small), fp32 LU, solvers/new_equilibrium_utils_yaping.py:177-180 - the engine's 25-start ensembles of the six chaotic
    measurements reproduce the reference AS IT IS: the six-measurement mean within 3 SE of the difference (the exact-Gram engine sits 0.02
    below, previous test; 100 starts: 21.430 +- 0.003 against 21.439 +- 0.005, profiles/r05_config2_reference_arithmetic_100seeds.json), and
    the reference's signature measurement - traffic m2, 21.53 +- 0.01 dB in the reference as it is, 21.31-21.41 under every exact-Gram
    implementation incl. the float64 denoiser AND under an unbiased fp32 Gram of the same error size
    (profiles/r05_config2_reference_arithmetic_chain64.json) - above 21.45.  On a well-conditioned configuration the two arithmetics agree to
    1e-5 and both hold the reference's run at 1e-4."""
    solver, _ = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 180)
    eng = DEQSCIEngine(solver.nonlinear_op, iterator="anderson", m=5, beta=1.0, lam=1e-2, max_iter=180, tol=1e-5, anderson_arith="reference")
    report, _, _ = _config2_ensembles(eng, chaotic_only=True)
    chaotic = [(ps, ra, rb) for _, ps, _, ra, rb, _ in report]
    (mb, sb), (ma_, sa) = _pooled(chaotic, 0), _pooled(chaotic, 1)
    for (mid, ps, *_), (_, ra, _) in zip(report, chaotic):
        print("%-22s reference arithmetic on the GPU %.4f +- %.4f | reference as it is %.4f +- %.4f" % (mid, np.mean(ps), _se(ps), np.mean(ra), _se(ra)))
    print("six chaotic measurements: engine with the reference's Anderson arithmetic %.4f +- %.4f | reference as it is %.4f +- %.4f (diff %+.4f)"
          % (mb, sb, ma_, sa, mb - ma_))
    assert abs(mb - ma_) <= 3 * float(np.hypot(sb, sa)), (mb, ma_, sb, sa)
    m2 = [ps for mid, ps, *_ in report if mid.endswith(":2")][0]
    assert np.mean(m2) > 21.45, np.mean(m2)
    # well-conditioned: SimpleCNN @ 180 on traffic m0 against the reference's run, both arithmetics
    gold = np.load(os.path.join(GOLDEN, "e2e_SimpleCNN_anderson_180_rec.npz"))
    d = _clip("traffic_cacti.mat")
    Phi, y = d["mask"][None].to(DEV), d["meas"][None, ..., 0].contiguous().to(DEV)
    net = build_pipeline("SimpleCNN", checkpoint.shipped("cnn"), 180)[0].nonlinear_op
    r_ref = DEQSCIEngine(net, max_iter=180, anderson_arith="reference").reconstruct(y, Phi).cpu().numpy()
    r_def = DEQSCIEngine(net, max_iter=180).reconstruct(y, Phi).cpu().numpy()
    want = gold["traffic_m0"]
    assert rel_l2(r_ref, r_def) < 1e-5 and rel_l2(r_ref, want) < 1e-4 and rel_l2(r_def, want) < 1e-4


def test_conv64_rounding_on_the_networks_own_data():
    """What decides the conv64 policy (engine.py): the rounding of each 64->64 kernel against a float64 convolution on FFDNet's OWN
    data - folded net_gray weights, the activations of a noisy first iterate (x0) and of a settled one (30 Anderson iterations).
    Random data mislead here: on randn inputs F(4x4,3x3) is 6-8x noisier than a direct convolution, on a settled iterate it is on
    par with F(2x2,3x3).  Bounds = measured (profiles/r03_conv_error_real.json) + margin: F(2x2,3x3) <= 3e-7 everywhere (<= the
    direct fp32 convolution's, MIOpen: 2.4-3.7e-7); F(4x4,3x3) <= 1.2e-6 on x0, <= 3.5e-7 on the settled iterate; the split-fp16
    direct convolution (fp32 operands as hi + lo fp16 pairs on the f16 matrix cores, two accumulation chains) <= 3e-7 everywhere
    (measured: median 1.6e-7, worst layer 2.0-2.6e-7 - the most accurate of the four) and in every single layer no noisier than
    MIOpen's fp32 direct convolution of the same operands - the bar VERDICT r2 #8 set for it."""
    import torch.nn.functional as Fn
    from deqsci_amd.engine import SIGMA0
    d = _clip("traffic_cacti.mat")
    Phi = d["mask"][None].to(DEV)
    y = d["meas"][None, ..., 0].contiguous().to(DEV)
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 30)[0].nonlinear_op
    eng = DEQSCIEngine(net, max_iter=30, use_graph=False)
    inputs = {"x0": deqsci_amd.initial_point(y, Phi, None, None), "iterate30": eng.reconstruct(y, Phi)}
    den = eng.den
    for name, z in inputs.items():
        x = z.permute(0, 3, 1, 2).reshape(8, 1, 256, 256).contiguous()
        h = _hip.ffdnet_head(x, den.head_w, torch.full((1,), SIGMA0, device=DEV))
        worst = {"f22": 0.0, "f44": 0.0, "s16": 0.0, "direct": 0.0}
        for li in range(1, len(den.fast) - 1):
            w, b, relu = den.fast[li]
            ref = torch.relu(Fn.conv2d(h.double(), w.double(), b.double(), padding=1))
            e = lambda t: float((t.double() - ref).norm() / ref.norm())
            got22 = _hip.conv3x3_c64_winograd(h, den.wino[li].f22, b, relu)
            worst["f22"] = max(worst["f22"], e(got22))
            worst["f44"] = max(worst["f44"], e(_hip.conv3x3_c64_winograd44(h, den.wino[li].f44, b, relu)))
            e_direct = e(torch.relu(Fn.conv2d(h, w, b, padding=1)))
            e_s16 = e(_hip.conv3x3_c64_split16(_hip.to_split16(h), den.wino[li].s16, b, relu, out_f32=True))
            assert e_s16 < 1.1 * e_direct + 2e-8, (name, li, e_s16, e_direct)   # layer by layer no noisier than the vendor's fp32 convolution
            worst["direct"], worst["s16"] = max(worst["direct"], e_direct), max(worst["s16"], e_s16)
            h = got22
        print(name, {k: "%.2e" % v for k, v in worst.items()})
        assert worst["f22"] < 3e-7 and worst["s16"] < 3e-7 and worst["s16"] < worst["direct"], (name, worst)
        assert worst["f44"] < (1.2e-6 if name == "x0" else 3.5e-7), (name, worst)


def test_denoiser_rounding_along_the_loop():
    """One whole denoiser call on the ACTUAL inputs of a real run (z1 = GAP(X_k) captured at f-calls 0, 8, 40 of FFDNet + Anderson on
    traffic m0) against the same folded network in float64: the default path (split-fp16 64->64 layers, matrix-core head and tail) is no
    noisier than the all-fp32 Winograd F(2x2,3x3) path and well below the network run on MIOpen's fp32 convolutions
    (profiles/r03_fcall_error_along_loop.json: 0.85x and 0.5x)."""
    import torch.nn.functional as Fn
    d = _clip("traffic_cacti.mat")
    Phi = d["mask"][None].to(DEV)
    y = d["meas"][None, ..., 0].contiguous().to(DEV)
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 45)[0].nonlinear_op
    eng = DEQSCIEngine(net, max_iter=45, use_graph=False, conv64="f22")
    den, captured = eng.den, {}
    orig = den.run

    def spy(z1, call, **kw):
        if call in (0, 8, 40):
            captured[call] = z1.clone()
        return orig(z1, call, **kw)
    den.run = spy
    eng.reconstruct(y, Phi)
    den.run = orig
    assert sorted(captured) == [0, 8, 40]
    mi = DEQSCIEngine(net, max_iter=45, use_graph=False, winograd=False)
    mi.den.prepare(60, DEV)
    for call, z1 in captured.items():
        x = z1.view(8, 1, 256, 256)
        sig = den.sigma_table[call:call + 1]
        h = torch.cat((sig.double().view(1, 1, 1, 1).expand(8, 1, 128, 128), Fn.pixel_unshuffle(x.double(), 2)), 1)
        for w, b, relu in den.fast:
            h = Fn.conv2d(h, w.double(), None if b is None else b.double(), padding=1)
            h = torch.relu(h) if relu else h
        ref = Fn.pixel_shuffle(h, 2)
        err = {}
        for pol in ("f22", "fast"):
            den.conv64 = den._policy = pol
            err[pol] = float((den.run(z1, call, calibrate=True)[0].double().view_as(ref) - ref).norm() / ref.norm())
        err["miopen"] = float((mi.den.run(z1, call)[0].double().view_as(ref) - ref).norm() / ref.norm())
        print(call, {k: "%.2e" % v for k, v in err.items()})
        assert err["fast"] < 1.05 * err["f22"] and err["fast"] < 0.7 * err["miopen"], (call, err)


def test_engine_conv_layout_and_kernel_choice():
    """The engine's 64->64 layers under its conv64 policies: the F(4x4,3x3) kernel with the activations kept in its own blk32 layout
    between layers is BIT-identical to the same kernel on channels_last tensors (same arithmetic, other addresses); the kernels
    agree with one another to their rounding; the engine-level policy equals the forced kernel bit for bit; conv64_f22_calls runs
    exactly the first K f-calls on F(2x2,3x3)."""
    d = _clip("traffic_cacti.mat")
    Phi = d["mask"][None].to(DEV)
    ys = d["meas"].permute(2, 0, 1).contiguous().to(DEV)[:2]            # 2 measurements = 16 images of 128x128: 16 x 32-tile territory
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 6)[0].nonlinear_op
    assert _hip.conv64_kernel_for(16, 128, 128) == "s16" and _hip.conv64_kernel_for(16, 128, 128, policy="fast32") == "f44"
    eng = DEQSCIEngine(net, max_iter=6, use_graph=False, conv64="fast32")
    assert eng.den.blk32
    a = eng.reconstruct(ys, Phi).clone()
    eng.den.blk32 = False
    b = eng.reconstruct(ys, Phi).clone()
    eng.den.blk32 = True
    assert torch.equal(a, b)
    old = _hip.FORCE_CONV64
    try:
        _hip.FORCE_CONV64 = "f22"
        c = eng.reconstruct(ys, Phi).clone()
    finally:
        _hip.FORCE_CONV64 = old
    assert not torch.equal(a, c) and rel_l2(a.cpu().numpy(), c.cpu().numpy()) < 2e-5
    e22 = DEQSCIEngine(net, max_iter=6, use_graph=False, conv64="f22")
    assert torch.equal(e22.reconstruct(ys, Phi), c)
    # the default: split-fp16 at this size (stack=False: a launch per layer, so that the hook sees every layer - two measurements would
    # otherwise go out as two stack launches per f-call; es16 below takes those, bit-identically)
    dflt = DEQSCIEngine(net, max_iter=6, use_graph=False, stack=False)
    assert dflt.conv64 == "auto" and dflt.conv64_policy == "fast" and dflt.conv64_f22_calls is None
    seen = []
    _hip.CONV64_EVENT_HOOK = lambda kind, n, H, W: seen.append(kind)
    try:
        s_ = dflt.reconstruct(ys, Phi).clone()
        assert set(seen) == {"s16"} and len(seen) == 13 * dflt.last_info["f_calls"]
        es16 = DEQSCIEngine(net, max_iter=6, use_graph=False, conv64="s16", stack_kernel="s16")
        assert torch.equal(es16.reconstruct(ys, Phi), s_)
        ew16 = DEQSCIEngine(net, max_iter=6, use_graph=False)          # the default: the Winograd kernel under the stack launches
        sw_ = ew16.reconstruct(ys, Phi)
        assert ew16.den.stack_kernel == "w16" and ew16.last_info["stack_launches"] > 0 and not torch.equal(sw_, s_)
        assert rel_l2(sw_.cpu().numpy(), s_.cpu().numpy()) < 2e-5
        assert not torch.equal(s_, c) and rel_l2(s_.cpu().numpy(), c.cpu().numpy()) < 2e-5
        # conv64_f22_calls = K: the first K f-calls on F(2x2,3x3), the rest on the policy's kernel; K >= all calls == "f22" bit for bit
        del seen[:]
        mixed = DEQSCIEngine(net, max_iter=6, use_graph=False, conv64_f22_calls=3, stack=False)
        mixed.reconstruct(ys, Phi)
        assert seen == ["f22"] * 39 + ["s16"] * (13 * (mixed.last_info["f_calls"] - 3))
        allf = DEQSCIEngine(net, max_iter=6, use_graph=False, conv64_f22_calls=100)
        assert torch.equal(allf.reconstruct(ys, Phi), c)
    finally:
        _hip.CONV64_EVENT_HOOK = None


@pytest.mark.parametrize("anderson_arith", ["float64", "reference"])
def test_engine_graph_replay_is_bit_identical_to_eager(anderson_arith):
    """The hipGraph path replays the same kernels with the same arguments: first call of a shape eager, second captured,
    later ones replayed with new inputs - all bit-identical to an engine that never uses a graph; and when the tolerance
    test would have fired inside a replayed run the call is redone eagerly and stops where the reference stops.  In both Anderson
    arithmetics: the reference's (its Gram kernels hand out their term slots with an atomic counter - the ORDER of the slots varies from
    run to run, the sums do not) is as capturable and as deterministic as the exact one."""
    d = _clip("traffic_cacti.mat")
    Phi = d["mask"][None].to(DEV)
    ys = d["meas"].permute(2, 0, 1).contiguous().to(DEV)
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 14)[0].nonlinear_op
    eager = DEQSCIEngine(net, max_iter=14, use_graph=False, anderson_arith=anderson_arith)
    graph = DEQSCIEngine(net, max_iter=14, use_graph=True, anderson_arith=anderson_arith)
    want = [eager.reconstruct(ys[i:i + 1], Phi).clone() for i in range(3)]
    assert eager.last_info["graph"] is False
    got0 = graph.reconstruct(ys[0:1], Phi)                   # first call of the shape: eager warm-up
    assert graph.last_info["graph"] is False and torch.equal(got0, want[0])
    got0 = graph.reconstruct(ys[0:1], Phi)                   # capture + first replay
    assert graph.last_info["graph"] is True and torch.equal(got0, want[0])
    for i in (1, 2):                                         # replays with new measurements in the static input buffers
        got = graph.reconstruct(ys[i:i + 1], Phi)
        assert graph.last_info["graph"] is True and torch.equal(got, want[i])
    assert graph.last_info["f_calls"] == eager.last_info["f_calls"] == 15
    assert graph.last_info["res"] == eager.last_info["res"] and graph.last_info["res_per_sample"] == eager.last_info["res_per_sample"]
    # "auto": one 256x256 measurement is graph territory, a batch of 6 is not
    auto = DEQSCIEngine(net, max_iter=5)
    auto.reconstruct(ys[0:1], Phi)
    auto.reconstruct(ys[0:1], Phi)
    assert auto.last_info["graph"] is True
    auto.reconstruct(ys, Phi)
    auto.reconstruct(ys, Phi)
    assert auto.last_info["graph"] is False
    # early stop inside a replayed run -> eager redo with the reference's stopping rule
    Phi2, Phie2, x2, z2, y2, Ps2 = make_case(1, 16, 16, 8, seed=11)
    e1 = DEQSCIEngine(_Contract(), max_iter=50, tol=1e-4, use_graph=False)
    g1 = DEQSCIEngine(_Contract(), max_iter=50, tol=1e-4, use_graph=True)
    ref = e1.reconstruct(G(y2), G(Phi2)).clone()
    for _ in range(3):
        out = g1.reconstruct(G(y2), G(Phi2))
        assert torch.equal(out, ref) and g1.last_info["f_calls"] == e1.last_info["f_calls"] < 40 and g1.last_info["graph"] is False


def test_harness_graph_replay_equals_eager_measurement_by_measurement():
    """The drop-in usage (FFDNet @180, one measurement per call, the reference's schedule) on the hipGraph path: per-measurement
    PSNRs agree with the eager engine's to the last digit.  (Its speed is reported by tools/parity_report.py, not asserted here.)"""
    import time
    from deqsci_amd.harness import SCITestDataset, test_solver_sci
    _, deq = _pipeline("ffdnet", 180)
    ds = SCITestDataset(orc.DATA_DIR)
    rec_g, rec_e = [], []
    test_solver_sci(deq, test_dataloader=ds, save_img_path="", verbose=False, save_image=False, batch_measurements=False)   # warm-up + capture
    test_solver_sci(deq, test_dataloader=ds, save_img_path="", verbose=False, save_image=False, batch_measurements=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    test_solver_sci(deq, test_dataloader=ds, save_img_path="", verbose=False, save_image=False, batch_measurements=False, records=rec_g)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert deq._engine_for().last_info["graph"] is True
    deq._engine_for().use_graph = False
    test_solver_sci(deq, test_dataloader=ds, save_img_path="", verbose=False, save_image=False, batch_measurements=False, records=rec_e)
    assert [r["psnr"] for r in rec_g] == [r["psnr"] for r in rec_e]
    print(f"\nharness FFDNet@180 measurement-by-measurement (hipGraph): 64 frames in {dt:.3f} s = {64 / dt:.1f} frames/s")


def test_rccl_single_rank_collectives():
    """RCCL itself on this box, as far as ONE GPU lets it run: a fresh process creates the "nccl" process group (world size 1, device_id as
    distributed.init_from_env passes it) and issues exactly the collectives bench.py's N > 1 path uses - barrier, all_gather_into_tensor of a
    reconstruction-shaped tensor, all_reduce(MAX) of the step time, all_gather_object of the per-rank record.  The N > 1 run over xGMI is the
    driver's; this pins that the library loads, initialises and completes a collective on the device (HSA_ENABLE_IPC_MODE_LEGACY=0 as exported)."""
    import subprocess
    import sys
    from deqsci_amd.distributed import free_port
    code = (
        "import os, torch, torch.distributed as dist\n"
        "torch.cuda.set_device(0); dev = torch.device('cuda', 0)\n"
        "dist.init_process_group(backend='nccl', rank=0, world_size=1, device_id=dev)\n"
        "assert dist.get_backend() == 'nccl'\n"
        "dist.barrier()\n"
        "x = torch.rand(2, 256, 256, 8, device=dev); full = torch.empty_like(x)\n"
        "dist.all_gather_into_tensor(full, x); torch.cuda.synchronize(); assert torch.equal(full, x)\n"
        "t = torch.tensor([1.5], device=dev, dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert float(t) == 1.5\n"
        "got = [None]; dist.all_gather_object(got, {'rank': 0}); assert got == [{'rank': 0}]\n"
        "dist.barrier(); dist.destroy_process_group(); print('rccl ok')\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "rccl ok" in out.stdout, (out.stdout[-500:], out.stderr[-1500:])


def test_bench_two_ranks_on_one_gpu_real_engine():
    """World size 2 with the REAL engine: `bench.py --gpus 2 --ranks-share-gpu0` starts two rank processes that share cuda:0
    and gather through gloo (RCCL refuses two ranks on one device; the build pool has one GPU per box).  Checks the launcher,
    the sharded step and the gathered result's bookkeeping - not speed, not RCCL."""
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--ranks-share-gpu0", "--batch-per-gpu", "2", "--iters", "12",
                          "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-hbm-stream", "--no-graph"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 4 and rec["config"]["f_calls_per_step"] == 13
    assert rec["allgather_ms_per_step"] > 0 and rec["value"] > 0 and 0 < rec["final_res"] < 1
    assert rec["scaling"] == "weak" and "roofline" in rec and rec["roofline"]["bound"] in ("mfma", "hbm")        # N > 1 lines keep the roofline (per-launch timing needs --no-graph at this size)
    # two processes on one GPU is exactly what can strand a stack launch's workgroups: the line says whether it happened (rank 0's engine:
    # timed-out calls were redone per layer, so the result is valid either way), and what ran
    assert rec["config"]["stack_timeouts"] in (0, 1, 2) and rec["config"]["stack_kernel"] == "w16"
    print("two ranks on one GPU: stack_timeouts =", rec["config"]["stack_timeouts"])
    # the strong-scaling mode on the same rig, ragged: 3 measurements over 2 ranks (2 + 1, the tail padded and masked)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--ranks-share-gpu0", "--global-batch", "3", "--iters", "12",
                          "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-hbm-stream"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["scaling"] == "strong" and rec["config"]["global_batch"] == 3 and rec["config"]["batch_per_gpu"] == 2
    assert rec["allgather_bytes_per_step"] == 2 * 2 * 256 * 256 * 8 * 4 and rec["value"] > 0 and 0 < rec["final_res"] < 1


def test_engine_512x512x16_ffdnet_vs_oracle():
    """BASELINE config 4 with its own denoiser: FFDNet at 512x512x16 (half-resolution planes of 256x256 through the head / tail
    stencils and the Winograd layers, the B = 16 kernel instantiations, batch 2 with per-sample masks), and_maxiters=5, against
    the CPU oracle with an exactly accumulated Gram (see test_engine_512x512x16_vs_oracle for why) and, per f-call, against
    the oracle's single-iterate map."""
    g = torch.Generator().manual_seed(14)
    H = W = 512
    B = 16
    Phi = (torch.rand(2, H, W, B, generator=g) < 0.5).float()
    x = torch.rand(2, H, W, B, generator=g)
    y = orc.sci_forward(x, Phi)
    Ps = orc.phi_sum(Phi)
    kw = dict(m=5, beta=1.0, lam=1e-2, max_iter=5, tol=1e-5)
    f64 = orc.ProxGradSCI("ffdnet")
    want64, wres = orc.deq_forward(f64, orc.andersonexp, y, Phi, Ps, orc.initial_point(y, Phi), gram_dtype=torch.float64, **kw)
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 5)[0].nonlinear_op
    eng = DEQSCIEngine(net, max_iter=5, anderson_arith="float64")
    got = eng.reconstruct(G(y), G(Phi)).cpu().numpy()
    assert eng.last_info["f_calls"] == 6 and f64.calls == 7
    assert rel_l2(got, want64.numpy()) < 5e-5
    assert abs(eng.last_info["res"] - wres) < 5e-3 * wres
    # one f-call on its own (GAP + FFDNet at sigma_0): the drop-in solver against the oracle map
    solver, _ = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 5)
    f1 = orc.ProxGradSCI("ffdnet")
    z0 = orc.initial_point(y, Phi)
    with torch.no_grad():
        out = solver(G(z0), G(y), G(Phi), G(Ps))
    assert rel_l2(out.cpu().numpy(), f1(z0, y, Phi, Ps).numpy()) < 1e-5


# ----------------------------------------------------------------------------- split-fp16 under Winograd F(2,3) x direct (csrc/conv_w16.hip)
@pytest.mark.parametrize("shape", [(1, 8, 64), (8, 128, 128), (3, 40, 56), (1, 18, 14), (2, 17, 23), (1, 1, 1), (70, 64, 80), (33, 50, 46),
                                   (300, 16, 16), (2, 250, 130), (1, 9, 65)])
def test_wino16_conv64_vs_torch(shape):
    """The 64->64 layer (networks/ffdnet/models.py:53-58: Conv2d(64,64,3,padding=1) + BatchNorm(eval) + ReLU) as split-fp16 Winograd F(2,3)
    along x nested in the direct sum along y vs conv2d in fp64: random asymmetric weights, block tiles of 8 x 64 pixels that stick out of
    the image on every side, single tiles and many per workgroup, odd widths (a Winograd tile = two columns), a three-layer chain of p32
    activations, repeated launches.  Bound on random data 2.5e-7, the direct kernel's (measured 1.2-1.5e-7).  The p32 round trip is exact."""
    import torch.nn.functional as Fn
    n, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(12)
    x = torch.randn(n, 64, H, W, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    ws = [torch.randn(64, 64, 3, 3, device=DEV, generator=g) * 0.05 for _ in range(3)]
    bs = [torch.randn(64, device=DEV, generator=g) * 0.3 for _ in range(3)]
    Ws = [_hip.Wino16Weights(w) for w in ws]

    def err(a, b):
        return float((a.double() - b).norm() / b.norm())
    xp = _hip.P32.from_nchw(x)
    assert torch.equal(xp.to_nchw(), x)
    ref = Fn.conv2d(x.double(), ws[0].double(), padding=1)
    got = _hip.conv3x3_c64_wino16(xp, Ws[0], None, relu=False)
    assert isinstance(got, _hip.P32) and err(got.to_nchw(), ref) < 2.5e-7
    want = torch.relu(ref + bs[0].double().view(1, -1, 1, 1))
    o = _hip.P32.empty(n, H, W, DEV)
    o.t.fill_(float("nan"))
    got_b = _hip.conv3x3_c64_wino16(xp, Ws[0], bs[0], True, out=o)
    assert got_b is o and bool(torch.isfinite(o.to_nchw()).all()) and err(o.to_nchw(), want) < 2.5e-7
    assert torch.equal(_hip.conv3x3_c64_wino16(xp, Ws[0], bs[0], True).to_nchw(), o.to_nchw())       # repeated launches: nothing left behind
    h, wd = xp, x.double()
    for i in range(3):
        h = _hip.conv3x3_c64_wino16(h, Ws[i], bs[i], True)
        wd = torch.relu(Fn.conv2d(wd, ws[i].double(), bs[i].double(), padding=1))
    assert err(h.to_nchw(), wd) < 5e-7
    with pytest.raises(_hip.DeqsciHipError):
        _hip.conv3x3_c64_wino16(_hip.to_split16(x), Ws[0], bs[0], True)          # an Sp16 is not a P32
    with pytest.raises(_hip.DeqsciHipError):
        _hip.conv3x3_c64_wino16(xp, _hip.Split16Weights(ws[0]), bs[0], True)


@pytest.mark.parametrize("n,H,W,n_layers,data_ranges", [(8, 128, 128, 13, True),      # the reference's usage: one measurement, FFDNet's 13 layers, ONE tile per CU
                                                        (3, 128, 128, 5, True),       # 96 tiles: every image's tiles straddle XCDs
                                                        (5, 64, 96, 4, False),        # 80 tiles, ragged width (two block columns, the second half empty), fixed exponents
                                                        (1, 100, 76, 2, True),        # ragged edges, 26 tiles
                                                        (7, 8, 64, 3, True),          # one tile per image: no neighbours at all
                                                        (32, 128, 128, 13, True),     # a slice of a batch: FOUR tiles per workgroup
                                                        (19, 128, 128, 4, False),     # 608 tiles: workgroups with two and with three tiles
                                                        (2, 256, 512, 3, True),       # 512 tiles, a workgroup's own tiles are neighbours
                                                        (4, 64, 64, 13, True),        # 32 tiles, 1 MB per activation: EVERYTHING stays in the L2s for 13 layers
                                                        (6, 48, 80, 12, False)])      # 72 tiles dealt over the XCDs: a stale line would be read
def test_wino16_stack_is_bit_identical_to_single_launches(n, H, W, n_layers, data_ranges):
    """deqsci_conv3x3_c64_wino16_stack against the same layers as single launches of the same kernel: the same bits, five launches on the
    same progress words through the 32-bit wrap, both ping-pong buffers poisoned before every launch (see
    test_split16_stack_is_bit_identical_to_single_launches: the protocol is that kernel's; the fetch runs two half-stages ahead here, the
    poll half a tile, a tile's outputs leave during the next tile's first half-stage)."""
    g = torch.Generator(device=DEV).manual_seed(100 * n + n_layers)
    x = (torch.relu(torch.randn(n, 64, H, W, device=DEV, generator=g)) * torch.logspace(0, -2, n, device=DEV).view(n, 1, 1, 1))
    x = x.contiguous(memory_format=torch.channels_last)
    ws = [torch.randn(64, 64, 3, 3, device=DEV, generator=g) * 0.06 for _ in range(n_layers)]
    bs = [None if i == 1 else torch.randn(64, device=DEV, generator=g) * 0.1 for i in range(n_layers)]
    relus = [i != n_layers - 2 for i in range(n_layers)]
    Ws = [_hip.Wino16Weights(w) for w in ws]
    rng = torch.zeros(n_layers + 1, n, device=DEV) if data_ranges else None
    if data_ranges:                                             # (the ranges as the engine's measuring f-call finds them: the direct kernel's measuring launches)
        _hip.absmax(x, rng[0])
        hs = _hip.to_split16(x, rng=rng[0])
        for i in range(n_layers):
            W16 = _hip.Split16Weights(ws[i])
            _hip.conv3x3_c64_split16(hs, W16, bs[i], relus[i], track=rng[i + 1])
            hs = _hip.conv3x3_c64_split16(hs, W16, bs[i], relus[i], out_rng=rng[i + 1])
    h0 = _hip.P32.from_nchw(x, rng=None if rng is None else rng[0])
    h = h0
    for i in range(n_layers):
        h = _hip.conv3x3_c64_wino16(h, Ws[i], bs[i], relus[i], out_rng=None if rng is None else rng[i + 1])
    want = h.to_nchw()
    assert bool(torch.isfinite(want).all())
    if data_ranges:                                             # ... and the two kernels agree on what they computed (up to 13 layers of ~1.5e-7 each)
        assert rel_l2(want.cpu().numpy(), hs.to_nchw().cpu().numpy()) < 3e-6
    stack = _hip.Wino16Stack(list(zip(Ws, bs, relus)), DEV)
    flags, bufs = stack.flags(n, H, W), stack.state(n, H, W)
    flags.view(-1, 32)[:-1, 0] = -20                            # the words count on for ever: start them 20 below the 32-bit wrap
    for rep in range(5):
        for b in bufs:
            b.t.fill_(float("nan"))
        out = _hip.conv3x3_c64_wino16_stack(h0, stack, rng, per_launch=n)
        assert out is bufs[(n_layers - 1) % 2]
        assert torch.equal(out.to_nchw(), want), (rep, float((out.to_nchw() - want).abs().max()))
        assert out.exponents() == h.exponents()
    fl = flags.cpu().view(-1, 32)[:, 0]
    assert int(fl[-1]) == 0 and bool((fl[:-1] == 5 * n_layers - 20).all())      # every tile: five launches of n_layers layers, no time-out
    assert not stack.timed_out()


@pytest.mark.parametrize("n,H,W,per", [(16, 128, 128, 8), (20, 128, 128, 8), (64, 128, 128, None), (40, 128, 128, None), (7, 100, 76, 3)])
def test_wino16_stack_slices_a_batch(n, H, W, per):
    """Slices of a batch, one stack launch after the other (two launch shapes with their own progress words; the default policy's 32-image
    slices): the per-layer launches' bits over the whole batch, per-image ranges read through the slice's offset into the slot table."""
    n_layers = 4
    g = torch.Generator(device=DEV).manual_seed(n)
    x = (torch.relu(torch.randn(n, 64, H, W, device=DEV, generator=g)) * torch.logspace(0, -3, n, device=DEV).view(n, 1, 1, 1))
    x = x.contiguous(memory_format=torch.channels_last)
    Ws = [_hip.Wino16Weights(torch.randn(64, 64, 3, 3, device=DEV, generator=g) * 0.06) for _ in range(n_layers)]
    bs = [torch.randn(64, device=DEV, generator=g) * 0.1 for _ in range(n_layers)]
    rng = torch.zeros(n_layers + 1, n, device=DEV)
    _hip.absmax(x, rng[0])
    h0 = _hip.P32.from_nchw(x, rng=rng[0])
    h = h0
    for i in range(n_layers):                                   # (ranges: measured on a first pass with a generous fixed scale, as an fp32 tensor)
        t = _hip.conv3x3_c64_wino16(_hip.P32.from_nchw(h.to_nchw(), exp=4), Ws[i], bs[i], True, out_exp=4).to_nchw()
        _hip.absmax(t.contiguous(), rng[i + 1])
        h = _hip.conv3x3_c64_wino16(h, Ws[i], bs[i], True, out_rng=rng[i + 1])
    assert bool(torch.isfinite(h.to_nchw()).all())
    stack = _hip.Wino16Stack([(w, b, True) for w, b in zip(Ws, bs)], DEV)
    assert _hip.split16_stack_per_launch(64, 128, 128, tile=stack.TILE) == 32 and _hip.split16_stack_per_launch(40, 128, 128, cus=256, tile=stack.TILE) == 32
    for rep in range(3):
        for b in stack.state(n, H, W):
            b.t.fill_(float("nan"))
        out = _hip.conv3x3_c64_wino16_stack(h0, stack, rng, per_launch=per)
        assert torch.equal(out.to_nchw(), h.to_nchw()), rep
    assert not stack.timed_out()


@pytest.mark.parametrize("shape", [(3, 16, 32), (2, 13, 37), (8, 128, 128), (5, 8, 64), (2, 40, 130)])
def test_ffdnet_edges_p32_vs_split16(shape):
    """FFDNet's first and last layer writing / reading p32 (deqsci_ffdnet_head_p32 / _tail_p32: in front of / behind a run of Winograd
    layers) against their sp16 forms on the same data: the head's p32 output rounded to hi + lo IS the sp16 output (same arithmetic, the
    split moved to the consumer), the tail on a p32 of the same activation equals the sp16 tail to fp32 rounding and both sit on the
    float64 reference (networks/ffdnet/models.py:46-64, functions.py:16-81)."""
    import torch.nn.functional as Fn
    n, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(7)
    x = torch.rand(n, 1, 2 * H, 2 * W, device=DEV, generator=g)
    sig = torch.full((1,), 0.2, device=DEV)
    wh = torch.randn(64, 5, 3, 3, device=DEV, generator=g) * 0.2
    wt = torch.randn(4, 64, 3, 3, device=DEV, generator=g) * 0.1
    Wh, Wt = _hip.HeadSplit16Weights(wh), _hip.TailSplit16Weights(wt)
    rng = torch.zeros(2, n, device=DEV)
    _hip.absmax(x, rng[0])
    _hip.ffdnet_head_split16(x, Wh, sig, in_rng=rng[0], out_exp=0, track=rng[1])
    hs = _hip.ffdnet_head_split16(x, Wh, sig, in_rng=rng[0], out_rng=rng[1])
    hp = _hip.ffdnet_head_p32(x, Wh, sig, in_rng=rng[0], out_rng=rng[1])
    a, b = hs.to_nchw(), hp.to_nchw()
    assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 2.0 ** -21 and float((a - b).abs().max()) <= float(b.abs().max()) * 2.0 ** -21
    full = torch.cat((sig.double().view(1, 1, 1, 1).expand(n, 1, H, W), Fn.pixel_unshuffle(x.double(), 2)), 1)
    ref_h = torch.relu(Fn.conv2d(full, wh.double(), padding=1))
    assert rel_l2(b.double().cpu().numpy(), ref_h.cpu().numpy()) < 3e-7
    ref_t = Fn.pixel_shuffle(Fn.conv2d(b.double(), wt.double(), padding=1), 2)
    tp = _hip.ffdnet_tail_p32(hp, Wt)
    ts = _hip.tail_split16(hs, Wt)
    assert tp.shape == (n, 1, 2 * H, 2 * W)
    assert rel_l2(tp.double().cpu().numpy(), ref_t.cpu().numpy()) < 3e-7 and rel_l2(tp.cpu().numpy(), ts.cpu().numpy()) < 5e-7


@pytest.mark.parametrize("shape", [(2, 40, 72), (1, 256, 256), (3, 17, 130)])
def test_simplecnn_edges_p32_vs_split16(shape):
    """SimpleCNN's first and last layer (SimpleCNN_models.py:43-45, 55-56) writing / reading p32 - in front of / behind conv3x3_c64_wino16
    layers - against their sp16 forms: the head's p32 output rounded to hi + lo IS the sp16 output, the tail on the p32 equals the sp16 tail to
    fp32 rounding, both on the float64 reference."""
    import torch.nn.functional as Fn
    n, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(9)
    x = torch.rand(n, 1, H, W, device=DEV, generator=g)
    wh = torch.randn(64, 1, 3, 3, device=DEV, generator=g) * 0.3
    wt = torch.randn(1, 64, 3, 3, device=DEV, generator=g) * 0.1
    Wh, Wt = _hip.pack_c1_to_64_weights(wh), _hip.TailSplit16Weights(wt)
    rng = torch.zeros(2, n, device=DEV)
    _hip.conv3x3_c1_to_64(x, Wh, relu=True, sp16=True, out_exp=0, track=rng[1])
    hs = _hip.conv3x3_c1_to_64(x, Wh, relu=True, sp16=True, out_rng=rng[1])
    hp = _hip.conv3x3_c1_to_64(x, Wh, relu=True, p32=True, out_rng=rng[1])
    a, b = hs.to_nchw(), hp.to_nchw()
    assert float((a - b).abs().max()) <= float(b.abs().max()) * 2.0 ** -21
    ref_h = torch.relu(Fn.conv2d(x.double(), wh.double(), padding=1))
    assert rel_l2(b.double().cpu().numpy(), ref_h.cpu().numpy()) < 3e-7
    ref_t = Fn.conv2d(b.double(), wt.double(), padding=1)
    tp = _hip.ffdnet_tail_p32(hp, Wt)
    ts = _hip.tail_split16(hs, Wt)
    assert tp.shape == (n, 1, H, W)
    assert rel_l2(tp.double().cpu().numpy(), ref_t.cpu().numpy()) < 3e-7 and rel_l2(tp.cpu().numpy(), ts.cpu().numpy()) < 5e-7


def test_wino16_rounding_on_the_networks_own_data():
    """VERDICT r4 #1's bar for the Winograd kernel, on FFDNet's own data (folded net_gray weights, the activations of a noisy first iterate
    and of a settled one): every layer <= 3e-7 against a float64 convolution and no noisier than 1.1 x MIOpen's fp32 direct convolution of
    the same operands (measured: median 1.3e-7; the direct split-fp16 kernel 1.6e-7, tools/wino16_numerics.py predicted as much)."""
    import torch.nn.functional as Fn
    from deqsci_amd.engine import SIGMA0
    d = _clip("traffic_cacti.mat")
    Phi = d["mask"][None].to(DEV)
    y = d["meas"][None, ..., 0].contiguous().to(DEV)
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 30)[0].nonlinear_op
    eng = DEQSCIEngine(net, max_iter=30, use_graph=False)
    inputs = {"x0": deqsci_amd.initial_point(y, Phi, None, None), "iterate30": eng.reconstruct(y, Phi)}
    den = eng.den
    for name, z in inputs.items():
        x = z.permute(0, 3, 1, 2).reshape(8, 1, 256, 256).contiguous()
        h = _hip.ffdnet_head(x, den.head_w, torch.full((1,), SIGMA0, device=DEV))
        worst = {"w16": 0.0, "s16": 0.0, "direct": 0.0}
        for li in range(1, len(den.fast) - 1):
            w, b, relu = den.fast[li]
            ref = torch.relu(Fn.conv2d(h.double(), w.double(), b.double(), padding=1))
            e = lambda t: float((t.double() - ref).norm() / ref.norm())
            slot = torch.zeros(8, device=DEV)
            _hip.absmax(h, slot)
            oslot = torch.zeros(8, device=DEV)
            _hip.absmax(ref.float().contiguous(), oslot)
            e_w16 = e(_hip.conv3x3_c64_wino16(_hip.P32.from_nchw(h, rng=slot), den.wino[li].w16, b, relu, out_rng=oslot).to_nchw())
            e_direct = e(torch.relu(Fn.conv2d(h, w, b, padding=1)))
            e_s16 = e(_hip.conv3x3_c64_split16(_hip.to_split16(h, rng=slot), den.wino[li].s16, b, relu, out_f32=True))
            assert e_w16 < 3e-7 and e_w16 < 1.1 * e_direct + 2e-8, (name, li, e_w16, e_direct)
            worst = {"w16": max(worst["w16"], e_w16), "s16": max(worst["s16"], e_s16), "direct": max(worst["direct"], e_direct)}
            h = _hip.conv3x3_c64_winograd(h, den.wino[li].f22, b, relu)
        print(name, {k: "%.2e" % v for k, v in worst.items()})
        assert worst["w16"] < worst["direct"]


def test_ranges_of_the_first_call_serve_the_whole_loop():
    """the engine measures the activations' ranges ONCE, at f-call 0, and reuses them for every later call.  That regime
    directly: ranges measured on the call-0 input only, then the denoiser (the default path: matrix-core head, the Winograd stack launch,
    matrix-core tail) on the inputs of f-calls 8, 40 and 150 of a real run WITHOUT re-measuring - still no noisier than the all-fp32
    F(2x2,3x3) path (x 1.05) and well below MIOpen's fp32 network (x 0.7), as test_denoiser_rounding_along_the_loop holds with fresh ranges."""
    import torch.nn.functional as Fn
    d = _clip("traffic_cacti.mat")
    Phi = d["mask"][None].to(DEV)
    y = d["meas"][None, ..., 0].contiguous().to(DEV)
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 160)[0].nonlinear_op
    eng = DEQSCIEngine(net, max_iter=160, use_graph=False, conv64="f22")
    den, captured = eng.den, {}
    orig = den.run

    def spy(z1, call, **kw):
        if call in (0, 8, 40, 150):
            captured[call] = z1.clone()
        return orig(z1, call, **kw)
    den.run = spy
    eng.reconstruct(y, Phi)
    den.run = orig
    assert sorted(captured) == [0, 8, 40, 150]
    mi = DEQSCIEngine(net, max_iter=160, use_graph=False, winograd=False)
    mi.den.prepare(170, DEV)
    fast = DEQSCIEngine(net, max_iter=160, use_graph=False)      # the default policy: its ranges come from call 0 and stay
    fast.den.prepare(170, DEV, n_img=8)
    fast.den.run(captured[0], 0, calibrate=True)
    r0 = fast.den.ranges.clone()
    for call in (8, 40, 150):
        z1 = captured[call]
        x = z1.view(8, 1, 256, 256)
        sig = den.sigma_table[call:call + 1]
        h = torch.cat((sig.double().view(1, 1, 1, 1).expand(8, 1, 128, 128), Fn.pixel_unshuffle(x.double(), 2)), 1)
        for w, b, relu in den.fast:
            h = Fn.conv2d(h, w.double(), None if b is None else b.double(), padding=1)
            h = torch.relu(h) if relu else h
        ref = Fn.pixel_shuffle(h, 2)
        err = {}
        launches = fast.den.stack_launches
        err["fast"] = float((fast.den.run(z1, call)[0].double().view_as(ref) - ref).norm() / ref.norm())       # NO calibrate: call-0 ranges
        assert fast.den.stack_launches == launches + 1 and torch.equal(fast.den.ranges, r0)                       # ... on the stack launch
        den.conv64 = den._policy = "f22"
        err["f22"] = float((den.run(z1, call)[0].double().view_as(ref) - ref).norm() / ref.norm())
        err["miopen"] = float((mi.den.run(z1, call)[0].double().view_as(ref) - ref).norm() / ref.norm())
        print(call, {k: "%.2e" % v for k, v in err.items()})
        assert err["fast"] < 1.05 * err["f22"] and err["fast"] < 0.7 * err["miopen"], (call, err)
    assert not fast.den.stack_timed_out()
