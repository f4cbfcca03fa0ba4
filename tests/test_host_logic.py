"""CPU: host-side logic of the product - denoiser modules / checkpoint keys / loader / PSNR / sigma
table / sharding arithmetic / CLI plumbing - against the oracle and the reference goldens.
No HIP compute happens here; the product refuses CPU tensors."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, rel_l2
from oracle import deqsci_oracle as orc

import deqsci_amd
from deqsci_amd import _hip, checkpoint, harness
from deqsci_amd.cli import build_denoiser, build_pipeline, main as cli_main
from deqsci_amd.distributed import shard_bounds
from deqsci_amd.engine import DEQSCIEngine, sigma_schedule


def test_sigma_table_matches_reference_sequence():
    g = np.load(os.path.join(GOLDEN, "sigma.npz"))["sigma"]
    assert np.array_equal(sigma_schedule(len(g)), g)


def test_denoiser_modules_load_reference_checkpoints_and_match_goldens():
    g = np.load(os.path.join(GOLDEN, "nets.npz"))
    x = torch.from_numpy(g["x"])
    ff = build_denoiser("ffdnet").eval()
    sd, _ = checkpoint.read_state_dict(checkpoint.shipped("ffdnet_gray"))
    assert set(sd) == {k for k in ff.state_dict() if not k.endswith("num_batches_tracked")} and len(sd) == 67
    assert "intermediate_dncnn.itermediate_dncnn.41.weight" in sd       # the reference's key spelling
    ff.load_state_dict(sd)
    assert sum(v.numel() for v in sd.values()) == 487744
    with torch.no_grad():
        for k in (0, 1, 50):
            out = ff(x, torch.from_numpy(g[f"ffdnet_sigma_k{k}"]))
            assert rel_l2(out.numpy(), g[f"ffdnet_noise_k{k}"]) < 1e-6
    solver, deq = build_pipeline("SimpleCNN", checkpoint.shipped("cnn"), 10, device="cpu")
    assert sorted(solver.state_dict()) == [f"nonlinear_op.dncnn.{i}.weight" for i in (0, 2, 4, 6)]
    with torch.no_grad():
        assert rel_l2(solver.nonlinear_op(x).numpy(), g["cnn_noise"]) < 1e-6
    rsn, _ = build_pipeline("RealSN_SimpleCNN", checkpoint.shipped("rsn_cnn"), 10, device="cpu")
    assert len(rsn.state_dict()) == 12 and rsn.nonlinear_op.tag == "denoiser"
    w = np.load(checkpoint.shipped("rsn_cnn"))["nonlinear_op.dncnn.2.weight"]
    assert np.array_equal(rsn.nonlinear_op.dncnn[2].weight.numpy(), w)
    with pytest.raises(NotImplementedError):
        build_denoiser("unet")
    with pytest.raises(FileNotFoundError):
        checkpoint.read_state_dict("/nonexistent/ffdnet.ckpt")


def test_folded_bn_ffdnet_matches_module():
    from deqsci_amd.engine import _Denoiser
    g = np.load(os.path.join(GOLDEN, "nets.npz"))
    ff = build_denoiser("ffdnet").eval()
    ff.load_state_dict(checkpoint.read_state_dict(checkpoint.shipped("ffdnet_gray"))[0])
    den = _Denoiser(ff)
    den.prepare(60, "cpu")
    x = torch.from_numpy(g["x"]).view(1, 8, 32, 32)
    with torch.no_grad():
        out, is_noise = den.run(x, 50)
    # folding BN into the conv weights re-rounds them: ~1e-5 on the (small) predicted noise, ~1e-6 on z1 - noise
    assert is_noise and rel_l2(out.reshape(8, 1, 32, 32).numpy(), g["ffdnet_noise_k50"]) < 5e-5
    xin = g["x"]
    assert rel_l2(xin - out.reshape(8, 1, 32, 32).numpy(), xin - g["ffdnet_noise_k50"]) < 2e-6
    # load_state_dict after construction must invalidate the folded copy
    with torch.no_grad():
        ff.intermediate_dncnn.itermediate_dncnn[0].weight.mul_(0.5)
    den.prepare(60, "cpu")
    with torch.no_grad():
        out2, _ = den.run(x, 50)
    assert rel_l2(out2.numpy(), out.numpy()) > 1e-3


def test_loader_and_psnr_match_oracle():
    ds = harness.SCITestDataset(orc.DATA_DIR)
    assert ds.filelist == ["drop8_cacti.mat", "runner8_cacti.mat", "traffic_cacti.mat"] == orc.list_clips()
    d = ds[2]
    o = orc.load_clip(os.path.join(orc.DATA_DIR, "traffic_cacti.mat"))
    for k in ("gt", "mask", "meas"):
        assert d[k].dtype == np.float32 and np.array_equal(d[k], o[k])
    assert d["gt"].shape == (256, 256, 48) and d["meas"].shape == (256, 256, 6) and d["file"] == "traffic_cacti.mat"
    # shipped data is self-consistent: sum_b mask*orig == meas
    assert np.abs((d["mask"] * d["gt"][..., :8]).sum(2) - d["meas"][..., 0]).max() < 1e-5
    a, b = np.random.RandomState(0).rand(1, 8, 8, 8).astype(np.float32), np.random.RandomState(1).rand(1, 8, 8, 8).astype(np.float32)
    assert harness.psnr(a, b) == orc.psnr(a, b)
    img = harness.tensor_to_np(torch.tensor([[0.5, 1.5], [-1.0, 0.25]]))
    assert img.shape == (2, 2, 1) and img.max() == 255.0 and img.min() == 0.0


def test_product_refuses_cpu_tensors_and_missing_library(monkeypatch):
    z = torch.zeros(1, 4, 4, 8)
    for fn in (lambda: deqsci_amd.A_torch_(z, z), lambda: deqsci_amd.At_torch_(z[..., 0], z),
               lambda: deqsci_amd.andersonexp(lambda t: t, z), lambda: deqsci_amd.phi_sum(z)):
        with pytest.raises(_hip.DeqsciHipError):
            fn()
    eng = DEQSCIEngine(build_denoiser("SimpleCNN").eval(), max_iter=5)
    with pytest.raises(_hip.DeqsciHipError):
        eng.reconstruct(z[..., 0], z)
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "_LIB_PATH", "/nonexistent/libdeqsci_hip.so")
    with pytest.raises(_hip.DeqsciHipError, match="no CPU"):
        _hip.load()


def test_shard_bounds_cover_batch_exactly():
    for M in (1, 7, 8, 64, 65):
        for R in (1, 2, 4, 8):
            spans = [shard_bounds(M, R, r) for r in range(R)]
            assert spans[0][0] == 0 and max(s[1] for s in spans) == M
            assert sum(s[1] - s[0] for s in spans) == M
            assert all(s[2] == -(-M // R) for s in spans)


def test_cli_flags_and_errors():
    with pytest.raises(SystemExit):
        cli_main(["--inference", "False"])                               # training is out of scope -> refused
    with pytest.raises(SystemExit):
        cli_main(["--denoiser", "ffdnet", "--and_maxiters", "3"])        # no GPU here -> refuses, no CPU path
    with pytest.raises(SystemExit):
        cli_main(["--and_maxiters", "abc"])


def test_deqfixedpoint_routing_rules():
    solver, deq = build_pipeline("SimpleCNN", checkpoint.shipped("cnn"), 7, device="cpu")
    eng = deq._engine_for()
    assert isinstance(eng, DEQSCIEngine) and eng.max_iter == 7 and eng.m == 5 and eng.lam == 1e-2 and eng.iterator == "anderson"
    assert deq._engine_for() is eng                                       # cached
    deq2 = deqsci_amd.DEQFixedPoint(solver, deqsci_amd.forward_iteration, max_iter=9, tol=1e-5)
    assert deq2._engine_for().iterator == "picard"
    deq3 = deqsci_amd.DEQFixedPoint(solver, lambda f, x0, **kw: (x0, 0.0))
    assert deq3._engine_for() is None                                     # foreign iterator -> generic path
    other = deqsci_amd.EquilibriumProxGradSCI(A=lambda x, P: x, At=deqsci_amd.At_torch_, nonlinear_operator=solver.nonlinear_op, eta=0.2)
    assert deqsci_amd.DEQFixedPoint(other, deqsci_amd.andersonexp)._engine_for() is None
    with pytest.raises(TypeError):
        deqsci_amd.DEQFixedPoint(solver, deqsci_amd.forward_iteration, m=5)._engine_for()   # forward_iteration has no m


def test_golden_metadata_present():
    for tag in ("SimpleCNN_anderson_10", "ffdnet_anderson_10", "ffdnet_anderson_30", "ffdnet_picard_180_first"):
        with open(os.path.join(GOLDEN, f"e2e_{tag}.json")) as fh:
            meta = json.load(fh)
        assert meta["measurements"] and all("psnr" in m and "sha16_clip" in m for m in meta["measurements"])


def test_weight_packing_layouts_on_cpu():
    """The host-side weight re-orderings the HIP conv kernels consume, checked without a GPU: unpack every packed
    layout back by its documented index formula and (for Winograd) run the F(2x2,3x3) algebra in torch."""
    import torch.nn.functional as Fn
    g = torch.Generator().manual_seed(21)
    w = torch.randn(64, 64, 3, 3, generator=g)
    U = _hip.pack_winograd_weights(w)                     # [c][xi][wn][q][i][j][s]
    assert tuple(U.shape) == (8, 16, 2, 4, 16, 2, 2)
    G = torch.tensor([[1.0, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1.0]])
    Uref = G @ w @ G.t()                                  # (cout, cin, 4, 4)
    for (c, xi, wn, q, i, j, s_) in [(0, 0, 0, 0, 0, 0, 0), (3, 7, 1, 2, 5, 1, 1), (7, 15, 1, 3, 15, 1, 0), (5, 9, 0, 1, 8, 0, 1)]:
        cout, cin = 32 * wn + 16 * j + i, 8 * c + 2 * q + s_
        assert abs(float(U[c, xi, wn, q, i, j, s_]) - float(Uref[cout, cin, xi // 4, xi % 4])) < 1e-6
    # Winograd algebra with the packed weights == conv2d (one 4x4 patch -> 2x2 outputs)
    x = torch.randn(1, 64, 4, 4, generator=g)
    Bt = torch.tensor([[1.0, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
    At = torch.tensor([[1.0, 1, 1, 0], [0, 1, -1, -1]])
    V = Bt @ x[0] @ Bt.t()                                # (cin, 4, 4)
    M = torch.einsum("oiab,iab->oab", Uref, V)
    Y = At @ M @ At.t()                                   # (cout, 2, 2)
    want = Fn.conv2d(x, w)[0]                             # valid conv of the 4x4 patch = the 2x2 outputs
    assert rel_l2(Y.numpy(), want.numpy()) < 1e-5
    wt = torch.randn(4, 64, 3, 3, generator=g)
    pt = _hip.pack_tail_weights(wt)                       # [half][tap][cin32][cout]
    assert tuple(pt.shape) == (2, 9, 32, 4) and float(pt[1, 5, 7, 2]) == float(wt[2, 32 + 7, 1, 2])
    wh = torch.randn(64, 5, 3, 3, generator=g)
    ph = _hip.pack_head_weights(wh)                       # [ch*9+tap][cout//4][cout%4]
    assert tuple(ph.shape) == (45, 16, 4) and float(ph[2 * 9 + 4, 3, 1]) == float(wh[13, 2, 1, 1])
    w1 = torch.randn(64, 1, 3, 3, generator=g)
    assert float(_hip.pack_c1_to_64_weights(w1)[7, 10, 2]) == float(w1[42, 0, 2, 1])
    w2 = torch.randn(1, 64, 3, 3, generator=g)
    assert float(_hip.pack_c64_to_1_weights(w2)[1, 3, 9]) == float(w2[0, 41, 1, 0])
    with pytest.raises(_hip.DeqsciHipError):
        _hip.pack_winograd_weights(torch.zeros(64, 32, 3, 3))




def test_pack_winograd44_weights_layout():
    """U = G g G^T of F(4x4,3x3) in the LDS order of csrc/winograd44.hip: element [c][s][rg][cgp][q][i][j][ks] is the weight of
    transform position (3 rg + s // 6, s % 6), cout 32 cgp + 16 j + i, cin 8 c + 2 q + ks."""
    import torch
    from deqsci_amd import _hip
    g = torch.Generator().manual_seed(4)
    w = torch.randn(64, 64, 3, 3, generator=g)
    U = _hip.pack_winograd44_weights(w).reshape(8, 18, 2, 2, 4, 16, 2, 2)
    G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
                     dtype=torch.float64)
    full = (G @ w.double() @ G.t()).float()                   # (cout, cin, 6, 6)
    for (c, s_, rg, cgp, q, i, j, ks) in ((0, 0, 0, 0, 0, 0, 0, 0), (7, 17, 1, 1, 3, 15, 1, 1), (3, 7, 0, 1, 2, 5, 0, 1), (5, 11, 1, 0, 1, 9, 1, 0)):
        assert U[c, s_, rg, cgp, q, i, j, ks] == full[32 * cgp + 16 * j + i, 8 * c + 2 * q + ks, 3 * rg + s_ // 6, s_ % 6]
    assert U.numel() == 36 * 64 * 64
    import pytest
    with pytest.raises(_hip.DeqsciHipError):
        _hip.pack_winograd44_weights(torch.zeros(64, 32, 3, 3))


def test_sigma_restart_survives_address_reuse():
    """a new measurement allocated at the freed address of the previous one must still restart sigma at
    60/255 (solvers/equilibrium_solvers_yaping.py:408-413 compares y.mean(), not tensor identity)."""
    solver, _ = build_pipeline("ffdnet", None, 10, device="cpu")
    starts, ptrs = [], []

    def one_measurement(seed):
        y = torch.rand(1, 64, 64, generator=torch.Generator().manual_seed(seed))      # function-local: freed on return
        ptrs.append(y.data_ptr())
        s0 = float(solver._sigma(y, 8)[0])
        for _ in range(3):
            solver._sigma(y, 8)                                                        # same tensor: decays, no restart
        starts.append(s0)
        return float(solver.noise_sigma[0])
    ends = [one_measurement(s) for s in range(6)]                                   # (the allocator usually recycles an address here - not always)
    assert all(abs(s - 60 / 255) < 1e-7 for s in starts), starts
    assert all(abs(e - float(sigma_schedule(4)[3])) < 1e-7 for e in ends)
    # the same measurement passed again as a NEW tensor with an equal mean keeps decaying, as in the reference
    y = torch.rand(1, 64, 64, generator=torch.Generator().manual_seed(5))
    assert abs(float(solver._sigma(y.clone(), 8)[0]) - float(sigma_schedule(5)[4])) < 1e-7
    # in-place modification of y is noticed
    y2 = torch.rand(1, 64, 64)
    solver._sigma(y2, 8)
    y2.mul_(0.5)
    assert abs(float(solver._sigma(y2, 8)[0]) - 60 / 255) < 1e-7
    # the same situation made deterministic: NEW tensor objects over ONE storage, i.e. a new measurement at the address of the previous one
    store = torch.empty(64 * 64)
    for seed in range(3):
        store.copy_(torch.rand(64 * 64, generator=torch.Generator().manual_seed(100 + seed)))
        y_new = store.view(1, 64, 64)                                                  # a fresh object, the same data_ptr every time
        assert y_new.data_ptr() == store.data_ptr()
        assert abs(float(solver._sigma(y_new, 8)[0]) - 60 / 255) < 1e-7, seed
        assert abs(float(solver._sigma(y_new, 8)[0]) - float(sigma_schedule(2)[1])) < 1e-7


def test_harness_clip_helpers():
    assert harness.scored_measurements("drop8_cacti.mat", 5) == [0]
    assert harness.scored_measurements("runner8_cacti.mat", 5) == [0]
    assert harness.scored_measurements("traffic_cacti.mat", 6) == [0, 1, 2, 3, 4, 5]
    ds = harness.SCITestDataset(orc.DATA_DIR)
    plain = harness.as_clip(ds[0])
    batched = harness.as_clip(next(iter(torch.utils.data.DataLoader(ds, batch_size=1))))
    for k in ("gt", "mask", "meas"):
        assert plain[k].dim() == 3 and torch.equal(plain[k], batched[k])
    assert plain["file"] == batched["file"] == "drop8_cacti.mat"
    rec = torch.rand(2, 4, 5, 8)
    r = harness.ClipResult(name="traffic_cacti.mat", rec=rec, psnr=[20.0, 30.0], res=[1e-3, 2e-3], frames=16)
    assert r.mean_psnr == 25.0
    pay = harness.png_payloads(r, "out/")
    assert len(pay) == 16 and "out/traffic_cacti.mat_reconstruction_15.png" in pay
    assert np.array_equal(pay["out/traffic_cacti.mat_reconstruction_9.png"][..., 0], rec[1, :, :, 1].numpy() * 255.)


def test_cli_accepts_every_reference_flag():
    """video_sci_proxgrad.py:23-49: a script that passes the training-only flags must still parse."""
    from deqsci_amd.cli import parser
    a = parser().parse_args(["--gpu_ids", "0,1", "--n_epochs", "3", "--batch_size", "2", "--and_maxiters", "180", "--and_beta", "0.9",
                             "--and_m", "4", "--lr", "1e-3", "--etainit", "0.9", "--lr_gamma", "0.5", "--sched_step", "2",
                             "--savepath", "s/", "--trainpath", "t/", "--testpath", "d/", "--loadpath", "m.ckpt", "--denoiser", "SimpleCNN",
                             "--inference", "True", "--print_every_n_steps", "1", "--save_every_n_steps", "5", "--sigma", "10"])
    assert a.and_maxiters == 180 and a.and_m == 4 and a.etainit == 0.9 and a.sigma == 10 and a.gpu_ids == "0,1"
    # this build's own flags: the batching of the harness (none = the reference's one-by-one schedule; bare = a clip per call; all = every clip
    # of one frame size in one call) and the arithmetic of alpha (the reference's by default)
    assert a.batch_measurements is None and a.anderson_arith == "reference"
    assert parser().parse_args(["--batch_measurements"]).batch_measurements == "clip"
    assert parser().parse_args(["--batch_measurements", "all", "--anderson_arith", "float64"]).batch_measurements == "all"
    with pytest.raises(SystemExit):
        parser().parse_args(["--anderson_arith", "reference-bmm"])
    with pytest.raises(NotImplementedError):
        cli_main(["--denoiser", "unet"])


def test_blk32_layout_helpers():
    """Blk32 (the F(4x4,3x3) kernel's own activation layout): host-side conversion both ways, padding columns, position table."""
    import torch
    from deqsci_amd import _hip
    x = torch.randn(2, 64, 5, 45)
    b = _hip.Blk32.from_nchw(x)
    assert tuple(b.t.shape) == (2, 8, 5, 2, 32, 8) and (b.n, b.H, b.W) == (2, 5, 45)
    assert torch.equal(b.to_nchw(), x.contiguous(memory_format=torch.channels_last))
    pos = _hip.Blk32._pos()
    assert sorted(pos.tolist()) == list(range(32)) and pos[:5].tolist() == [8, 16, 24, 0, 9]      # column m -> 8 ((m+1)&3) + ((m+1)>>2) - [(m+1)&3 == 0]
    # pixel (n=1, channel 13, row 3, column 37) -> chunk 1, channel 5 of the chunk, block 1, column 5 of the block
    assert float(b.t[1, 1, 3, 1, int(pos[5]), 5]) == float(x[1, 13, 3, 37])
    assert bool(torch.isnan(b.t[:, :, :, 1][:, :, :, pos][:, :, :, 13:]).all())                    # columns >= 45 of the last block: padding


def test_reference_pickle_checkpoint_formats(tmp_path):
    """the reference's own on-disk formats, not this build's .npz re-serialisations -
    (1) the training checkpoint `torch.save({'solver_state_dict', 'epoch', 'optimizer_state_dict', 'scheduler_state_dict'})`
    of training/sci_equilibrium_training.py:126-130, with the `module.` prefix nn.DataParallel leaves on every key
    (stripped at video_sci_proxgrad.py:217-222); (2) a bare FFDNet state dict with `module.` keys (networks/ffdnet/models/net_gray.pth);
    (3) a missing path raises instead of silently running with random weights (video_sci_proxgrad.py:211)."""
    cnn, _ = checkpoint.read_state_dict(checkpoint.shipped("cnn"))
    assert sorted(cnn) == [f"nonlinear_op.dncnn.{i}.weight" for i in (0, 2, 4, 6)]
    path = str(tmp_path / "cnn.ckpt")
    torch.save({"solver_state_dict": {"module." + k: v for k, v in cnn.items()}, "epoch": 7,
                "optimizer_state_dict": {"state": {}, "param_groups": []}, "scheduler_state_dict": {"last_epoch": 7}}, path)
    solver, _ = build_pipeline("SimpleCNN", None, 10, device="cpu")
    before = {k: v.clone() for k, v in solver.state_dict().items()}
    assert checkpoint.load_solver(solver, path) == 7
    after = solver.state_dict()
    assert sorted(after) == sorted(cnn) and all(torch.equal(after[k], cnn[k]) for k in cnn)
    assert any(not torch.equal(before[k], after[k]) for k in cnn)                      # the load really replaced the random init
    # the same checkpoint without the DataParallel prefix
    torch.save({"solver_state_dict": dict(cnn), "epoch": 3}, path)
    solver2, _ = build_pipeline("SimpleCNN", None, 10, device="cpu")
    assert checkpoint.load_solver(solver2, path) == 3
    assert all(torch.equal(solver2.state_dict()[k], cnn[k]) for k in cnn)
    # (2) bare denoiser pickle, keys `module.intermediate_dncnn...` -> loaded into solver.nonlinear_op
    ff, _ = checkpoint.read_state_dict(checkpoint.shipped("ffdnet_gray"))
    pth = str(tmp_path / "net_gray.pth")
    torch.save({"module." + k: v for k, v in ff.items()}, pth)
    sd, epoch = checkpoint.read_state_dict(pth)
    assert epoch is None and sorted(sd) == sorted(ff) and all(torch.equal(sd[k], ff[k]) for k in ff)
    fsolver, _ = build_pipeline("ffdnet", pth, 10, device="cpu")
    got = fsolver.nonlinear_op.state_dict()
    assert all(torch.equal(got[k], ff[k]) for k in ff)
    assert sorted(k for k in fsolver.state_dict()) == sorted("nonlinear_op." + k for k in got)   # keys are exactly nonlinear_op.* (S1)
    # (3)
    with pytest.raises(FileNotFoundError):
        checkpoint.load_solver(solver, str(tmp_path / "nope.ckpt"))


def test_conv64_policy_resolution():
    """DEQSCIEngine(conv64=...): "auto" = "fast" (split-fp16 direct convolution / Winograd F(2x2,3x3), the faster per launch) for every
    configuration; "fast32" restricts to fp32 MFMA arithmetic; explicit kernels are honoured; conv64_f22_calls is off unless asked for."""
    ff = build_denoiser("ffdnet").eval()
    cnn = build_denoiser("SimpleCNN").eval()
    pol = lambda net, **kw: (lambda e: (e.conv64_policy, e.conv64_f22_calls, e.den.f22_calls, e.den.conv64))(DEQSCIEngine(net, **kw))
    assert pol(ff) == ("fast", None, None, "fast") and pol(cnn, iterator="picard") == ("fast", None, None, "fast")
    assert pol(ff, conv64="fast32", conv64_f22_calls=40) == ("fast32", 40, 40, "fast32")
    for k in ("f22", "f44", "s16"):
        assert pol(ff, conv64=k) == (k, None, None, k)
    with pytest.raises(ValueError):
        DEQSCIEngine(ff, conv64="f33")
    for k in ("f22", "f44", "s16"):
        assert _hip.conv64_kernel_for(64, 128, 128, policy=k) == k and _hip.conv64_kernel_for(1, 16, 16, policy=k) == k
    with pytest.raises(_hip.DeqsciHipError):
        _hip.conv64_kernel_for(1, 16, 16, policy="direct")
    # the size rule (256 CUs): one block tile per CU and more -> the 16 x 32-tile kernel; below -> F(2x2,3x3); huge images -> F(2x2,3x3)
    _hip._CUS[0] = 256
    try:
        import torch
        dev = torch.device("cuda", 0)
        assert [_hip.conv64_kernel_for(n, 128, 128, dev, "fast") for n in (1, 4, 6, 8, 64)] == ["f22", "f22", "s16", "s16", "s16"]
        assert [_hip.conv64_kernel_for(n, 128, 128, dev, "fast32") for n in (1, 4, 6, 8, 64)] == ["f22", "f22", "f44", "f44", "f44"]
        assert _hip.conv64_kernel_for(2, 256, 256, dev, "fast") == "s16" and _hip.conv64_kernel_for(1, 256, 256, dev, "fast") == "f22"
        assert _hip.conv64_kernel_for(64, 2900, 2900, dev, "fast") == "f22"
    finally:
        _hip._CUS.pop(0, None)


def test_split16_weight_pack_layout():
    """Split16Weights: hi + lo == 2^sw w to 2^-22, max |2^sw w| in [2^13, 2^14), and the LDS order of csrc/conv_s16.hip
    [chunk][tap][piece][cout group][lane][j] with cout = 32 g + lane % 32, cin = 16 c + 8 (lane // 32) + j."""
    g = torch.Generator().manual_seed(3)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.07
    W = _hip.Split16Weights(w)
    ws = w * 2.0 ** W.sw
    assert 2 ** 13 <= float(ws.abs().max()) < 2 ** 14
    assert tuple(W.packed.shape) == (4, 9, 2, 2, 2, 32, 8) and W.packed.dtype == torch.float16
    rec = W.packed[:, :, 0].float() + W.packed[:, :, 1].float()                       # [c][tap][g][kb][m][j]
    for (c, tap, gq, kb, m, j) in ((0, 0, 0, 0, 0, 0), (3, 8, 1, 1, 31, 7), (2, 4, 1, 0, 5, 3), (1, 7, 0, 1, 17, 6)):
        cout, cin = 32 * gq + m, 16 * c + 8 * kb + j
        want = float(ws[cout, cin, tap // 3, tap % 3])
        assert abs(float(rec[c, tap, gq, kb, m, j]) - want) <= 2.0 ** -21 * abs(want) + 1e-30
    assert W.sw == _hip._weight_exp(w) and _hip._weight_exp(torch.zeros(2)) == 0
    rel = float(((rec.permute(2, 4, 0, 3, 5, 1).reshape(64, 64, 9) - ws.reshape(64, 64, 9)).norm()) / ws.norm())
    assert rel < 2e-7
