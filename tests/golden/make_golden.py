#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference
(/root/reference, CPU, with tests/golden/ref_shims.py) and running its own code.

Runs only in the build container (the GPU box has no /root/reference).  Nothing in this
file is product code and nothing from the reference's source text is stored: only
seeded inputs and the tensors / scalars the reference computes from them.

    python tests/golden/make_golden.py g1 g2 g3 g4 g6        # seconds .. a minute
    python tests/golden/make_golden.py g5:SimpleCNN:10 ...   # end-to-end harness runs
    python tests/golden/make_golden.py weights               # re-serialise checkpoints

Groups:
  g1  ops.npz            A_torch_/At_torch_/initial_point/Phi_sum/GAP step
  g2  anderson_toy.npz   andersonexp + forward_iteration on a seeded contractive toy f
  g3  nets.npz           FFDNet(net_gray) / SimpleCNN(cnn.ckpt) single forwards
  g4  trace_*.npz        teacher-forced f trace, traffic m0 64x64 crop, and_maxiters=10
  g5  e2e_*.npz/.json    the reference's real test_solver_sci over data/test_gray
  g6  sigma.npz          FFDNet sigma sequence (repeated fp32 multiply)
  g7  admm_toy.npz       ADMM variant on a toy denoiser
  g9  training_toy.npz   the reference's train_solver_sci for 2 epochs x 3 steps on seeded toy batches
  g8  backward.npz       training-mode DEQFixedPoint: implicit-differentiation gradients (SimpleCNN, cnn.ckpt)
"""
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()

from networks.ffdnet.models import FFDNet  # noqa: E402
from networks.provable.model.SimpleCNN_models import DnCNN  # noqa: E402
from solvers.equilibrium_solvers_yaping import EquilibriumProxGradSCI  # noqa: E402
from solvers import new_equilibrium_utils_yaping as eq_utils  # noqa: E402
from utils.cg_utils import A_torch_, At_torch_, initial_point  # noqa: E402
from utils.sci_dataloader import SCITestDataset, load_test_data  # noqa: E402
from training import sci_equilibrium_training as sci_train  # noqa: E402

REF = ref_shims.REFERENCE_ROOT
DATA = REF + "/data/test_gray/"


def sha16(t):
    return hashlib.sha256(np.ascontiguousarray(t, dtype=np.float32).tobytes()).hexdigest()[:16]


def build_denoiser(name):
    """video_sci_proxgrad.py:163,175-177,191 + checkpoint load :211-223."""
    if name == "ffdnet":
        net = FFDNet(num_input_channels=1, tag="ffdnet")
        sd = torch.load(REF + "/networks/ffdnet/models/net_gray.pth", map_location="cpu",
                        weights_only=False)
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        net.load_state_dict(sd)
    elif name == "SimpleCNN":
        net = DnCNN(1, num_of_layers=4, lip=0.0, no_bn=True, tag="denoiser")
    else:
        raise NotImplementedError(name)
    net.eval()
    return net


def build_solver(name):
    net = build_denoiser(name)
    solver = EquilibriumProxGradSCI(A=A_torch_, At=At_torch_, nonlinear_operator=net,
                                    eta=0.2, minval=-1, maxval=1)
    if name == "SimpleCNN":
        sd = torch.load(REF + "/models/cnn.ckpt", map_location="cpu",
                        weights_only=False)["solver_state_dict"]
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        solver.load_state_dict(sd)
    return solver


def build_deq(name, iters, iterator="anderson"):
    solver = build_solver(name)
    if iterator == "anderson":
        deq = eq_utils.DEQFixedPoint(solver, eq_utils.andersonexp, m=5, beta=1.0, lam=1e-2,
                                     max_iter=iters, tol=1e-5)
    else:
        deq = eq_utils.DEQFixedPoint(solver, eq_utils.forward_iteration, max_iter=iters, tol=1e-5)
    return solver, deq


# --------------------------------------------------------------------------- g1
class _ZeroNoise(torch.nn.Module):
    tag = "denoiser"

    def forward(self, x):
        return torch.zeros_like(x)


def g1():
    out = {}
    g = torch.Generator().manual_seed(20240101)
    for ci, (bsz, H, W, B) in enumerate([(2, 16, 16, 8), (1, 32, 32, 16), (3, 8, 12, 5)]):
        Phi = (torch.rand(bsz, H, W, B, generator=g) < 0.5).float()
        Phi[:, 0, :3, :] = 0.0                     # pixels with Phi_sum == 0
        x = torch.rand(bsz, H, W, B, generator=g)
        z = torch.randn(bsz, H, W, B, generator=g)
        y = A_torch_(x, Phi)
        Phi_sum = torch.sum(Phi, axis=3)           # sci_equilibrium_training.py:162-163
        Phi_sum[Phi_sum == 0] = 1
        solver = EquilibriumProxGradSCI(A=A_torch_, At=At_torch_, nonlinear_operator=_ZeroNoise(),
                                        eta=0.2)
        z1 = solver(z, y, Phi, Phi_sum)            # GAP step only (noise == 0)
        p = f"c{ci}_"
        out.update({p + "Phi": Phi, p + "x": x, p + "z": z, p + "y": y, p + "Phi_sum": Phi_sum,
                    p + "Az": A_torch_(z, Phi), p + "Aty": At_torch_(y, Phi),
                    p + "x0": initial_point(y, Phi, Phi_sum, None), p + "z1": z1})
    # non-binary (grey) mask: the reference still divides by sum(Phi)
    Phi = torch.rand(1, 8, 8, 8, generator=g)
    z = torch.randn(1, 8, 8, 8, generator=g)
    y = torch.rand(1, 8, 8, generator=g) * 4
    Phi_sum = torch.sum(Phi, axis=3)
    Phi_sum[Phi_sum == 0] = 1
    solver = EquilibriumProxGradSCI(A=A_torch_, At=At_torch_, nonlinear_operator=_ZeroNoise(), eta=0.2)
    out.update({"grey_Phi": Phi, "grey_z": z, "grey_y": y, "grey_Phi_sum": Phi_sum,
                "grey_z1": solver(z, y, Phi, Phi_sum)})
    np.savez_compressed(HERE + "/ops.npz", **{k: v.numpy() for k, v in out.items()})
    print("g1 ->", HERE + "/ops.npz")


# --------------------------------------------------------------------------- g2
def toy_f_factory(a, c):
    """Contractive elementwise toy map used by g2 (restated in tests): f(z)=a*z+0.3*sin(z)+c."""
    def f(z):
        return a * z + 0.3 * torch.sin(z) + c
    return f


def g2():
    out = {}
    g = torch.Generator().manual_seed(777)
    for bsz in (1, 3):
        shape = (bsz, 6, 10, 4)
        a = torch.rand(shape, generator=g) * 0.5
        c = torch.randn(shape, generator=g)
        x0 = torch.randn(shape, generator=g)
        out[f"b{bsz}_a"], out[f"b{bsz}_c"], out[f"b{bsz}_x0"] = a, c, x0
        for max_iter in (3, 7, 12, 40):
            fed = []
            base = toy_f_factory(a, c)

            def f(z):
                fed.append(z.clone())
                return base(z)
            zs, res = eq_utils.andersonexp(f, x0, m=5, lam=1e-2, max_iter=max_iter, tol=1e-5, beta=1.0)
            p = f"b{bsz}_it{max_iter}_"
            out[p + "z"] = zs.clone()
            out[p + "res"] = torch.tensor(res, dtype=torch.float64)
            out[p + "fed"] = torch.stack(fed)
        # early stop: loose tol so the break at :186-187 fires
        fed = []

        def f2(z):
            fed.append(z.clone())
            return base(z)
        zs, res = eq_utils.andersonexp(f2, x0, m=5, lam=1e-2, max_iter=40, tol=1e-3, beta=1.0)
        out[f"b{bsz}_early_z"] = zs.clone()
        out[f"b{bsz}_early_res"] = torch.tensor(res, dtype=torch.float64)
        out[f"b{bsz}_early_ncalls"] = torch.tensor(len(fed))
        # beta != 1 and m != 5
        zs, res = eq_utils.andersonexp(base, x0, m=3, lam=1e-3, max_iter=9, tol=1e-5, beta=0.7)
        out[f"b{bsz}_m3beta_z"] = zs.clone()
        out[f"b{bsz}_m3beta_res"] = torch.tensor(res, dtype=torch.float64)
        # Picard
        zs, res = eq_utils.forward_iteration(base, x0, max_iter=15, tol=1e-5)
        out[f"b{bsz}_picard_z"] = zs.clone()
        out[f"b{bsz}_picard_res"] = torch.tensor(res, dtype=torch.float64)
        zs, res = eq_utils.forward_iteration(base, x0, max_iter=60, tol=1e-3)
        out[f"b{bsz}_picard_early_z"] = zs.clone()
        out[f"b{bsz}_picard_early_res"] = torch.tensor(res, dtype=torch.float64)
    np.savez_compressed(HERE + "/anderson_toy.npz", **{k: v.numpy() for k, v in out.items()})
    print("g2 ->", HERE + "/anderson_toy.npz")


# --------------------------------------------------------------------------- g3
def g3():
    out = {}
    d = load_test_data(DATA + "traffic_cacti.mat")
    x = torch.from_numpy(d["gt"][96:128, 64:96, :8]).permute(2, 0, 1)[:, None].contiguous()  # (8,1,32,32)
    g = torch.Generator().manual_seed(5)
    x = x + 0.1 * torch.randn(x.shape, generator=g)
    out["x"] = x
    ff = build_denoiser("ffdnet")
    cnn = build_solver("SimpleCNN").nonlinear_op
    sig = torch.FloatTensor([60 / 255]).expand(8)
    with torch.no_grad():
        for k in range(51):
            if k in (0, 1, 50):
                out[f"ffdnet_sigma_k{k}"] = sig.clone()
                out[f"ffdnet_noise_k{k}"] = ff(x, sig)
            sig = sig * 0.971
        out["cnn_noise"] = cnn(x)
    np.savez_compressed(HERE + "/nets.npz", **{k: v.numpy() for k, v in out.items()})
    print("g3 ->", HERE + "/nets.npz")


# --------------------------------------------------------------------------- g4
def g4():
    d = load_test_data(DATA + "traffic_cacti.mat")
    sl = (slice(96, 160), slice(64, 128))
    Phi = torch.from_numpy(d["mask"][sl])[None]
    y = torch.from_numpy(d["meas"][sl][..., 0])[None]
    gt = torch.from_numpy(d["gt"][sl][..., :8])[None]
    Phi_sum = torch.sum(Phi, axis=3)
    Phi_sum[Phi_sum == 0] = 1
    for name in ("SimpleCNN", "ffdnet"):
        solver, deq = build_deq(name, 10)
        fed, ret, sig = [], [], []
        orig_forward = solver.forward

        def traced(z, yy, P, Ps, _o=orig_forward):
            fed.append(z.detach().clone())
            r = _o(z, yy, P, Ps)
            ret.append(r.detach().clone())
            sig.append(solver.noise_sigma.clone())
            return r
        solver.forward = traced
        x0 = initial_point(y, Phi, Phi_sum, gt)
        rec = deq.forward(y, Phi, Phi_sum, initial_point=x0, train_flag=False)
        np.savez_compressed(
            HERE + f"/trace_{name}.npz", Phi=Phi.numpy(), y=y.numpy(), gt=gt.numpy(),
            Phi_sum=Phi_sum.numpy(), x0=x0.numpy(), fed=torch.stack(fed).numpy(),
            ret=torch.stack(ret).numpy(), sigma=torch.stack(sig).numpy(),
            rec=rec.detach().numpy(), res=np.float64(deq.forward_res))
        print("g4 ->", name, "calls", len(fed), "res", deq.forward_res)


# --------------------------------------------------------------------------- g6
def g6():
    sig = torch.FloatTensor([60 / 255]).expand(8)
    seq = [sig[0].item()]
    for _ in range(200):
        sig = sig * 0.971
        seq.append(sig[0].item())
    np.savez(HERE + "/sigma.npz", sigma=np.array(seq, dtype=np.float32))
    print("g6 ->", seq[0] * 255, seq[181] * 255)


# --------------------------------------------------------------------------- g5
KEEP_REC = {  # (denoiser, iterator, iters) -> measurement ids whose full rec is stored
    ("SimpleCNN", "anderson", 180): ("drop8_cacti.mat:0", "runner8_cacti.mat:0", "traffic_cacti.mat:0"),
    ("ffdnet", "picard", 180): ("traffic_cacti.mat:0",),
    ("ffdnet", "anderson", 30): ("traffic_cacti.mat:0",),
    ("ffdnet", "anderson", 10): ("traffic_cacti.mat:0", "drop8_cacti.mat:0"),
    ("SimpleCNN", "anderson", 10): ("traffic_cacti.mat:3",),
}


def g5(name, iterator, iters, only_first=False):
    """Drive the reference's real harness (training/sci_equilibrium_training.py:152-205)."""
    solver, deq = build_deq(name, iters, iterator)
    ncalls = [0]
    orig_forward = solver.forward

    def counted(*a, _o=orig_forward):
        ncalls[0] += 1
        return _o(*a)
    solver.forward = counted

    log, recs = [], {}
    orig_deq_forward = deq.forward
    cur = {"file": None, "fi": 0}

    class Loader:
        """DataLoader(batch_size=1, shuffle=False, drop_last=True) over SCITestDataset
        (video_sci_proxgrad.py:138-141), tracking which file is being processed."""
        def __init__(self):
            self.dl = torch.utils.data.DataLoader(dataset=SCITestDataset(DATA), batch_size=1,
                                                  shuffle=False, drop_last=True)

        def __iter__(self):
            for b in self.dl:
                if only_first and "traffic" not in b["file"][0]:
                    continue
                cur["file"], cur["fi"] = b["file"][0], 0
                if only_first:
                    b["meas"] = b["meas"][..., :1]
                    b["gt"] = b["gt"][..., :8]
                yield b

    def deq_forward(y, Phi, Phi_sum, initial_point=None, train_flag=True):
        t0 = time.time()
        c0 = ncalls[0]
        rec = orig_deq_forward(y, Phi, Phi_sum, initial_point=initial_point, train_flag=train_flag)
        res = deq.forward_res
        if isinstance(res, list):
            res = res[-1]
        mid = f"{cur['file']}:{cur['fi']}"
        r = rec.detach().numpy()
        log.append({"id": mid, "res": float(res), "f_calls": ncalls[0] - c0,
                    "sha16_clip": sha16(r.clip(0, 1)), "seconds": time.time() - t0})
        if mid in KEEP_REC.get((name, iterator, iters), ()):
            recs[mid.replace(".mat:", "_m").replace("_cacti", "")] = r
        cur["fi"] += 1
        return rec
    deq.forward = deq_forward

    ref_shims.PSNR_LOG.clear()
    t0 = time.time()
    avg, images = sci_train.test_solver_sci(deq, test_dataloader=Loader(), save_img_path="",
                                            verbose=True, save_image=False)
    wall = time.time() - t0
    for e, p in zip(log, ref_shims.PSNR_LOG):
        e["psnr"] = p
    meta = {"denoiser": name, "iterator": iterator, "and_maxiters": iters, "avg_psnr": float(avg),
            "n_png_payloads": len(images), "wall_s": wall, "threads": torch.get_num_threads(),
            "torch": torch.__version__, "measurements": log,
            "weights": "net_gray.pth (substitute for missing ffdnet.ckpt)" if name == "ffdnet" else "cnn.ckpt"}
    tag = f"{name}_{iterator}_{iters}" + ("_first" if only_first else "")
    with open(HERE + f"/e2e_{tag}.json", "w") as fh:
        json.dump(meta, fh, indent=1)
    if recs:
        np.savez_compressed(HERE + f"/e2e_{tag}_rec.npz", **recs)
    print("g5 ->", tag, "avg", avg, "wall", wall)


def weights():
    """Re-serialise the shipped checkpoints as plain tensor archives (data, not code)."""
    wd = os.path.join(os.path.dirname(os.path.dirname(HERE)), "deqsci_amd", "weights")
    os.makedirs(wd, exist_ok=True)
    ck = torch.load(REF + "/models/cnn.ckpt", map_location="cpu", weights_only=False)
    np.savez(wd + "/cnn.npz", __epoch__=np.int64(ck["epoch"]),
             **{k: v.numpy() for k, v in ck["solver_state_dict"].items()})
    sd = torch.load(REF + "/networks/ffdnet/models/net_gray.pth", map_location="cpu", weights_only=False)
    np.savez(wd + "/ffdnet_gray.npz", **{k: v.numpy() for k, v in sd.items()})
    ck = torch.load(REF + "/models/rsn_cnn.ckpt", map_location="cpu", weights_only=False)
    np.savez(wd + "/rsn_cnn.npz", __epoch__=np.int64(ck["epoch"]),
             **{k: v.numpy() for k, v in ck["solver_state_dict"].items()})
    print("weights ->", wd)




# --------------------------------------------------------------------------- g7 (ADMM variant)
class _ToyDenoiser(torch.nn.Module):
    conv3d = False

    def forward(self, x):
        return 0.8 * x + 0.02 * torch.tanh(x)


def g7():
    from solvers.equilibrium_solvers_yaping import EquilibriumADMMSCI
    from utils.cg_utils import initial_point_admm
    g = torch.Generator().manual_seed(99)
    Phi = (torch.rand(2, 12, 16, 8, generator=g) < 0.5).float()
    Phi[:, 0, :2, :] = 0
    x = torch.rand(2, 12, 16, 8, generator=g)
    y = A_torch_(x, Phi)
    Phi_sum = torch.sum(Phi, axis=3)
    Phi_sum[Phi_sum == 0] = 1
    f = EquilibriumADMMSCI(A=A_torch_, At=At_torch_, nonlinear_operator=_ToyDenoiser(), eta=0.2)
    init = initial_point_admm(y, Phi, Phi_sum, None)
    out = {"Phi": Phi, "y": y, "Phi_sum": Phi_sum, "x0": init[0], "u0": init[1]}
    z1, u1 = f(init[0], init[1], y, Phi, Phi_sum)
    out["step_z"], out["step_u"] = z1, u1
    for it, tol in ((8, 1e-9), (40, 5e-2)):
        deq = eq_utils.DEQFixedPointADMM(f, eq_utils.admmexp, None, max_iter=it, tol=tol)
        z = deq.forward(y, Phi, Phi_sum, initial_point=[init[0].clone(), init[1].clone()])
        out[f"it{it}_z"] = z
        out[f"it{it}_res"] = torch.tensor(deq.forward_res, dtype=torch.float64)
    np.savez_compressed(HERE + "/admm_toy.npz", **{k: v.numpy() for k, v in out.items()})
    print("g7 ->", HERE + "/admm_toy.npz")


def g8(kind="SimpleCNN"):
    """Training-mode DEQFixedPoint (new_equilibrium_utils_yaping.py:241-281): forward solve without tape, one taped f call,
    implicit-differentiation backward hook solved with the same Anderson settings; MSE loss as in
    training/sci_equilibrium_training.py:69; gradients of every denoiser parameter."""
    g = torch.Generator().manual_seed(2024)
    bsz, H, W, B = 2, 24, 20, 4
    Phi = (torch.rand(bsz, H, W, B, generator=g) < 0.5).float()
    Phi[:, 0, :2, :] = 0
    gt = torch.rand(bsz, H, W, B, generator=g)
    y = A_torch_(gt, Phi)
    Phi_sum = torch.sum(Phi, axis=3)
    Phi_sum[Phi_sum == 0] = 1
    solver = build_solver(kind)
    for p in solver.parameters():
        p.requires_grad_(True)
    deq = eq_utils.DEQFixedPoint(solver, eq_utils.andersonexp, m=5, beta=1.0, lam=1e-2, max_iter=12, tol=1e-9)
    init = initial_point(y, Phi, Phi_sum, gt)
    rec = deq(y, Phi, Phi_sum, initial_point=init)
    loss = torch.nn.MSELoss()(rec, gt)
    solver.zero_grad()
    loss.backward()
    out = {"Phi": Phi, "gt": gt, "y": y, "Phi_sum": Phi_sum, "rec": rec.detach(), "loss": loss.detach().double(),
           "forward_res": torch.tensor(deq.forward_res, dtype=torch.float64),
           "backward_res": torch.tensor(deq.backward_res, dtype=torch.float64)}
    for name, p in solver.named_parameters():
        out["grad." + name] = p.grad.detach()
    if kind == "ffdnet":
        out["sigma_after"] = solver.noise_sigma.detach().clone()
    fn = HERE + ("/backward.npz" if kind == "SimpleCNN" else f"/backward_{kind}.npz")
    np.savez_compressed(fn, **{k: v.numpy() for k, v in out.items()})
    print("g8 ->", fn, "loss", float(loss), "fwd res", deq.forward_res, "bwd res", deq.backward_res,
          {k: float(v.norm()) for k, v in out.items() if k.startswith("grad.")})


def _toy_training_data(seed=77, steps=3, bsz=2, H=24, W=20, B=8):
    g = torch.Generator().manual_seed(seed)
    train = []
    for _ in range(steps):
        Phi = (torch.rand(bsz, H, W, B, generator=g) < 0.5).float()
        gt = torch.rand(bsz, H, W, B, generator=g)
        train.append({"gt": gt, "meas": A_torch_(gt, Phi), "mask": Phi})
    Phi = (torch.rand(1, H, W, B, generator=g) < 0.5).float()
    gt = torch.rand(1, H, W, B, generator=g)
    test = [{"gt": gt, "meas": A_torch_(gt, Phi).unsqueeze(3), "mask": Phi, "file": ["toy"]}]
    return train, test


def g9():
    """The reference's own training loop (training/sci_equilibrium_training.py:28-150) for 2 epochs of 3 Adam steps on seeded
    toy batches (SimpleCNN from cnn.ckpt, lr 1e-4, StepLR(1, 0.9), MSE mean, Anderson m=5 max_iter=8): loss of every step,
    the PSNR the loop prints, the weights after training, what the epoch checkpoint holds."""
    import tempfile

    class _Writer:                                  # torch.utils.tensorboard is not installed here
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

        def flush(self):
            pass
    sci_train.tensorboard.SummaryWriter = _Writer
    train, test = _toy_training_data()
    solver = build_solver("SimpleCNN")
    w0 = {k: v.detach().clone() for k, v in solver.state_dict().items()}
    deq = eq_utils.DEQFixedPoint(solver, eq_utils.andersonexp, m=5, beta=1.0, lam=1e-2, max_iter=8, tol=1e-5)
    opt = torch.optim.Adam(params=solver.parameters(), lr=1e-4)
    sched = torch.optim.lr_scheduler.StepLR(optimizer=opt, step_size=1, gamma=0.9)
    losses = []
    mse = torch.nn.MSELoss(reduction="mean")

    def loss_fn(rec, gt):
        l = mse(rec, gt)
        losses.append(float(l.detach()))
        return l
    tmp = tempfile.mkdtemp() + "/"
    ref_shims.PSNR_LOG.clear()
    sci_train.train_solver_sci(single_iterate_solver=solver, train_dataloader=train, test_dataloader=test, optimizer=opt,
                               save_model_path=tmp, deep_eq_module=deq, loss_function=loss_fn, n_epochs=2, scheduler=sched,
                               print_every_n_steps=1, save_every_n_steps=1000, start_epoch=0, train_img_path=tmp,
                               test_img_path=tmp, best_img_path=tmp, tflog_path=tmp)
    ck = torch.load(tmp + "epoch_1.ckpt", map_location="cpu", weights_only=False)
    out = {"losses": np.array(losses, dtype=np.float64), "psnr_log": np.array(ref_shims.PSNR_LOG, dtype=np.float64),
           "ckpt_epoch": np.array(ck["epoch"]), "lr_after": np.array(opt.param_groups[0]["lr"], dtype=np.float64)}
    assert sorted(ck.keys()) == ["epoch", "optimizer_state_dict", "scheduler_state_dict", "solver_state_dict"]
    for i, b in enumerate(train):
        out[f"train{i}.gt"], out[f"train{i}.mask"] = b["gt"].numpy(), b["mask"].numpy()
    out["test.gt"], out["test.mask"] = test[0]["gt"].numpy(), test[0]["mask"].numpy()
    for k, v in solver.state_dict().items():
        out["w." + k] = v.detach().numpy()
        out["dw." + k] = (v.detach() - w0[k]).numpy()
        assert torch.equal(ck["solver_state_dict"][k], v)
    np.savez_compressed(HERE + "/training_toy.npz", **out)
    print("g9 ->", HERE + "/training_toy.npz", losses, ref_shims.PSNR_LOG)


if __name__ == "__main__":
    torch.manual_seed(0)
    for arg in sys.argv[1:]:
        if arg.startswith("g8:"):
            g8(arg.split(":")[1])
        elif arg.startswith("g5"):
            parts = arg.split(":")
            g5(parts[1], parts[2], int(parts[3]), only_first=(len(parts) > 4 and parts[4] == "first"))
        else:
            globals()[arg]()
