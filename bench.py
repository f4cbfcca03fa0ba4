#!/usr/bin/env python3
"""bench.py - headline metric of BASELINE.json on MI355X: reconstructed frames/s at 256x256x8,
180 DEQ (Anderson) iterations, FFDNet denoiser, plus the MFMA roofline of the dominant kernel (the
denoiser's 64->64 layers: split-fp16 arithmetic on the f16 matrix cores under Winograd F(2,3) along x nested in the direct sum along y - as
the STACK launch, one launch of deqsci::w16::conv_w16_kernel<1> per 13 layers and slice of 32 images, csrc/conv_w16.hip; --stack-kernel s16:
round 4's direct form, deqsci::s16::conv_s16_kernel<0, 0, 1>; --no-stack: one launch per layer), the HBM roofline of the fused
Phi/Phi^T + GAP-update kernel, a parity spot check of the very configuration that is timed against the CPU oracle, the other
configurations a reader asks about (one measurement per call, the shipped clips of BASELINE config 2 with their PSNR, the fp32-only policy,
the exact-Gram arithmetic, one stream instead of two: `summary`, the LAST key of the line), and the reference algorithm timed on the host CPU.

    python bench.py --gpus N --steps K --warmup W          # N > 1: starts N rank processes itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full reconstruction pass (x0 = Phi^T y, max_iter+1 f-calls, final all-gather) over
one synthetic batch of `--batch-per-gpu` measurements per GPU (SURVEY 8(d) C3 recipe: Bernoulli(0.5)
masks, x ~ U[0,1), y = Phi x, seed 1234), inputs resident in HBM when the timed region starts.
Weak scaling (default): the per-GPU batch is fixed (BASELINE config 3: 64 measurements over 8 GPUs = 8 per GPU,
which is also the 8 shipped measurements of config 2 at N=1).  Strong scaling: `--global-batch 64` fixes the
total instead (64/N per GPU, BASELINE config 3 as stated).  Either way measurement i of the global batch is
generated from seed (1234, i) by the rank that owns it - no rank ever holds the global batch - and the shards go
through deqsci_amd.distributed.reconstruct_shard (contiguous slices, no data-path collective, ONE all-gather).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from deqsci_amd import distributed  # noqa: E402  (no GPU work at import)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3
MFMA_F16_PEAK_TFLOPS = 2500.0     # dense fp16 / bf16 MFMA (the 5 PF headline figure includes 2:1 sparsity)


def mix_gap_bytes(bsz, H, W, B, n):
    """Algorithmic bytes of one launch of the fused K7+K3 kernel: per pixel read
    n history rows + Phi (4B each per frame) + y + Phi_sum, write X_k and z1."""
    return bsz * H * W * (4 * B * (n + 3) + 8)


def gap_bytes(bsz, H, W, B):
    """K3 = 12B+8 bytes per pixel."""
    return bsz * H * W * (12 * B + 8)


def make_batch(lo, hi, H, W, B, seed, device):
    """Measurements [lo, hi) of the synthetic global batch (Phi ~ Bernoulli(0.5), x ~ U[0,1), y = Phi x).  Each
    measurement has its own generator seeded by its GLOBAL index, so a rank builds exactly its slice and the batch does not
    depend on how it is sharded."""
    n = max(hi - lo, 0)
    Phi = torch.empty(n, H, W, B, device=device)
    x = torch.empty(n, H, W, B, device=device)
    for j in range(n):
        g = torch.Generator(device=device).manual_seed(seed * 1000003 + lo + j)
        Phi[j] = (torch.rand(H, W, B, device=device, generator=g) < 0.5).float()
        x[j] = torch.rand(H, W, B, device=device, generator=g)
    y = (x * Phi).sum(3)
    return y, Phi, x


def parity_spot_check(args, dev, H, W, B, n_meas=2, iters=10):
    """Outside the timed region: the first `n_meas` measurements of the synthetic batch that is timed (same seeds, same size, same
    denoiser), `iters` Anderson iterations on a fresh default engine (its 64->64 layers on the kernel the policy picks at this size:
    named in the result), against the CPU oracle's restatement of the reference (video_sci_proxgrad.py:210-245 wiring)."""
    from oracle import deqsci_oracle as orc
    from deqsci_amd import _hip
    saved = args.iters
    args.iters = iters
    try:
        eng = build_engine(args, dev)
    finally:
        args.iters = saved
    y, Phi, _ = make_batch(0, n_meas, H, W, B, 1234, dev)
    kinds = set()
    hook, _hip.CONV64_EVENT_HOOK = _hip.CONV64_EVENT_HOOK, (lambda k, n, h, w, layers=1: kinds.add(k))
    try:
        rec = eng.reconstruct(y, Phi).cpu()
    finally:
        _hip.CONV64_EVENT_HOOK = hook
    yc, Pc = y.cpu(), Phi.cpu()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    worst = 0.0
    for i in range(n_meas):
        Ps = orc.phi_sum(Pc[i:i + 1])
        want, _ = orc.deq_forward(orc.ProxGradSCI(args.denoiser), orc.andersonexp, yc[i:i + 1], Pc[i:i + 1], Ps, orc.initial_point(yc[i:i + 1], Pc[i:i + 1]),
                                  m=5, beta=1.0, lam=1e-2, max_iter=iters, tol=1e-5)
        worst = max(worst, float((rec[i:i + 1] - want).norm() / want.norm()))
    return {"rel_l2": worst, "bound": 1e-4, "ok": worst < 1e-4, "conv64_kernels": sorted(kinds),
            "what": f"measurements 0..{n_meas - 1} of the timed synthetic batch ({H}x{W}x{B}, {args.denoiser}), and_maxiters={iters}, default engine "
                    f"vs the CPU oracle (worst rel-L2 over the measurements; chaos-free horizon, SURVEY F9)"}


def cpu_baseline(iters_sample, full_calls, H, W, B, kind):
    """The CPU oracle (restatement of the reference algorithm, torch-CPU fp32) on ONE synthetic
    measurement for `iters_sample` Anderson iterations; per-f-call cost scaled to the full run."""
    from oracle import deqsci_oracle as orc
    g = torch.Generator().manual_seed(1234)
    Phi = (torch.rand(1, H, W, B, generator=g) < 0.5).float()
    x = torch.rand(1, H, W, B, generator=g)
    y = orc.sci_forward(x, Phi)
    Ps = orc.phi_sum(Phi)
    threads = min(16, os.cpu_count() or 1)              # best of an 8..128 thread sweep on the 128-core GPU host
    torch.set_num_threads(threads)
    f = orc.ProxGradSCI(kind)
    f(orc.initial_point(y, Phi), y, Phi, Ps)           # warm-up call (oneDNN primitive creation)
    f.calls = 0
    t0 = time.perf_counter()
    orc.deq_forward(f, orc.andersonexp, y, Phi, Ps, orc.initial_point(y, Phi), m=5, beta=1.0, lam=1e-2,
                    max_iter=iters_sample, tol=1e-5)
    dt = time.perf_counter() - t0
    per_call = dt / f.calls
    fps = B / (per_call * full_calls)
    return {"value": fps, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 synthetic {H}x{W}x{B} measurement, {kind}, and_maxiters={iters_sample} ({f.calls} f-calls, "
                      f"{dt:.1f} s) scaled per f-call to the {full_calls} f-calls the GPU step executes (the reference runs "
                      f"one more, dead, call: new_equilibrium_utils_yaping.py:271-272)"}


class _PlumbingStub:
    """--plumbing-selftest only: stands in for DEQSCIEngine so that the launcher, the rendezvous, the sharding and the
    all-gather can be exercised on a box without GPUs (tests/test_distributed_gloo.py).  It reconstructs nothing
    (returns Phi^T y) and its output line says so; it is not a fallback of any product path."""
    m = 5

    def __init__(self, iters):
        self.last_info = {"f_calls": iters + 1, "res": None}

    def reconstruct(self, y, Phi):
        return (y.unsqueeze(-1) * Phi).contiguous()


def hbm_stream_roofline(H, W, B, m, dev, bsz=64, sets=3, budget_s=1.5):
    """Outside the timed frames/s region: the two fused streaming kernels of the DEQ loop on a working set far beyond
    the 256 MiB Infinity Cache (bsz 64, three rotating buffer sets), HIP-event timed, algorithmic bytes per launch
    against the 8 TB/s HBM3E peak (SURVEY 8(d) "making the GB/s honest")."""
    from deqsci_amd import _hip
    HWB, BHW = _hip.LAYOUT_HWB, _hip.LAYOUT_BHW
    N = H * W * B
    S = []
    for s in range(sets):
        g = torch.Generator(device=dev).manual_seed(77 + s)
        zp = torch.randn(bsz, B, H, W, device=dev, generator=g)
        Phip = (torch.rand(bsz, B, H, W, device=dev, generator=g) < 0.5).float()
        y = torch.rand(bsz, H, W, device=dev, generator=g) * 4
        Ps = _hip.phi_sum(Phip, BHW)
        ws = _hip.AndersonWorkspace(bsz, N, m, dev)
        ws.F.normal_(generator=g)
        ws.G.normal_(generator=g)
        ws.alpha[:, :m] = 1.0 / m
        S.append((zp, Phip, y, Ps, ws, torch.empty_like(zp), torch.empty_like(zp)))
    cases = {
        "gap_update_bhw (K3)": (gap_bytes(bsz, H, W, B),
                                lambda t: _hip.gap_update(t[0], t[1], t[2], t[3], BHW, BHW, out=t[5])),
        f"mix_gap_bhw n={m} (K7+K3)": (mix_gap_bytes(bsz, H, W, B, m),
                                       lambda t: _hip.anderson_mix_gap(t[4], 1.0, m, t[1], t[2], t[3], t[5], t[6], BHW)),
    }
    out = {}
    for name, (nbytes, fn) in cases.items():
        for i in range(2 * sets):
            fn(S[i % sets])
        torch.cuda.synchronize()
        launches = max(sets, int(budget_s / len(cases) / (nbytes / 5e12)) // sets * sets)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(launches):
            fn(S[i % sets])
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / launches
        out[name] = {"achieved": nbytes / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / t / 1e9 / HBM_PEAK_GBS,
                     "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": 1e6 * t, "launches_timed": launches}
    return {"workload": f"bsz {bsz} at {H}x{W}x{B}, {sets} rotating buffer sets (working set >> 256 MiB Infinity Cache)", "kernels": out}


def config2_shipped_clips(args, dev):
    """BASELINE configs[1] on the data it names: the 3 shipped clips = 8 measurements through the build's evaluation harness
    (deqsci_amd.harness.evaluate: upload, reconstruction, PSNR; file reading excluded) - one measurement per call (the reference's
    schedule, training/sci_equilibrium_training.py:171-181), a clip's measurements as one batch, and all eight as ONE batch (each with its
    own clip's mask: harness.reconstruct_clips_together) - with the average PSNR next to the
    reference's own run and its perturbation band (tests/golden/e2e_ffdnet_anderson_180*.json; FFDNet + Anderson at 180 iterations is
    chaotic on the traffic clip: the band, not the digit, is the comparison - DESIGN section 5)."""
    from deqsci_amd import checkpoint
    from deqsci_amd.cli import build_pipeline
    from deqsci_amd.harness import SCITestDataset, evaluate
    clips = list(SCITestDataset(os.path.join(ROOT, "data", "test_gray")))
    _, deq = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), args.iters)
    out = {"what": "the 8 shipped measurements (drop8 1, runner8 1, traffic 6) through harness.evaluate, FFDNet (net_gray.pth), Anderson, "
                   f"and_maxiters={args.iters}; wall time of upload + reconstruction + PSNR"}
    for name, batch in (("one_by_one", False), ("clip_batched", True), ("all_clips_one_batch", "all")):
        for _ in range(2):                                     # eager warm-up of every shape, then its hipGraph capture
            evaluate(deq, clips, batch=batch)
        dts = []
        for _ in range(2):                                     # two timed passes, the faster one (a pass is 0.45-0.6 s: one allocator hiccup is 15 % of it)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            avg, _ = evaluate(deq, clips, batch=batch)
            torch.cuda.synchronize()
            dts.append(time.perf_counter() - t0)
        out[name] = {"value": 64 / min(dts), "unit": "frames/s", "avg_psnr_db": avg, "passes_s": [round(v, 4) for v in dts]}
    if args.iters == 180:
        g = os.path.join(ROOT, "tests", "golden")
        with open(os.path.join(g, "e2e_ffdnet_anderson_180.json")) as fh:
            out["reference_avg_psnr_db"] = json.load(fh)["avg_psnr"]
        bands = []
        for f in ("e2e_ffdnet_anderson_180_spread.json", "e2e_ffdnet_anderson_180_spread_gram64.json"):
            with open(os.path.join(g, f)) as fh:
                d = json.load(fh)
            bands += [d["avg_psnr_min"], d["avg_psnr_max"]]
        out["reference_avg_psnr_band_db"] = [min(bands), max(bands)]     # 25 runs each: as it is and with an exact Gram, x0 (1 + 1e-7 randn)
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-per-gpu", type=int, default=8, help="weak scaling (default): measurements per GPU")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: total number of measurements, sharded over the GPUs (BASELINE config 3: 64)")
    ap.add_argument("--conv64", default="auto", choices=["auto", "fast", "fast32", "f22", "f44", "s16"],
                    help="kernel of the 64->64 layers (DEQSCIEngine): auto = fast = split-fp16 direct convolution on the f16 matrix cores "
                         "where it is faster, Winograd F(2x2,3x3) below; fast32 = fp32 MFMA arithmetic only (F(4x4,3x3) / F(2x2,3x3))")
    ap.add_argument("--conv64-f22-calls", type=int, default=None, help="run the first K f-calls on F(2x2,3x3) whatever the policy")
    ap.add_argument("--no-other-kernel", action="store_true", help="skip the extra steps under the other conv64 policies")
    ap.add_argument("--other-steps", type=int, default=5, help="timed steps per other conv64 policy (behind one warm-up step)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short runs of the other BASELINE configurations and variants (reference f-call count, the other stack kernel, the "
                         "reference's Anderson arithmetic, SimpleCNN, 512x512x16)")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the oracle spot check of the timed configuration (2 measurements, 10 iterations)")
    ap.add_argument("--iters", type=int, default=180)
    ap.add_argument("--denoiser", default="ffdnet", choices=["ffdnet", "SimpleCNN"])
    ap.add_argument("--size", default="256x256x8")
    ap.add_argument("--cpu-iters", type=int, default=28)
    ap.add_argument("--no-channels-last", action="store_true")
    ap.add_argument("--no-fused-epilogue", action="store_true")
    ap.add_argument("--no-fused-edges", action="store_true")
    ap.add_argument("--no-winograd", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the reconstruction as a hipGraph whatever the batch size (default: up to 4 measurements per call)")
    ap.add_argument("--act-range", default="data", choices=["data", "fixed"], help="scales of the split-fp16 activations (A/B; fixed = 2^8, round 3)")
    ap.add_argument("--stack-per-launch", type=int, default=None, help="images per stack launch (A/B; default: the engine's choice)")
    ap.add_argument("--no-slice-edges", action="store_true", help="FFDNet's first and last layer over the whole batch instead of slice by slice around the stack launches (A/B)")
    ap.add_argument("--stack-kernel", default="w16", choices=["w16", "s16"],
                    help="kernel of FFDNet's stack launches: w16 = split-fp16 under Winograd F(2,3) x direct (csrc/conv_w16.hip), s16 = split-fp16 direct (A/B)")
    ap.add_argument("--anderson-arith", default="reference", choices=["reference", "float64"],
                    help="arithmetic of Anderson's alpha: reference = fp32 Gram + fp32 LU as solvers/new_equilibrium_utils_yaping.py:177-180 (the default of "
                         "every entry point); float64 = exactly accumulated Gram")
    ap.add_argument("--groups", default="1", choices=["1", "2"],
                    help="2: a batch of at least two stack slices runs as two half batches on two streams, issued alternately up to every stack launch "
                         "(DEQSCIEngine groups; bit-identical results; measured -4 % ... +1.3 %: an A/B, not the default); 1: one stream")
    ap.add_argument("--no-stack", action="store_true", help="one launch per 64->64 layer even where a run of layers fits one launch (A/B at small batches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-hbm-stream", action="store_true")
    ap.add_argument("--ranks-share-gpu0", action="store_true",
                    help="test rig for a one-GPU box: all ranks on cuda:0, gloo collective (real engine, real sharding; not a measurement)")
    ap.add_argument("--plumbing-selftest", action="store_true",
                    help="CPU + gloo + a stub engine: exercises launcher/sharding/all-gather only (no GPU, no reconstruction)")
    return ap.parse_args(argv)


def build_engine(args, dev, conv64=None, f22_calls="args", denoiser=None, **over):
    from deqsci_amd import checkpoint
    from deqsci_amd.cli import build_denoiser
    from deqsci_amd.engine import DEQSCIEngine
    # MIOpen find mode (cudnn.benchmark) is deliberately left off: on this conv it picks a slower igemm tile
    # (642 us vs 579 us) and writes that choice into the user find-db (measured, tools/gpu_12.sh).
    denoiser = denoiser or args.denoiser
    net = build_denoiser(denoiser).eval()
    net.load_state_dict({k.replace("nonlinear_op.", ""): v for k, v in
                         checkpoint.read_state_dict(checkpoint.shipped("ffdnet_gray" if denoiser == "ffdnet" else "cnn"))[0].items()})
    net = net.to(dev)
    kw = {}
    if args.no_graph:
        kw["use_graph"] = False
    if args.graph:
        kw["use_graph"] = True
    if args.act_range != "data":
        kw["act_range"] = args.act_range
    if args.no_stack:
        kw["stack"] = False
    if args.stack_kernel != "w16":
        kw["stack_kernel"] = args.stack_kernel
    kw["anderson_arith"] = args.anderson_arith
    kw["groups"] = int(args.groups)
    kw.update(over)
    eng = DEQSCIEngine(net, iterator="anderson", m=5, beta=1.0, lam=1e-2, max_iter=args.iters, tol=1e-5,
                       channels_last=False if args.no_channels_last else None, fused_epilogue=not args.no_fused_epilogue,
                       fused_edges=not args.no_fused_edges, winograd=not args.no_winograd,
                       conv64=args.conv64 if conv64 is None else conv64, conv64_f22_calls=args.conv64_f22_calls if f22_calls == "args" else f22_calls, **kw)
    if args.stack_per_launch is not None:
        eng.den.stack_per_launch = args.stack_per_launch
    if args.no_slice_edges:
        eng.den.slice_edges = False
    return eng


def make_step(eng, y_local, Phi_local, M, gather_timer):
    """One bench step = the product's multi-GPU entry point: this rank's contiguous slice of the M measurements through the
    engine, then the path's one all-gather (identity at world size 1)."""
    def step():
        return distributed.reconstruct_shard(eng.reconstruct, y_local, Phi_local, M, timer=gather_timer)
    return step


def run_rank(args):
    selftest = args.plumbing_selftest
    if not selftest and not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback for the product path)")
    rank, world, local_rank, dev = distributed.init_from_env("gloo" if selftest else ("gloo+cuda0" if args.ranks_share_gpu0 else "nccl"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    H, W, B = (int(v) for v in args.size.split("x"))
    strong = args.global_batch > 0
    M = args.global_batch if strong else world * args.batch_per_gpu          # measurements of the whole job
    lo, hi, per = distributed.shard_bounds(M, world, rank)
    bsz = hi - lo                                                            # this rank's
    if selftest:
        eng = _PlumbingStub(args.iters)
        _hip = None
    else:
        from deqsci_amd import _hip
        eng = build_engine(args, dev)
    # this rank's slice [lo, hi) of the global batch, generated here by global measurement index
    y, Phi, _ = make_batch(lo, hi, H, W, B, 1234, dev)
    gather_timer = distributed.GatherTimer()
    step = make_step(eng, y, Phi, M, gather_timer)

    # per-launch timing of the fused Phi/Phi^T+GAP-update kernel and of the Winograd conv from the dispatch's own
    # HIP-event timestamps (hipExtLaunchKernelGGL start/stop events on the stream the kernel runs on); all events are
    # created before the timed region and destroyed after it
    # the hipGraph path (small batches) replays captured launches: no per-launch events there, and the two calls that set it up
    # (eager warm-up of the shape, then the capture) are made here, before the W warm-up steps, so that every timed step is a replay
    graph_mode = (not selftest) and (eng.use_graph is True or (eng.use_graph == "auto" and bsz * H * W <= eng.GRAPH_AUTO_PIXELS))
    if graph_mode:
        for _ in range(2):
            step()
    timing = not (selftest or args.no_kernel_timing or graph_mode) and rank == 0
    timer = None
    conv_timers, conv_shape, conv_launches = {}, {}, {"f22": 0, "f44": 0, "s16": 0, "s16stack": 0, "w16": 0, "w16stack": 0}
    if timing:
        timer = _hip.KernelTimer(capacity=args.steps * max(args.iters, 1))
        conv_timers = {k: _hip.KernelTimer(capacity=400) for k in conv_launches}   # a sample of launches of each kernel is enough
        timing_on = [False]
        orig = _hip.anderson_mix_gap

        mix_gap_bsz = [bsz]

        def timed_mix_gap(ws, beta, n, *a):
            if not timing_on[0] or n != eng.m:
                return orig(ws, beta, n, *a)
            mix_gap_bsz[0] = ws.bsz                            # (a grouped reconstruction launches it per half batch)
            return timer.mix_gap(ws, beta, n, *a)
        _hip.anderson_mix_gap = timed_mix_gap

        def conv_hook(kind, n, H, W, layers=1):                # _hip.CONV64_EVENT_HOOK: a (start, stop) event pair per 64->64 launch
            if not timing_on[0]:                               # ("s16stack": ONE launch = `layers` layers over a slice of n images)
                return None
            conv_launches[kind] += 1
            conv_shape[kind] = [n, H, W, layers]
            return conv_timers[kind].pair()
        _hip.CONV64_EVENT_HOOK = conv_hook

    def fence():
        if world > 1:
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    gather_timer.total_seconds()
    gather_timer.seconds, gather_timer.calls = 0.0, 0
    if timing:
        timing_on[0] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    local_elapsed = elapsed
    if timing:
        timing_on[0] = False
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # N > 1: what proves, on the first hardware run, that the collective library saw N ranks on N different devices - gathered once,
    # outside the timed region
    ranks_info = None
    if world > 1:
        mine = {"rank": rank, "local_rank": local_rank, "device": str(dev), "ms_per_step": 1e3 * local_elapsed / max(args.steps, 1),
                "measurements": [lo, hi]}
        if dev.type == "cuda":
            props = torch.cuda.get_device_properties(dev)
            mine.update({"name": props.name, "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None),
                         "pci_device_id": getattr(props, "pci_device_id", None), "compute_units": props.multi_processor_count})
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        ranks_info = gathered
    frames = M * B * args.steps
    value = frames / elapsed
    info = eng.last_info or {}                                               # (an idle rank - more GPUs than measurements - never ran)
    f_calls = info.get("f_calls")
    out = {
        "metric": "reconstructed frames/sec at 256x256x8, 180 DEQ iters",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "f32 (tensors and accumulation; the 64->64 layers' products as split-fp16 MFMAs, see arithmetic)", "data": "synthetic",
        "config": {"workload": (f"synthetic batch of {M} measurements sharded over {world} GPU(s) ({per} per GPU; BASELINE config 3 as stated), "
                                if strong else f"synthetic batch of {per} measurements per GPU, ") + f"{H}x{W}x{B}, Bernoulli(0.5) masks "
                               f"(BASELINE config 3 per-GPU shard = config 2's 8 measurements at N=1); "
                               f"{args.denoiser} denoiser ({'net_gray.pth weights, substitute for the missing ffdnet.ckpt' if args.denoiser == 'ffdnet' else 'cnn.ckpt'}), "
                               f"Anderson m=5 lam=1e-2 beta=1, and_maxiters={args.iters}, tol=1e-5",
                   "global_batch": M, "batch_per_gpu": per, "frames_per_measurement": B, "f_calls_per_step": f_calls,
                   "f_calls_reference": args.iters + 2,            # new_equilibrium_utils_yaping.py:259-272: max_iter + 2, the last one dead (:271-272)
                   "conv64_policy": None if selftest else (f"{eng.conv64} -> F(2x2,3x3) for f-calls < {eng.conv64_f22_calls}, then {eng.conv64_policy}"
                                                           if eng.conv64_f22_calls else f"{eng.conv64} -> {eng.conv64_policy}"),
                   "parallelism": f"measurements sharded over {world} GPU(s), one all-gather per step" if world > 1 else "single GPU",
                   "launch_mode": "hipGraph replay of the whole reconstruction (captured before the warm-up steps)" if graph_mode else "eager launches",
                   # f-calls whose run of 64->64 layers went out as ONE launch (csrc/conv_s16.hip, STACK: at most one block tile per CU, i.e. one
                   # measurement of 256x256x8 per call); 0 = a launch per layer
                   "stack_launches_per_step": info.get("stack_launches", 0),
                   "stack_kernel": None if selftest else getattr(eng.den, "stack_kernel", None),
                   # reconstructions (warm-up included) whose stack launch timed out and were redone with a launch per layer: foreign work on the
                   # device's CUs (two ranks on one GPU provoke it); 0 on a device of one's own
                   "stack_timeouts": None if selftest else getattr(eng, "stack_timeouts_total", None),
                   "anderson_arith": None if selftest else getattr(eng, "anderson_arith", None),
                   # [lo, hi) measurements of the half batches a grouped step runs on two streams (None: one stream)
                   "groups": info.get("groups")},
        "arithmetic": ("fp32 tensors, fp32 accumulation everywhere.  64->64 conv layers under conv64 policy 'fast' (default): products on the f16 matrix "
                       "cores from hi + lo fp16 pairs of the fp32 operands (22 significant bits each, three MFMAs per product, fp32 accumulation; the "
                       "power-of-two scale of every activation follows max |activation| measured on the device at the first f-call, so the path is "
                       "scale-free like fp32).  stack_kernel 'w16' (default): FFDNet's 13 layers under Winograd F(2,3) along x nested in the direct sum "
                       "along y (U = G g in float64, B^T d in fp32 then hi + lo, one accumulation chain of 36 MFMAs per position, fp32 activations "
                       "between the layers), 1.2-1.5e-7 per layer against float64; 's16' / the measuring f-call: the direct convolution, 1.6e-7 "
                       "(fp32 Winograd F(2x2,3x3) 2.0e-7, MIOpen's fp32 direct convolution 3.5e-7: profiles/r03_conv_error_real.json).  "
                       "other_conv64_policies gives the same step on fp32-MFMA kernels only."),
        "final_res": info.get("res"),
        "allgather_ms_per_step": 1e3 * gather_timer.total_seconds() / max(args.steps, 1) if world > 1 else 0.0,
        "allgather_bytes_per_step": world * per * H * W * B * 4 if world > 1 else 0,      # what every rank receives: (R*per,H,W,B) fp32
    }
    if world > 1:
        out["distributed"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": ranks_info,
                              "distinct_devices": len({(r.get("uuid") or r.get("pci_bus_id") or r["device"]) for r in ranks_info})}
    if args.ranks_share_gpu0:
        out["data"] = "synthetic; TEST RIG: all ranks share cuda:0 over gloo - not a throughput measurement"
    if selftest:
        out.update({"metric": "PLUMBING SELFTEST - launcher/sharding/all-gather only, no reconstruction", "data": "selftest",
                    "dtype": "none", "value": 0.0})
    if rank == 0:
        ms = timer.durations_ms() if timing else []
        if ms:
            avg_s = 1e-3 * sum(ms) / len(ms)
            nbytes = mix_gap_bytes(mix_gap_bsz[0], H, W, B, eng.m)
            traffic = None
            for tname in ("r04_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json", "r01_pmc_hbm_traffic.json"):   # rocprofv3 --pmc passes, tools/pmc_traffic.sh
                tfile = os.path.join(ROOT, "profiles", tname)
                if traffic is None and os.path.exists(tfile):
                    with open(tfile) as fh:
                        for rec in json.load(fh):
                            k = rec["kernels"].get(f"mix_gap_bhw_kernel<{B}>")
                            if rec["bsz"] == mix_gap_bsz[0] and rec["size"] == args.size and k:
                                traffic = k["hbm_bytes_per_launch"]
            out["hbm_roofline"] = {"kernel": f"deqsci::mix_gap_bhw_kernel<{B}> (K7+K3: Anderson mix fused with the Phi/Phi^T GAP update), in the DEQ loop",
                                   "bound": "hbm", "achieved": nbytes / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": nbytes / avg_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                                   "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": 1e6 * avg_s, "launches_timed": len(ms),
                                   "note": f"in-loop figure at bsz {mix_gap_bsz[0]} per launch: the kernel's {nbytes / 2**20:.0f} MiB working set sits partly in the "
                                           "256 MiB Infinity Cache" + (", and in a grouped step the other half batch's kernels run beside it" if mix_gap_bsz[0] != bsz else "")
                                           + "; hbm_stream_roofline is the same kernel alone on a working set far beyond the cache"}
        # "roofline" = the DOMINANT kernel of the step: the denoiser's 64->64 conv layers (13 launches per f-call), on the kernel that takes
        # the largest share of the step; any other 64->64 kernel that ran is reported under "roofline_other_kernels".
        # Algorithmic flops per launch = the MFMA flops the kernel's algorithm executes: split-fp16 direct
        # convolution 3 x the direct flops (three f16 products per multiplication) against the dense f16 MFMA peak; Winograd F(4x4,3x3)
        # direct / 4, F(2x2,3x3) direct / 2.25 against the dense fp32 MFMA peak.  "hbm_roofline" is the fused streaming kernel of the DEQ loop.
        forms = {}
        for kind, ct in (conv_timers.items() if timing else ()):
            cms = ct.durations_ms()
            if not cms:
                continue
            nimg, ch, cw, nlay = conv_shape[kind]
            direct = 2.0 * 64 * 64 * 9 * ch * cw * nimg * nlay      # (a stack launch: all its layers)
            # (split-fp16 under Winograd F(2,3) x direct: three f16 products per multiplication, 6 multiplications per output instead of 9)
            mult, peak = {"s16": (3.0, MFMA_F16_PEAK_TFLOPS), "s16stack": (3.0, MFMA_F16_PEAK_TFLOPS), "w16": (2.0, MFMA_F16_PEAK_TFLOPS),
                          "w16stack": (2.0, MFMA_F16_PEAK_TFLOPS), "f44": (1 / 4.0, MFMA_F32_PEAK_TFLOPS), "f22": (1 / 2.25, MFMA_F32_PEAK_TFLOPS)}[kind]
            cavg = 1e-3 * sum(cms) / len(cms)
            share = cavg * conv_launches[kind] / elapsed
            wtraffic, wsource = None, None                        # fabric-side bytes per launch from the PMC passes of tools/pmc_winograd.sh (FETCH_SIZE +
                                                                  # WRITE_SIZE count at the L2's memory side, IN FRONT of the Infinity Cache: not HBM bytes)
            for wname in {"s16": ("r04_pmc_conv_s16.json", "r03_pmc_conv_s16.json"), "s16stack": ("r04_pmc_conv_s16_stack.json",),
                          "w16": ("r05_pmc_conv_w16.json",), "w16stack": ("r05_pmc_conv_w16_stack.json",),
                          "f44": ("r03_pmc_winograd44.json", "r02_pmc_winograd44.json"),
                          "f22": ("r03_pmc_winograd.json", "r02_pmc_winograd.json", "r01_pmc_winograd.json")}[kind]:
                wfile = os.path.join(ROOT, "profiles", wname)
                if wtraffic is None and os.path.exists(wfile):
                    with open(wfile) as fh:
                        rec = json.load(fh)
                    if rec.get("shape") == [nimg, 64, ch, cw] and rec.get("layers", 1) == nlay:
                        wtraffic, wsource = rec["hbm_bytes_per_launch"], "profiles/" + wname
            kname = {"s16": "deqsci::s16::conv_s16_kernel<0, 0, 0> (conv3x3 64->64 + bias + ReLU, direct convolution on the f16 matrix cores: fp32 operands as hi + lo "
                            "fp16 pairs, three MFMAs per product, fp32 accumulation)",
                     "s16stack": f"deqsci::s16::conv_s16_kernel<0, 0, 1> (the same arithmetic; ONE launch = the denoiser's {nlay} consecutive conv3x3 64->64 + bias + ReLU "
                                 f"layers over a slice of {nimg} images, tiles synchronised by per-tile progress words, the slice's activations resident in the "
                                 "Infinity Cache; flops and time are the whole launch's)",
                     "w16": "deqsci::w16::conv_w16_kernel<0> (conv3x3 64->64 + bias + ReLU on the f16 matrix cores: fp32 operands as hi + lo fp16 pairs, three "
                            "MFMAs per product, under Winograd F(2,3) along x nested in the direct sum along y - 6 multiplications per output instead of 9)",
                     "w16stack": f"deqsci::w16::conv_w16_kernel<1> (split-fp16 arithmetic under Winograd F(2,3) along x nested in the direct sum along y; ONE launch = the "
                                 f"denoiser's {nlay} consecutive conv3x3 64->64 + bias + ReLU layers over a slice of {nimg} images, tiles synchronised by per-tile "
                                 "progress words, the slice's activations resident in the Infinity Cache; flops and time are the whole launch's)",
                     "f44": "deqsci::w44::winograd44_conv64_kernel (conv3x3 64->64 + bias + ReLU, Winograd F(4x4,3x3) on fp32 MFMA)",
                     "f22": "deqsci::winograd_conv64_kernel (conv3x3 64->64 + bias + ReLU, Winograd F(2x2,3x3) on fp32 MFMA)"}[kind]
            forms[kind] = {"kernel": kname,
                           "bound": "mfma", "achieved": direct * mult / cavg / 1e12, "peak": peak, "unit": "TFLOP/s",
                           "frac": direct * mult / cavg / 1e12 / peak, "traffic": wtraffic,
                           # where `traffic` comes from: a rocprofv3 --pmc pass of the same launch shape on another box (FETCH_SIZE / WRITE_SIZE:
                           # bytes at the L2's fabric side, in front of the Infinity Cache - NOT HBM bytes for a cache-resident slice); null: no pass
                           "traffic_source": wsource, "traffic_counts": "fabric-side bytes (L2 <-> Infinity Cache / HBM), not HBM bytes",
                           # executed = the MFMA flops the kernel's algorithm issues (what `achieved` / `frac` price: matrix-pipe utilisation);
                           # algorithmic = the direct-convolution flops of the layer, SURVEY 8(d) / section 6 (`frac_useful`)
                           "executed_mfma_flops_per_launch": direct * mult, "algorithmic_flops_per_launch": direct,
                           "achieved_useful": direct / cavg / 1e12, "frac_useful": direct / cavg / 1e12 / peak,
                           "avg_launch_us": 1e6 * cavg, "launches_timed": len(cms), "launches_per_step": conv_launches[kind] // max(args.steps, 1),
                           "share_of_step_time": round(share, 3),
                           "layers_per_launch": nlay, "images_per_launch": nimg,
                           "note": {"s16": "v_mfma_f32_32x32x16_f16, peak = dense f16 MFMA; on random operands the kernel runs against the chip's power limit "
                                           "(1.9-2.0 GHz, not 2.4): see DESIGN.md section 6",
                                    "s16stack": "v_mfma_f32_32x32x16_f16, peak = dense f16 MFMA; the launch runs against the chip's power limit (1.6 GHz, not 2.4): "
                                                "see DESIGN.md section 6.4",
                                    "w16": "v_mfma_f32_32x32x16_f16, peak = dense f16 MFMA; executed = 2 x the direct-convolution flops (3 products x 6 / 9)",
                                    "w16stack": "v_mfma_f32_32x32x16_f16, peak = dense f16 MFMA; executed = 2 x the direct-convolution flops (three f16 products per "
                                                "multiplication, 6 multiplications per output instead of 9): see DESIGN.md section 6.5",
                                    "f44": "F(4x4,3x3) executes 0.5625x the MFMA flops of F(2x2,3x3) for the same layer: frac prices executed MFMA work",
                                    "f22": "F(2x2,3x3): the launcher's choice below one wave of 16 x 32 block tiles"}[kind]}
        if forms:
            order = sorted(forms, key=lambda k: -forms[k]["share_of_step_time"])
            out["roofline"] = forms[order[0]]
            out["config"]["conv64_kernel"] = " + ".join(f"{k} ({forms[k]['launches_per_step']} launches per step)" for k in order)
            if len(order) > 1:
                out["roofline_other_kernels"] = {k: forms[k] for k in order[1:]}
        if "roofline" not in out and "hbm_roofline" in out:      # a run without the Winograd kernel (--no-winograd)
            out["roofline"] = out["hbm_roofline"]
        if timing:
            timer.close()
            _hip.CONV64_EVENT_HOOK = None
            for ct in conv_timers.values():
                ct.close()
        if world == 1 and not selftest:
            if not args.no_other_kernel and not args.no_winograd:
                # the same step under the OTHER conv64 policies, `--other-steps` timed steps each behind a warm-up step, outside the timed region:
                # what the all-fp32-MFMA paths deliver on this very box, sustained (the reader who does not accept split-fp16 operands as fp32
                # arithmetic takes "fast32")
                out["other_conv64_policies"] = {}
                for name, pol, k in (("fast32", "fast32", None), ("fast32, first 40 f-calls on F(2x2,3x3)", "fast32", 40), ("f22", "f22", None)):
                    if pol == eng.conv64_policy and k == eng.conv64_f22_calls:
                        continue
                    eng2 = build_engine(args, dev, conv64=pol, f22_calls=k)
                    step2 = make_step(eng2, y, Phi, M, distributed.GatherTimer())
                    step2()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(args.other_steps):
                        step2()
                    torch.cuda.synchronize()
                    dt2 = (time.perf_counter() - t1) / args.other_steps
                    out["other_conv64_policies"][name] = {"value": M * B / dt2, "unit": "frames/s", "ms_per_step": 1e3 * dt2, "steps": args.other_steps,
                                                          "warmup": 1}
                    del eng2, step2
            if not args.no_other_configs and args.denoiser == "ffdnet" and args.size == "256x256x8":
                # short runs OUTSIDE the timed region, same box, each behind a warm-up step: (a) the reference's f-call count measured, not
                # scaled (extra_call = the dead f(z) of new_equilibrium_utils_yaping.py:271-272: 182 calls); (b) the other stack kernel; (c) the
                # reference's Anderson arithmetic (fp32 Gram: anderson_arith="reference"); (d) BASELINE config 5 (SimpleCNN) and (e) config 4
                # (512x512x16, one step): driver-run numbers for the configurations the headline is not quoted on
                def short(eng_, y_, Phi_, M_, steps):
                    st_ = make_step(eng_, y_, Phi_, M_, distributed.GatherTimer())
                    st_()
                    torch.cuda.synchronize()
                    t1_ = time.perf_counter()
                    for _ in range(steps):
                        st_()
                    torch.cuda.synchronize()
                    dt_ = (time.perf_counter() - t1_) / steps
                    return {"value": M_ * Phi_.shape[-1] / dt_, "unit": "frames/s", "ms_per_step": 1e3 * dt_, "steps": steps, "warmup": 1,
                            "f_calls_per_step": (eng_.last_info or {}).get("f_calls")}
                oc = {}
                e_ = build_engine(args, dev, extra_call=True)
                oc["value_at_reference_f_calls"] = dict(short(e_, y, Phi, M, 3), what="the same step with the reference's dead extra f-call (extra_call=True): max_iter + 2 calls")
                del e_
                other_sk = "s16" if eng.den.stack_kernel == "w16" else "w16"
                e_ = build_engine(args, dev, stack_kernel=other_sk)
                oc["other_stack_kernel"] = dict(short(e_, y, Phi, M, 3), stack_kernel=other_sk)
                del e_
                other_aa = "reference" if eng.anderson_arith == "float64" else "float64"
                e_ = build_engine(args, dev, anderson_arith=other_aa)
                oc["other_anderson_arith"] = dict(short(e_, y, Phi, M, 3), anderson_arith=other_aa)
                del e_
                e_ = build_engine(args, dev, denoiser="SimpleCNN")
                oc["config5_simplecnn"] = dict(short(e_, y, Phi, M, 2), what="BASELINE configs[4]: DE-GAP-CNN denoiser (models/cnn.ckpt), same batch, 180 iterations")
                del e_
                y4, Phi4, _ = make_batch(0, 2, 512, 512, 16, args.seed if hasattr(args, "seed") else 1234, dev)
                e_ = build_engine(args, dev)
                oc["config4_512x512x16"] = dict(short(e_, y4, Phi4, 2, 1), what="BASELINE configs[3]: 2 measurements of 512x512x16, FFDNet, 180 iterations")
                del e_, y4, Phi4
                e_ = build_engine(args, dev, groups=1 if info.get("groups") else 2)
                oc["two_streams" if not info.get("groups") else "one_stream"] = dict(
                    short(e_, y, Phi, M, 3), what="the same step with DEQSCIEngine(groups=2): two half batches on two streams (bit-identical; not the default)"
                    if not info.get("groups") else "the same step on one stream (groups=1)")
                del e_
                # the reference's schedule (training/sci_equilibrium_training.py:171-181): ONE measurement per call - a hipGraph replay of the
                # whole reconstruction (the eager warm-up and the capture are the two calls in front of the warm-up step)
                e_ = build_engine(args, dev)
                st1 = make_step(e_, y[:1], Phi[:1], 1, distributed.GatherTimer())
                st1()
                st1()
                oc["one_measurement_per_call"] = dict(short(e_, y[:1], Phi[:1], 1, 3), launch_mode="hipGraph replay" if (e_.last_info or {}).get("graph") else "eager")
                del e_, st1
                oc["config2_shipped_clips"] = config2_shipped_clips(args, dev)
                out["other_configs"] = oc
            if not args.no_hbm_stream:
                del y, Phi
                out["hbm_stream_roofline"] = hbm_stream_roofline(H, W, B, eng.m, dev)
            if not args.no_parity_check:
                out["parity_spot_check"] = parity_spot_check(args, dev, H, W, B)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(args.cpu_iters, f_calls, H, W, B, args.denoiser)
            # the numbers a reader asks for next to the headline, as scalars: in `config` (which record keepers parse whole) and once more as
            # the LAST key of the line (which survives a tail cut).  frames/s unless named otherwise; all on this box, outside the timed region
            oc, op = out.get("other_configs", {}), out.get("other_conv64_policies", {})
            c2 = oc.get("config2_shipped_clips", {})
            summary = {
                "headline_frames_per_s": round(value, 2),
                "two_streams_groups_2": round(oc["two_streams"]["value"], 2) if "two_streams" in oc else None,
                "one_stream_groups_1": round(oc["one_stream"]["value"], 2) if "one_stream" in oc else None,
                "exact_gram_float64" if eng.anderson_arith == "reference" else "reference_gram": round(oc["other_anderson_arith"]["value"], 2) if "other_anderson_arith" in oc else None,
                "with_the_references_dead_182nd_f_call": round(oc["value_at_reference_f_calls"]["value"], 2) if "value_at_reference_f_calls" in oc else None,
                "fp32_mfma_only_policy_fast32": round(op["fast32"]["value"], 2) if "fast32" in op else None,
                "one_measurement_per_call": round(oc["one_measurement_per_call"]["value"], 2) if "one_measurement_per_call" in oc else None,
                "config2_shipped_clips_one_by_one": round(c2["one_by_one"]["value"], 2) if c2 else None,
                "config2_shipped_clips_clip_batched": round(c2["clip_batched"]["value"], 2) if c2 else None,
                "config2_shipped_clips_all_in_one_batch": round(c2["all_clips_one_batch"]["value"], 2) if c2 else None,
                "config2_avg_psnr_db": round(c2["one_by_one"]["avg_psnr_db"], 4) if c2 else None,
                "config2_reference_avg_psnr_db": round(c2["reference_avg_psnr_db"], 4) if c2.get("reference_avg_psnr_db") else None,
                "config2_reference_avg_psnr_band_db": [round(v, 4) for v in c2["reference_avg_psnr_band_db"]] if c2.get("reference_avg_psnr_band_db") else None,
                "config5_simplecnn": round(oc["config5_simplecnn"]["value"], 2) if "config5_simplecnn" in oc else None,
                "config4_512x512x16": round(oc["config4_512x512x16"]["value"], 2) if "config4_512x512x16" in oc else None,
                "stack_launch_us": round(out["roofline"]["avg_launch_us"], 1) if "avg_launch_us" in out.get("roofline", {}) else None,
                "roofline_frac_useful": round(out["roofline"]["frac_useful"], 4) if "frac_useful" in out.get("roofline", {}) else None,
                "hbm_stream_frac_K3": round(next(iter(out["hbm_stream_roofline"]["kernels"].values()))["frac"], 3) if "hbm_stream_roofline" in out else None,
                "parity_spot_check_rel_l2": out.get("parity_spot_check", {}).get("rel_l2"),
            }
            summary = {k: v for k, v in summary.items() if v is not None}
            out["config"]["summary"] = summary
            out["summary"] = summary
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    args = parse_args(argv)
    if distributed.relaunch_needed(args.gpus):
        # started as plain `python bench.py --gpus N`: become the launcher.  Nothing above has touched the GPU and the ranks are
        # NEW processes, one per GPU.
        # (the parent never calls into HIP, not even to count devices: visibility variables / KFD topology only)
        seen = distributed.visible_gpu_count()
        if not (args.plumbing_selftest or args.ranks_share_gpu0) and seen is not None and seen < args.gpus:
            sys.exit(f"--gpus {args.gpus} but only {seen} GPU(s) are visible")
        sys.exit(distributed.launch_ranks([sys.executable, os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv),
                                          args.gpus))
    run_rank(args)


if __name__ == "__main__":
    main()
