"""Command-line driver with the reference's flags (video_sci_proxgrad.py:23-49), inference only.

    python -m deqsci_amd.cli --denoiser ffdnet --loadpath deqsci_amd/weights/ffdnet_gray.npz \\
        --testpath data/test_gray/ --and_maxiters 180 --inference True [--gpu_ids 0,1,2,3]

Every flag of the reference parses (so its test_*.sh command lines run unchanged); the training-only ones
(--n_epochs --batch_size --lr --lr_gamma --sched_step --trainpath --print_every_n_steps --save_every_n_steps --etainit
--sigma) are accepted and ignored, and `--inference False` is refused: training is outside this build.
`--gpu_ids a,b,..` with more than one id is this build's addition: one process per listed GPU, every clip's
measurements sharded over them (deqsci_amd.distributed), rank 0 prints and writes the PNGs.
"""
import argparse
import os
import sys
import time

import torch

from . import checkpoint, distributed
from .harness import SCITestDataset, evaluate, png_payloads, write_png
from .networks import DnCNN, FFDNet
from .operators import A_torch_, At_torch_
from .solvers import DEQFixedPoint, EquilibriumProxGradSCI, andersonexp

SHIPPED = {'ffdnet': 'ffdnet_gray', 'SimpleCNN': 'cnn', 'RealSN_SimpleCNN': 'rsn_cnn'}


def build_denoiser(name, n_channels=1):
    """Factory of video_sci_proxgrad.py:145-185 restricted to the denoisers with shipped SCI weights."""
    if name == 'ffdnet':
        return FFDNet(num_input_channels=n_channels, tag='ffdnet')
    if name == 'SimpleCNN':
        return DnCNN(1, num_of_layers=4, lip=0.0, no_bn=True, tag='denoiser')
    if name == 'RealSN_SimpleCNN':
        return DnCNN(1, num_of_layers=4, lip=1.0, no_bn=True, tag='denoiser')
    raise NotImplementedError('unknown denoiser!')


def build_pipeline(denoiser, loadpath=None, and_maxiters=100, and_m=5, and_beta=1.0, device="cuda"):
    net = build_denoiser(denoiser).eval()
    solver = EquilibriumProxGradSCI(A=A_torch_, At=At_torch_, nonlinear_operator=net, eta=0.2, minval=-1, maxval=1)
    if loadpath:
        checkpoint.load_solver(solver, loadpath)
    solver = solver.to(device)
    deq = DEQFixedPoint(solver, andersonexp, m=and_m, beta=and_beta, lam=1e-2, max_iter=and_maxiters, tol=1e-5)
    return solver, deq


def parser():
    p = argparse.ArgumentParser(description="DEQ-SCI inference on MI355X")
    p.add_argument('--gpu_ids', default='0')
    p.add_argument('--and_maxiters', default=100, type=int)
    p.add_argument('--and_beta', type=float, default=1.0)
    p.add_argument('--and_m', type=int, default=5)
    p.add_argument('--denoiser', default='ffdnet')
    p.add_argument('--savepath', default="./save/test/")
    p.add_argument('--loadpath', default='')
    p.add_argument('--testpath', default="./data/test_gray/")
    p.add_argument('--inference', default='True')
    p.add_argument('--conv64', default='auto', choices=['auto', 'fast', 'fast32', 'f22', 'f44', 's16'],
                   help="(this build) kernel of the denoiser's 64->64 layers: auto = split-fp16 direct convolution on the f16 matrix cores / "
                        "Winograd F(2x2,3x3), the faster per launch; fast32 = fp32 MFMA arithmetic only")
    p.add_argument('--anderson_arith', default='reference', choices=['reference', 'float64'],
                   help="(this build) how alpha is computed: reference (default) = the reference's own arithmetic - the fp32 Gram of "
                        "new_equilibrium_utils_yaping.py:177-178 in the summation order of its torch.bmm, fp32 LU - which reproduces the reference's "
                        "ensemble statistics on the chaotic FFDNet + Anderson @180 configuration (DESIGN.md section 5); float64 = Gram and solve in "
                        "float64 (exact; 4 % faster at eight measurements per call, 14 % at one)")
    p.add_argument('--batch_measurements', nargs='?', const='clip', default=None, choices=['clip', 'all'],
                   help="(this build) clip: a clip's measurements as ONE engine batch instead of the reference's one-by-one schedule (implied by "
                        "more than one --gpu_ids entry, which shards them); all: the measurements of all clips of one frame size as one batch "
                        "(the three shipped clips: one call of eight measurements - what the device is fastest at)")
    ignored = p.add_argument_group("accepted for command-line compatibility, unused by inference")
    ignored.add_argument('--n_epochs', default=80)
    ignored.add_argument('--batch_size', type=int, default=1)
    ignored.add_argument('--lr', type=float, default=0.0001)
    ignored.add_argument('--etainit', type=float, default=0.9)
    ignored.add_argument('--lr_gamma', type=float, default=0.9)
    ignored.add_argument('--sched_step', type=int, default=10)
    ignored.add_argument('--trainpath', default="")
    ignored.add_argument('--print_every_n_steps', type=int, default=1)
    ignored.add_argument('--save_every_n_steps', type=int, default=50)
    ignored.add_argument('--sigma', type=int, default=0)
    return p


def run(args):
    """One rank (or the only process): build, evaluate every clip, rank 0 reports."""
    rank, world, _, dev = distributed.init_from_env("nccl")
    loadpath = args.loadpath or checkpoint.shipped(SHIPPED[args.denoiser])
    _, deq = build_pipeline(args.denoiser, loadpath, args.and_maxiters, args.and_m, args.and_beta, device=dev)
    opts = {}
    if args.conv64 != 'auto':
        opts["conv64"] = args.conv64
    if args.anderson_arith != 'reference':
        opts["anderson_arith"] = args.anderson_arith
    if opts:
        deq.engine_options = opts
    if rank == 0:
        print('loaded dict!')
        os.makedirs(args.savepath, exist_ok=True)
    images = {}

    def on_clip(r):
        if rank == 0:
            images.update(png_payloads(r, args.savepath))
            print([r.name], '  PSNR: %.2f dB' % r.mean_psnr)
    t0 = time.time()
    avg, results = evaluate(deq, SCITestDataset(args.testpath), device=dev, on_clip=on_clip,
                            batch="all" if args.batch_measurements == "all" else bool(args.batch_measurements or world > 1))
    dt = time.time() - t0
    if rank == 0:
        print('---------------------------------', 'Total Average PSNR: %.2f dB' % avg)
        for path, img in images.items():
            write_png(path, img)
        n = sum(r.frames for r in results)
        print(f"{n} frames in {dt:.2f} s -> {n / dt:.2f} frames/s on {world} GPU(s) (excl. PNG export)")
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return avg


def main(argv=None):
    args = parser().parse_args(argv)
    if str(args.inference).lower() in ('false', '0', ''):
        sys.exit("deqsci_amd is the inference hot path only: --inference False (training) is out of scope")
    if args.denoiser not in SHIPPED:
        raise NotImplementedError('unknown denoiser!')
    ids = [int(v) for v in str(args.gpu_ids).split(',') if v != '']
    if distributed.relaunch_needed(len(ids)):
        # the launcher parent never calls into HIP (not even to count devices): visibility variables / KFD topology only
        seen = distributed.visible_gpu_count()
        if seen is not None and seen <= max(ids):
            sys.exit(f"--gpu_ids {args.gpu_ids} but only {seen} GPU(s) are visible")
        cmd = [sys.executable, "-m", "deqsci_amd.cli"] + list(sys.argv[1:] if argv is None else argv)
        sys.exit(distributed.launch_ranks(cmd, len(ids), device_ids=ids))
    if not torch.cuda.is_available():
        sys.exit("deqsci_amd needs an MI355X: there is no CPU path")
    if len(ids) == 1 and "LOCAL_RANK" not in os.environ:
        os.environ["LOCAL_RANK"] = str(ids[0])
    return run(args)


if __name__ == "__main__":
    main()
