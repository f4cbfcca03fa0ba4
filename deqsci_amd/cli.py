"""Command-line driver with the reference's flags (video_sci_proxgrad.py:23-49), inference only.

    python -m deqsci_amd.cli --denoiser ffdnet --loadpath deqsci_amd/weights/ffdnet_gray.npz \\
        --testpath data/test_gray/ --and_maxiters 180 --inference True

Flags keep the reference's names; they are typed here (the reference leaves several as strings).
"""
import argparse
import os
import sys
import time

import torch

from . import checkpoint
from .harness import SCITestDataset, test_solver_sci
from .networks import DnCNN, FFDNet
from .operators import A_torch_, At_torch_
from .solvers import DEQFixedPoint, EquilibriumProxGradSCI, andersonexp


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


def train(args):
    """The training branch of video_sci_proxgrad.py (:129-133,:190-202,:258-268): <trainpath>/{gt/,measurement/,mask.mat},
    Adam + StepLR, MSE(mean), checkpoints under <savepath>/model/, images under <savepath>/img/{train,test,best}/."""
    from .harness import SCITrainingDatasetSubset, train_solver_sci
    save_model_path = args.savepath + 'model/'
    img = {k: args.savepath + f'img/{k}/' for k in ('train', 'test', 'best')}
    for path in (save_model_path, *img.values()):
        os.makedirs(path, exist_ok=True)
    dataset = SCITrainingDatasetSubset(args.trainpath + 'gt/', args.trainpath + 'measurement/', args.trainpath + 'mask.mat')
    loader = torch.utils.data.DataLoader(dataset=dataset, batch_size=args.batch_size, shuffle=True, drop_last=True, pin_memory=True)
    test_loader = torch.utils.data.DataLoader(dataset=SCITestDataset(args.testpath), batch_size=1, shuffle=False, drop_last=True)
    solver, deq = build_pipeline(args.denoiser, args.loadpath or None, args.and_maxiters, args.and_m, args.and_beta)
    solver.nonlinear_op.train()
    optimizer = torch.optim.Adam(params=solver.parameters(), lr=args.lr)
    scheduler = torch.optim.lr_scheduler.StepLR(optimizer=optimizer, step_size=args.sched_step, gamma=args.lr_gamma)
    train_solver_sci(single_iterate_solver=solver, train_dataloader=loader, test_dataloader=test_loader, optimizer=optimizer,
                     save_model_path=save_model_path, deep_eq_module=deq, loss_function=torch.nn.MSELoss(reduction='mean'),
                     n_epochs=args.n_epochs, scheduler=scheduler, print_every_n_steps=args.print_every_n_steps,
                     save_every_n_steps=args.save_every_n_steps, start_epoch=0, train_img_path=img['train'],
                     test_img_path=img['test'], best_img_path=img['best'], tflog_path=args.savepath)


def main(argv=None):
    p = argparse.ArgumentParser(description="DEQ-SCI inference on MI355X")
    p.add_argument('--n_epochs', default=80, type=int)
    p.add_argument('--batch_size', type=int, default=1)
    p.add_argument('--and_maxiters', default=100, type=int)
    p.add_argument('--and_beta', type=float, default=1.0)
    p.add_argument('--and_m', type=int, default=5)
    p.add_argument('--denoiser', default='ffdnet')
    p.add_argument('--savepath', default="./save/test/")
    p.add_argument('--loadpath', default=None)
    p.add_argument('--testpath', default="./data/test_gray/")
    p.add_argument('--inference', default='True')
    p.add_argument('--gpu_ids', default='0')
    p.add_argument('--lr', type=float, default=0.0001)
    p.add_argument('--lr_gamma', type=float, default=0.9)
    p.add_argument('--sched_step', type=int, default=10)
    p.add_argument('--trainpath', default="./data/train/")
    p.add_argument('--print_every_n_steps', type=int, default=1)
    p.add_argument('--save_every_n_steps', type=int, default=50)
    args = p.parse_args(argv)
    if not torch.cuda.is_available():
        sys.exit("deqsci_amd needs an MI355X: there is no CPU path")
    torch.cuda.set_device(int(str(args.gpu_ids).split(',')[0]))
    if str(args.inference).lower() in ('false', '0', ''):
        return train(args)
    loadpath = args.loadpath or checkpoint.shipped({'ffdnet': 'ffdnet_gray', 'SimpleCNN': 'cnn',
                                                    'RealSN_SimpleCNN': 'rsn_cnn'}[args.denoiser])
    _, deq = build_pipeline(args.denoiser, loadpath, args.and_maxiters, args.and_m, args.and_beta)
    print('loaded dict!')
    loader = torch.utils.data.DataLoader(dataset=SCITestDataset(args.testpath), batch_size=1, shuffle=False, drop_last=True)
    os.makedirs(args.savepath, exist_ok=True)
    t0 = time.time()
    avg, images = test_solver_sci(deq, test_dataloader=loader, save_img_path=args.savepath)
    dt = time.time() - t0
    print(f"{len(images)} frames in {dt:.2f} s -> {len(images) / dt:.2f} frames/s (incl. PNG export)")
    return avg


if __name__ == "__main__":
    main()
