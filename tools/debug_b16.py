import sys, torch
sys.path.insert(0, '.')
from oracle import deqsci_oracle as orc
from deqsci_amd import checkpoint
from deqsci_amd.cli import build_pipeline
from deqsci_amd.engine import DEQSCIEngine
net = build_pipeline("SimpleCNN", checkpoint.shipped("cnn"), 6)[0].nonlinear_op
W_ = orc.load_weights("cnn")
for (H, B) in ((64, 16), (128, 16), (256, 16), (512, 8), (512, 16)):
    g = torch.Generator().manual_seed(4)
    Phi = (torch.rand(1, H, H, B, generator=g) < 0.5).float()
    x = torch.rand(1, H, H, B, generator=g)
    y = orc.sci_forward(x, Phi); Ps = orc.phi_sum(Phi)
    # single denoiser call
    xin = x.permute(0, 3, 1, 2).reshape(B, 1, H, H).contiguous()
    with torch.no_grad():
        ref = orc.simplecnn_forward(W_, xin)
        got = net(xin.cuda()).cpu()
        got_cl = None
    e_net = float((got - ref).norm() / ref.norm())
    for it in (3, 6):
        want, wres = orc.deq_forward(orc.ProxGradSCI("SimpleCNN"), orc.andersonexp, y, Phi, Ps, orc.initial_point(y, Phi), m=5, beta=1.0, lam=1e-2, max_iter=it, tol=1e-5)
        for cl in (False, True):
            eng = DEQSCIEngine(net, max_iter=it, channels_last=cl)
            rec = eng.reconstruct(y.cuda(), Phi.cuda()).cpu()
            print(H, B, 'iters', it, 'cl', cl, 'net_err', f'{e_net:.2e}', 'e2e', f'{float((rec - want).norm() / want.norm()):.2e}', 'res', eng.last_info['res'], wres, flush=True)
