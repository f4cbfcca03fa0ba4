"""The reference's training loop for the DEQ-SCI solver, on the HIP path.

    train_solver_sci(...)      training/sci_equilibrium_training.py:28-150

Same arguments and the same sequence of events per step and per epoch: zero_grad, Phi_sum, x0 = At(y, Phi), the DEQ
forward with a tape (deqsci_amd.solvers.DEQFixedPoint: solve without tape, taped f, implicit-differentiation backward
hook), loss, NaN guard, backward, optimizer step, PSNR of the clipped reconstruction, log line every
`print_every_n_steps`, evaluation + 'best.ckpt' every `save_every_n_steps`; per epoch: mean PSNR, loss-explosion guard
(reload on the next epoch), scheduler step, 'epoch_<e>.ckpt' with the reference's four keys, evaluation.
Scalars go to TensorBoard when torch.utils.tensorboard is importable (it is an optional dependency here).
Not in the reference: `history` (a list that receives one dict per step) and `device`.
"""
import time
from collections import OrderedDict

import numpy as np
import torch

from . import harness, operators


def _summary_writer(path):
    try:
        from torch.utils import tensorboard
        return tensorboard.SummaryWriter(path)
    except Exception:                                          # tensorboard absent: train without event files
        return None


def _save(path, solver, epoch, optimizer, scheduler):
    torch.save({'solver_state_dict': solver.state_dict(), 'epoch': epoch,
                'optimizer_state_dict': optimizer.state_dict(), 'scheduler_state_dict': scheduler.state_dict()}, path)


def _evaluate(deep_eq_module, loader, img_path, device, **kw):
    return harness.test_solver_sci(test_dataloader=loader, deep_eq_module=deep_eq_module, save_img_path=img_path, device=device, **kw)


def _step(batch, deep_eq_module, loss_function, device):
    """One forward of the training loop (:55-69): batch to the GPU, Phi_sum, x0 = At(y, Phi) without tape, taped DEQ forward."""
    gt = batch['gt'].to(device)
    y, Phi = batch['meas'].to(device), batch['mask'].to(device)
    Phi_sum = operators.phi_sum(Phi)
    with torch.no_grad():
        x0 = operators.initial_point(y, Phi, Phi_sum, gt)
    rec = deep_eq_module.forward(y, Phi, Phi_sum, initial_point=x0)
    return rec, gt, loss_function(rec, gt)


def train_solver_sci(single_iterate_solver, train_dataloader, optimizer, save_model_path, loss_function, n_epochs,
                     deep_eq_module, use_dataparallel=False, scheduler=None, print_every_n_steps=100,
                     save_every_n_steps=1000, start_epoch=0, test_dataloader=None, train_img_path=None,
                     test_img_path=None, best_img_path=None, tflog_path=None, device="cuda", history=None):
    t_start = time.time()
    writer = _summary_writer(tflog_path)
    seen, best_psnr, first_loss, reload_next = 0, 0, 10.0, False
    for epoch in range(start_epoch, n_epochs):
        if reload_next:                                        # the loss blew up last epoch: back to the saved state (:45-48)
            saved = torch.load(save_model_path, map_location=device, weights_only=False)
            single_iterate_solver.load_state_dict(saved['solver_state_dict'])
            optimizer.load_state_dict(saved['optimizer_state_dict'])
            reload_next = False
        epoch_psnr, loss = 0, None
        for step, batch in enumerate(train_dataloader):
            seen += batch['gt'].size(0)
            optimizer.zero_grad()
            rec, gt, loss = _step(batch, deep_eq_module, loss_function, device)
            if np.isnan(loss.item()):
                print('Loss is nan!')
                reload_next = True
                break
            loss.backward()
            optimizer.step()
            if step == 0:
                first_loss = loss.item()
            cur = harness.psnr(rec.clip(0, 1).cpu().detach().numpy(), gt.cpu().detach().numpy())
            epoch_psnr += cur
            lr = optimizer.param_groups[0]['lr']
            stats = OrderedDict([('main/PSNR', cur), ('main/loss', loss.mean().item()), ('config/lr', lr), ('main/best_PSNR', best_psnr)])
            if history is not None:
                history.append({"epoch": epoch, "step": step, "loss": stats['main/loss'], "psnr": cur, "lr": lr,
                                "forward_res": deep_eq_module.forward_res, "backward_res": getattr(deep_eq_module, "backward_res", None)})
            if writer is not None:
                for name, value in stats.items():
                    writer.add_scalar(name, value, global_step=int(seen), walltime=time.time() - t_start)
                writer.flush()
            if step % print_every_n_steps == 0:                # the reference's log line, character for character (:94-98)
                print(f"Epoch: {epoch} Step: {step} Loss: {loss.cpu().detach().numpy()} PSNR: {cur:2.2f} dB"
                      f" best PSNR (test): {best_psnr:2.2f} dB lr: {lr:.8f}", flush=True)
            if (step + 1) % save_every_n_steps == 0:           # mid-epoch evaluation; keep the best model and its images (:100-124)
                test_psnr, images = _evaluate(deep_eq_module, test_dataloader, best_img_path, device, verbose=True, save_image=False)
                if test_psnr > best_psnr:
                    best_psnr = test_psnr
                    for path, img in images.items():
                        harness.write_png(path, img)
                    print('saving best model')
                    _save(save_model_path + 'best.ckpt', single_iterate_solver, epoch, optimizer, scheduler)
        print('avg PSNR in epoch %d: %.2f dB' % (epoch, epoch_psnr / len(train_dataloader)))
        if (first_loss - loss.item()) / first_loss < -10.0 or np.isnan(loss.item()):   # exploded: reload next epoch (:137-138)
            reload_next = True
        scheduler.step()
        if not reload_next:
            _save(save_model_path + 'epoch_%d.ckpt' % epoch, single_iterate_solver, epoch, optimizer, scheduler)
            print('dict saved!')
        _evaluate(deep_eq_module, test_dataloader, test_img_path, device)
