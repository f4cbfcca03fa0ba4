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


def train_solver_sci(single_iterate_solver, train_dataloader, optimizer, save_model_path, loss_function, n_epochs,
                     deep_eq_module, use_dataparallel=False, scheduler=None, print_every_n_steps=100,
                     save_every_n_steps=1000, start_epoch=0, test_dataloader=None, train_img_path=None,
                     test_img_path=None, best_img_path=None, tflog_path=None, device="cuda", history=None):
    start_time = time.time()
    cur_nimg = 0
    writer = _summary_writer(tflog_path)
    previous_loss = 10.0
    reset_flag = False
    best_psnr = 0
    for epoch in range(start_epoch, n_epochs):
        if reset_flag:                                         # the loss blew up last epoch: back to the saved state (:45-48)
            saved = torch.load(save_model_path, map_location=device, weights_only=False)
            single_iterate_solver.load_state_dict(saved['solver_state_dict'])
            optimizer.load_state_dict(saved['optimizer_state_dict'])
        reset_flag = False
        psnr_sum = 0
        loss = None
        for ii, sample_batch in enumerate(train_dataloader):
            cur_nimg += sample_batch['gt'].size(0)
            optimizer.zero_grad()
            gt_batch = sample_batch['gt'].to(device)
            y = sample_batch['meas'].to(device)
            Phi = sample_batch['mask'].to(device)
            Phi_sum = operators.phi_sum(Phi)
            with torch.no_grad():
                x0 = operators.initial_point(y, Phi, Phi_sum, gt_batch)
            reconstruction = deep_eq_module.forward(y, Phi, Phi_sum, initial_point=x0)
            loss = loss_function(reconstruction, gt_batch)
            if np.isnan(loss.item()):
                print('Loss is nan!')
                reset_flag = True
                break
            loss.backward()
            optimizer.step()
            if ii == 0:
                previous_loss = loss.item()
            PSNR = harness.psnr(reconstruction.clip(0, 1).cpu().detach().numpy(), gt_batch.cpu().detach().numpy())
            psnr_sum += PSNR
            stats = OrderedDict([('main/PSNR', PSNR), ('main/loss', loss.mean().item()),
                                 ('config/lr', optimizer.param_groups[0]['lr']), ('main/best_PSNR', best_psnr)])
            if history is not None:
                history.append({"epoch": epoch, "step": ii, "loss": stats['main/loss'], "psnr": PSNR,
                                "lr": stats['config/lr'], "forward_res": deep_eq_module.forward_res,
                                "backward_res": getattr(deep_eq_module, "backward_res", None)})
            if writer is not None:
                walltime = time.time() - start_time
                for name, value in stats.items():
                    writer.add_scalar(name, value, global_step=int(cur_nimg), walltime=walltime)
                writer.flush()
            if ii % print_every_n_steps == 0:
                print("Epoch: " + str(epoch) + " Step: " + str(ii) + " Loss: " + str(loss.cpu().detach().numpy()) +
                      " PSNR: %2.2f dB" % PSNR + " best PSNR (test): %2.2f dB" % best_psnr +
                      " lr: %.8f" % optimizer.param_groups[0]['lr'], flush=True)
            if (ii + 1) % save_every_n_steps == 0:
                cur_psnr, all_images = harness.test_solver_sci(test_dataloader=test_dataloader, deep_eq_module=deep_eq_module,
                                                               save_img_path=best_img_path, verbose=True, save_image=False,
                                                               device=device)
                if cur_psnr > best_psnr:
                    best_psnr = cur_psnr
                    for k in all_images:
                        harness.write_png(k, all_images[k])
                    print('saving best model')
                    _save(save_model_path + 'best.ckpt', single_iterate_solver, epoch, optimizer, scheduler)
        avg_psnr = psnr_sum / len(train_dataloader)
        print('avg PSNR in epoch %d: %.2f dB' % (epoch, avg_psnr))
        if (previous_loss - loss.item()) / previous_loss < -10.0 or np.isnan(loss.item()):
            reset_flag = True
        scheduler.step()
        if not reset_flag:
            _save(save_model_path + 'epoch_%d.ckpt' % epoch, single_iterate_solver, epoch, optimizer, scheduler)
            print('dict saved!')
        harness.test_solver_sci(test_dataloader=test_dataloader, deep_eq_module=deep_eq_module, save_img_path=test_img_path,
                                device=device)
