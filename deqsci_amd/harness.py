"""Evaluation harness: clips in, reconstructions + PSNR out.

The unit of work here is a CLIP (one .mat file: mask, M snapshot measurements, 8*M ground-truth frames).
`reconstruct_clip` hands ALL measurements of a clip to the DEQ module as one batch sharing the clip's mask
(the reference feeds them one by one, training/sci_equilibrium_training.py:171-181; rows of a batch are
independent problems, so the results are the same unless the whole-batch tolerance test fires - SURVEY 8(a)
caveat), optionally sharded over the ranks of a process group (deqsci_amd.distributed).  `evaluate` walks a
directory of clips; `test_solver_sci` is a thin adapter with the reference's signature, return value,
printed lines and PNG naming (ibid. :152-205) on top of the two.

    load_test_data / SCITestDataset   utils/sci_dataloader.py:241-274 (MATLAB v5 files, sorted file order)
    psnr                              skimage.metrics.peak_signal_noise_ratio for float input, data range 1
    frame_payload                     the float image the reference hands to cv2.imwrite (ibid. :19-21)
"""
import math
import os
from dataclasses import dataclass, field

import numpy as np
import torch

from . import distributed, operators

FIRST_MEASUREMENT_ONLY = ("drop", "runner")      # clips the reference scores on snapshot 0 only (:167-168)


def load_test_data(matfile):
    """-> {'gt' (H,W,B*M) in [0,1], 'mask' (H,W,B), 'meas' (H,W,M) scaled by 1/255}, all float32."""
    import scipy.io as sio
    try:
        f = sio.loadmat(matfile)
        meas, mask, orig = np.float32(f['meas']), np.float32(f['mask']), np.float32(f['orig'])
    except NotImplementedError as e:                      # MATLAB v7.3 = HDF5 (utils/sci_dataloader.py:249-254)
        try:
            import h5py
        except ImportError:
            raise NotImplementedError(f"{matfile}: MATLAB v7.3 files need h5py, which is not installed") from e
        with h5py.File(matfile, 'r') as f:                # HDF5 keeps MATLAB's column-major order: transpose back
            meas = np.float32(f['meas']).transpose()
            mask = np.float32(f['mask']).transpose()
            orig = np.float32(f['orig']).transpose()
    return {'gt': orig / 255, 'mask': mask, 'meas': meas / 255}


def directory_filelist(target_directory):
    return [f for f in sorted(os.listdir(target_directory))
            if os.path.isfile(os.path.join(target_directory, f)) and not f.startswith('.')]


class SCITestDataset(torch.utils.data.Dataset):
    def __init__(self, dir):
        self.dir = dir
        self.filelist = directory_filelist(dir)

    def __len__(self):
        return len(self.filelist)

    def __getitem__(self, item):
        data = load_test_data(os.path.join(self.dir, self.filelist[item]))
        data['file'] = self.filelist[item]
        return data


def psnr(rec, gt):
    a = np.asarray(gt, dtype=np.float32)
    b = np.asarray(rec, dtype=np.float32)
    return 10.0 * math.log10(1.0 / np.mean((a - b) ** 2, dtype=np.float64))


def clip_psnr(rec, gt, ids):
    """PSNR of every scored measurement of a clip: rec (M,H,W,B) on its device, gt (H,W,B*Mall) on the host.  The same
    arithmetic as `psnr` (float32 difference and square, float64 mean), evaluated where rec lives so that only M scalars
    cross PCIe."""
    B = rec.shape[-1]
    g = torch.stack([torch.as_tensor(gt[..., B * m:B * (m + 1)]) for m in ids]).to(rec.device, torch.float32)
    d = rec.detach().clamp(0, 1) - g
    mse = (d * d).double().mean(dim=(1, 2, 3))
    return [10.0 * math.log10(1.0 / float(v)) for v in mse.cpu()]


def frame_payload(frame):
    """One reconstructed frame (H,W) -> the (H,W,1) float image in [0,255] that goes to the PNG writer."""
    return (frame.detach().clamp(0, 1).cpu().numpy() * 255.)[:, :, None]


tensor_to_np = frame_payload          # the reference's name for it


def write_png(path, img):
    """cv2.imwrite(path, float image) casts with saturate_cast<uchar>(round-half-even); PIL is what is
    installed here, so that rounding is restated (np.rint = half-to-even, then clip) - unpinned: no cv2."""
    from PIL import Image
    a = np.clip(np.rint(np.asarray(img, dtype=np.float64)), 0, 255).astype(np.uint8)
    Image.fromarray(a[..., 0] if a.ndim == 3 else a).save(path)


# ----------------------------------------------------------------------------- clips
@dataclass
class ClipResult:
    name: str
    rec: torch.Tensor                 # (M,H,W,B) reconstructions of the scored measurements, on the device
    psnr: list                        # per measurement, dB
    res: list                         # per measurement: relative fixed-point residual at exit (None if unknown)
    frames: int = 0
    seconds: float = 0.0
    info: dict = field(default_factory=dict)

    @property
    def mean_psnr(self):
        return sum(self.psnr) / len(self.psnr)


def as_clip(sample):
    """A dataset item, with or without the DataLoader's leading batch dimension of 1 -> plain (H,W,*) tensors."""
    name = sample['file']
    if isinstance(name, (list, tuple)):
        name = name[0]

    def plain(v, nd):
        t = torch.as_tensor(v)
        return t[0] if t.dim() == nd + 1 else t
    return {'file': name, 'gt': plain(sample['gt'], 3), 'mask': plain(sample['mask'], 3), 'meas': plain(sample['meas'], 3)}


def scored_measurements(name, n_meas):
    """Indices of the snapshot measurements the benchmark scores for this clip."""
    return [0] if any(k in name for k in FIRST_MEASUREMENT_ONLY) else list(range(n_meas))


def _residuals(deep_eq_module, n):
    eng = getattr(deep_eq_module, "_engine", None)
    info = eng[1].last_info if eng else None
    if info and len(info.get("res_per_sample", ())) == n:
        return list(info["res_per_sample"]), info
    r = getattr(deep_eq_module, "forward_res", None)
    if isinstance(r, list):
        r = r[-1] if r else None
    return [r] * n, (info or {})


def reconstruct_clip(deep_eq_module, clip, device="cuda", batch=True, group=None):
    """All scored measurements of one clip through `deep_eq_module.forward(y, Phi, Phi_sum, initial_point=, train_flag=False)`.
    batch=True: one call with y (M,H,W) and the shared mask (1,H,W,B); batch=False: M calls of batch 1 (the reference's
    schedule).  With a process group the measurements are sharded over its ranks and all-gathered."""
    import time
    clip = as_clip(clip)
    Phi = clip['mask'].to(device)[None].contiguous()                  # (1,H,W,B)
    B = Phi.shape[-1]
    ids = scored_measurements(clip['file'], clip['meas'].shape[-1])
    y = clip['meas'].to(device).permute(2, 0, 1)[ids].contiguous()    # (M,H,W)
    Phi_sum = operators.phi_sum(Phi)

    def run(y_part, Phi_part):
        with torch.no_grad():
            x0 = operators.initial_point(y_part, Phi_part, Phi_sum, None)
        rec = deep_eq_module.forward(y_part, Phi_part, Phi_sum, initial_point=x0, train_flag=False)
        res, _ = _residuals(deep_eq_module, y_part.shape[0])
        return rec.detach(), res

    t0 = time.perf_counter()
    res = []
    if batch:
        def run_collect(y_part, Phi_part):
            rec, r = run(y_part, Phi_part)
            res.extend(r)
            return rec
        rec = distributed.sharded_reconstruct(run_collect, y, Phi, group=group)
        res = distributed.gather_scalars(res, group=group)
    else:
        parts = []
        for i in range(len(ids)):
            r, rr = run(y[i:i + 1], Phi)
            parts.append(r)
            res.extend(rr)
        rec = torch.cat(parts)
    if rec.is_cuda:
        torch.cuda.synchronize(rec.device)
    dt = time.perf_counter() - t0
    ps = clip_psnr(rec, clip['gt'], ids)
    return ClipResult(name=clip['file'], rec=rec, psnr=ps, res=res, frames=B * len(ids), seconds=dt,
                      info={"measurements": ids, "batched": bool(batch)})


def reconstruct_clips_together(deep_eq_module, clips, device="cuda", group=None):
    """The scored measurements of SEVERAL clips of one frame size as ONE engine batch, every measurement with its own clip's mask
    ((M,H,W,B) masks: nothing couples the measurements of a batch - alpha, residual, the ranges of the split-fp16 activations are all per
    measurement - so a measurement's reconstruction is the one it gets in any other batch, bit for bit).  What the reference's loop over
    clips and measurements (training/sci_equilibrium_training.py:157,171) becomes when the device wants eight measurements per call: the
    three shipped clips (1 + 1 + 6 measurements) are one call.  -> [ClipResult] in the clips' order; a clip's `seconds` is its share of
    the call by frames."""
    import time
    clips = [as_clip(c) for c in clips]
    ids = [scored_measurements(c['file'], c['meas'].shape[-1]) for c in clips]
    Phi = torch.cat([c['mask'].to(device)[None].expand(len(i), -1, -1, -1) for c, i in zip(clips, ids)]).contiguous()      # (M,H,W,B)
    y = torch.cat([c['meas'].to(device).permute(2, 0, 1)[i] for c, i in zip(clips, ids)]).contiguous()                        # (M,H,W)
    B = Phi.shape[-1]
    res = []

    def run(y_part, Phi_part):
        Ps = operators.phi_sum(Phi_part)
        with torch.no_grad():
            x0 = operators.initial_point(y_part, Phi_part, Ps, None)
        rec = deep_eq_module.forward(y_part, Phi_part, Ps, initial_point=x0, train_flag=False)
        res.extend(_residuals(deep_eq_module, y_part.shape[0])[0])
        return rec.detach()
    t0 = time.perf_counter()
    rec = distributed.sharded_reconstruct(run, y, Phi, group=group)
    res = distributed.gather_scalars(res, group=group)
    if rec.is_cuda:
        torch.cuda.synchronize(rec.device)
    dt = time.perf_counter() - t0
    out, a = [], 0
    for c, i in zip(clips, ids):
        r = rec[a:a + len(i)]
        out.append(ClipResult(name=c['file'], rec=r, psnr=clip_psnr(r, c['gt'], i), res=list(res[a:a + len(i)]), frames=B * len(i),
                              seconds=dt * len(i) / y.shape[0], info={"measurements": i, "batched": "all"}))
        a += len(i)
    return out


def evaluate(deep_eq_module, clips, device="cuda", batch=True, group=None, on_clip=None):
    """-> (mean over clips of the clip's mean PSNR, [ClipResult]).  batch: False = one measurement per call (the reference's schedule),
    True = a clip's measurements per call, "all" = the measurements of consecutive clips of one frame size per call
    (reconstruct_clips_together: the three shipped clips are ONE call of eight measurements)."""
    results = []
    if batch == "all":
        pending = []

        def flush():
            if pending:
                for r in reconstruct_clips_together(deep_eq_module, pending, device=device, group=group):
                    results.append(r)
                    if on_clip is not None:
                        on_clip(r)
                del pending[:]
        for sample in clips:
            c = as_clip(sample)
            if pending and tuple(as_clip(pending[0])['mask'].shape) != tuple(c['mask'].shape):
                flush()
            pending.append(c)
        flush()
        return sum(r.mean_psnr for r in results) / len(results), results
    for sample in clips:
        r = reconstruct_clip(deep_eq_module, sample, device=device, batch=batch, group=group)
        results.append(r)
        if on_clip is not None:
            on_clip(r)
    return sum(r.mean_psnr for r in results) / len(results), results


def png_payloads(result, prefix=""):
    """{path: (H,W,1) float image} for every frame of a clip, named like the reference's export (:185-187):
    '<prefix><file>_reconstruction_<frame index within the scored frames>.png'."""
    out = {}
    B = result.rec.shape[-1]
    for i in range(result.rec.shape[0]):
        for b in range(B):
            out[prefix + '%s_reconstruction_%d.png' % (result.name, i * B + b)] = frame_payload(result.rec[i, :, :, b])
    return out


def test_solver_sci(deep_eq_module, test_dataloader=None, save_img_path=None, verbose=True, save_image=True,
                    device="cuda", records=None, batch_measurements=False):
    """Adapter with the reference's signature (training/sci_equilibrium_training.py:152): returns
    (average PSNR, {png path: float image}); prints one line per clip and the total; writes the PNGs.
    Default = the reference's schedule, one measurement per call (:171-181); batch_measurements="all" hands the measurements of all clips of
    one frame size to the engine as ONE batch (the three shipped clips: one call of eight); batch_measurements=True hands a clip's
    measurements to the engine as one batch (faster; on the chaotic FFDNet + Anderson @180 clip a different - equally valid -
    realisation, because the FFDNet head kernel is chosen by launch size)."""
    images = {}

    def on_clip(r):
        images.update(png_payloads(r, save_img_path or ""))
        if records is not None:
            for i, m in enumerate(r.info["measurements"]):
                records.append({"id": f"{r.name}:{m}", "psnr": r.psnr[i], "res": r.res[i], "rec": r.rec[i:i + 1].cpu()})
        if verbose:
            print([r.name], '  PSNR: %.2f dB' % r.mean_psnr)
    avg, _ = evaluate(deep_eq_module, test_dataloader, device=device, batch=batch_measurements, on_clip=on_clip)
    if verbose:
        print('---------------------------------', 'Total Average PSNR: %.2f dB' % avg)
    if save_image:
        for path, img in images.items():
            write_png(path, img)
    return avg, images
