"""Multi-GPU sharding of independent measurements.

Each (y, Phi) -> reconstruction is independent (the reference loops them serially,
training/sci_equilibrium_training.py:157,171), so a global batch of M measurements is cut into
contiguous slices, one per rank, every rank runs the single-GPU engine on its slice with no
data-path collective, and ONE all_gather_into_tensor (RCCL over xGMI; gloo in the CPU tests) at
the end assembles the (M,H,W,B) result on every rank.  One process per GPU.
"""
import torch
import torch.distributed as dist


def shard_bounds(M, world_size, rank):
    """Contiguous slice [lo, hi) of rank; every rank gets ceil(M/R) slots, the tail is padding."""
    per = -(-M // world_size)
    lo = min(rank * per, M)
    return lo, min(lo + per, M), per


def sharded_reconstruct(reconstruct_fn, y, Phi, group=None, **kw):
    """y (M,H,W), Phi (M|1,H,W,B) hold the GLOBAL batch on every rank (or are generated
    identically); returns the (M,H,W,B) reconstruction on every rank.  `reconstruct_fn(y_local,
    Phi_local, **kw) -> (n,H,W,B)`."""
    if not (dist.is_available() and dist.is_initialized()):
        return reconstruct_fn(y, Phi, **kw)
    R, r = dist.get_world_size(group), dist.get_rank(group)
    M = y.shape[0]
    lo, hi, per = shard_bounds(M, R, r)
    shared = Phi.dim() == 3 or (Phi.shape[0] == 1 and M > 1)
    out_shape = (per,) + tuple(y.shape[1:]) + (Phi.shape[-1],)
    local = torch.zeros(out_shape, dtype=torch.float32, device=y.device)
    if hi > lo:
        local[:hi - lo] = reconstruct_fn(y[lo:hi].contiguous(), Phi if shared else Phi[lo:hi].contiguous(), **kw)
    full = torch.empty((R * per,) + out_shape[1:], dtype=torch.float32, device=y.device)
    dist.all_gather_into_tensor(full, local, group=group)
    return full[:M]


def gather_scalars(values, group=None):
    """Per-measurement scalars (PSNR, res, ...) of the local shard -> list over all ranks."""
    if not (dist.is_available() and dist.is_initialized()):
        return list(values)
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, list(values), group=group)
    return [v for part in out for v in part]
