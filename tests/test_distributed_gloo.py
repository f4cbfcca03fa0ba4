"""CPU, world_size 2, gloo: the N>1 path - contiguous sharding of independent measurements with no
data-path collective and ONE all_gather_into_tensor at the end (RCCL on the GPU box)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _fake_reconstruct(y, Phi):
    """Stands in for DEQSCIEngine.reconstruct (which needs a GPU): deterministic per measurement."""
    return (y.unsqueeze(3) * Phi * 2.0 + 1.0).contiguous()


def _worker(rank, world, port, M, shared, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deqsci_amd.distributed import gather_scalars, shard_bounds, sharded_reconstruct
    g = torch.Generator().manual_seed(5)
    y = torch.rand(M, 6, 5, generator=g)
    Phi = torch.rand(1 if shared else M, 6, 5, 4, generator=g)
    calls = []

    def rec(yl, Pl):
        calls.append(yl.shape[0])
        return _fake_reconstruct(yl, Pl)
    full = sharded_reconstruct(rec, y, Phi)
    lo, hi, per = shard_bounds(M, world, rank)
    scal = gather_scalars([float(v) for v in y[lo:hi].mean((1, 2))])
    ok = torch.equal(full, _fake_reconstruct(y, Phi.expand(M, -1, -1, -1))) and calls == ([hi - lo] if hi > lo else [])
    ok = ok and torch.allclose(torch.tensor(scal), y.mean((1, 2)))
    q.put((rank, bool(ok), tuple(full.shape)))
    dist.destroy_process_group()


def _run(M, shared):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, M, shared, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(ok for _, ok, _ in res), res
    assert all(shape[0] == M for _, _, shape in res)


def test_sharded_reconstruct_even_batch():
    _run(8, shared=False)


def test_sharded_reconstruct_ragged_batch_shared_mask():
    _run(5, shared=True)


def test_single_measurement_two_ranks():
    _run(1, shared=False)
