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



def _bench_worker(rank, world, port, q):
    """bench.py's own step() plumbing (make_step -> sharded_reconstruct -> gather_shards) with a stub engine."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import bench
    from deqsci_amd import distributed
    r, w, _, dev = distributed.init_from_env("gloo")
    assert (r, w, dev.type) == (rank, world, "cpu")
    # every rank generates ONLY its slice, by global measurement index; the whole batch (any sharding) is the same batch
    y_all, Phi_all, _ = bench.make_batch(0, world * 3, 8, 6, 4, 1234, dev)
    y, Phi, _ = bench.make_batch(rank * 3, rank * 3 + 3, 8, 6, 4, 1234, dev)
    assert torch.equal(y, y_all[rank * 3:rank * 3 + 3]) and torch.equal(Phi, Phi_all[rank * 3:rank * 3 + 3])
    calls = []

    class Stub:
        def reconstruct(self, yl, Pl):
            calls.append(tuple(yl.shape))
            return (yl.unsqueeze(-1) * Pl + rank).contiguous()
    timer = distributed.GatherTimer()
    step = bench.make_step(Stub(), y, Phi, world * 3, timer)
    full = step()
    want = torch.cat([y_all[i * 3:(i + 1) * 3].unsqueeze(-1) * Phi_all[i * 3:(i + 1) * 3] + i for i in range(world)])
    ok = torch.equal(full, want) and calls == [(3, 8, 6)] and timer.calls == 1 and timer.total_seconds() > 0
    # ragged global batch (strong scaling with M not divisible by the world size): rank 1 owns fewer, the tail is padding
    lo, hi, per = distributed.shard_bounds(5, world, rank)
    y5, P5, _ = bench.make_batch(lo, hi, 8, 6, 4, 1234, dev)
    full5 = distributed.reconstruct_shard(Stub().reconstruct, y5, P5, 5)
    want5 = torch.cat([y_all[i * 3:min(i * 3 + 3, 5)].unsqueeze(-1) * Phi_all[i * 3:min(i * 3 + 3, 5)] + i for i in range(world)])
    ok = ok and (lo, hi, per) == ((0, 3, 3) if rank == 0 else (3, 5, 3)) and torch.equal(full5, want5)
    try:
        distributed.reconstruct_shard(Stub().reconstruct, y5[:1], P5[:1], 5)          # a slice of the wrong size is refused
        ok = False
    except ValueError:
        pass
    q.put((rank, bool(ok), tuple(full.shape)))
    dist.destroy_process_group()


def test_bench_step_plumbing_two_ranks():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(ok for _, ok, _ in res), res


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without torchrun must start two ranks itself: exercised end to end with the
    CPU plumbing self-test (gloo, stub engine - the GPU box runs the same launcher with nccl and the real engine)."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-selftest", "--batch-per-gpu", "3",
                          "--size", "16x12x8", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                               # ONE JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 6 and rec["steps"] == 2 and rec["data"] == "selftest"
    assert rec["scaling"] == "weak" and rec["config"]["batch_per_gpu"] == 3
    assert rec["allgather_ms_per_step"] > 0 and rec["allgather_bytes_per_step"] == 6 * 16 * 12 * 8 * 4
    # the N > 1 line proves what the collective library saw: backend, world size as the process group reports it,
    # one record per rank with its device, its own slice and its own time
    d = rec["distributed"]
    assert d["backend"] == "gloo" and d["world_size"] == 2 and [r["rank"] for r in d["ranks"]] == [0, 1]
    assert [r["measurements"] for r in d["ranks"]] == [[0, 3], [3, 6]] and all(r["ms_per_step"] > 0 for r in d["ranks"])
    assert max(r["ms_per_step"] for r in d["ranks"]) <= rec["ms_per_step"] * 1.0001 and d["distinct_devices"] >= 1


def test_bench_strong_scaling_mode_two_ranks():
    """`--global-batch M`: the total is fixed and sharded (BASELINE config 3 as stated), ragged M included."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-selftest", "--global-batch", "5",
                          "--size", "16x12x8", "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["scaling"] == "strong" and rec["n_gpus"] == 2
    assert rec["config"]["global_batch"] == 5 and rec["config"]["batch_per_gpu"] == 3
    assert rec["allgather_bytes_per_step"] == 2 * 3 * 16 * 12 * 8 * 4


def test_bench_config3_eight_ranks_global_batch_64():
    """The only N > 1 the target machine has is 8, and BASELINE config 3 is 64 measurements over 8 GPUs: the launcher starting EIGHT ranks,
    their rendezvous on one port, the shard bounds, the gather bytes and the reaping, end to end (gloo, stub engine, 8 CPU processes -
    torch is imported eight times: slow, once)."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--plumbing-selftest", "--global-batch", "64",
                          "--size", "16x12x8", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["scaling"] == "strong" and rec["config"]["global_batch"] == 64 and rec["config"]["batch_per_gpu"] == 8
    assert rec["allgather_bytes_per_step"] == 64 * 16 * 12 * 8 * 4 and rec["allgather_ms_per_step"] > 0
    d = rec["distributed"]
    assert d["backend"] == "gloo" and d["world_size"] == 8 and [r["rank"] for r in d["ranks"]] == list(range(8))
    assert [r["measurements"] for r in d["ranks"]] == [[8 * r, 8 * r + 8] for r in range(8)]
    assert [r["local_rank"] for r in d["ranks"]] == list(range(8))
    # weak scaling, the driver's default form at N = 8: 8 measurements per GPU = the same 64
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--plumbing-selftest", "--size", "16x12x8", "--steps", "1",
                          "--warmup", "0"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 8 and rec["scaling"] == "weak" and rec["config"]["global_batch"] == 64 and rec["config"]["batch_per_gpu"] == 8


def test_bench_under_torch_distributed_run():
    """The driver's launch form: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P
    bench.py --gpus 2 ...` - bench.py must then run as ONE rank of the existing job (no second launcher), rank 0 prints the one line."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    from deqsci_amd.distributed import free_port
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-selftest", "--batch-per-gpu", "2",
           "--size", "16x12x8", "--steps", "1", "--warmup", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 4 and rec["scaling"] == "weak" and rec["allgather_ms_per_step"] > 0


def test_launch_ranks_propagates_failure(capfd):
    import sys
    import time
    from deqsci_amd.distributed import launch_ranks
    code = ("import os,sys,time; r=int(os.environ['RANK']); assert os.environ['WORLD_SIZE']=='2'; time.sleep(0.2 if r==0 else 30); "
            "print('rank', r, 'says goodbye', file=sys.stderr); sys.exit(3 if r==0 else 0)")
    t0 = time.time()
    assert launch_ranks([sys.executable, "-c", code], 2, timeout=60) == 3
    assert time.time() - t0 < 20                                          # the sleeping rank was stopped, not waited for
    err = capfd.readouterr().err
    assert "rank 0 of 2 failed (exit code 3)" in err and "rank 0 says goodbye" in err      # the failing rank's stderr tail is shown
    assert launch_ranks([sys.executable, "-c", "import os; assert os.environ['LOCAL_RANK'] in ('5','7')"], 2, device_ids=[5, 7], timeout=60) == 0
    # a rank killed by a signal is reported as 128 + signal, a hung run as 124 - and nothing is left running either way
    assert launch_ranks([sys.executable, "-c", "import os,signal,time; os.kill(os.getpid(), signal.SIGKILL) if os.environ['RANK']=='1' else time.sleep(30)"],
                        2, timeout=60) == 128 + 9
    assert launch_ranks([sys.executable, "-c", "import time; time.sleep(30)"], 2, timeout=1) == 124


def test_launch_ranks_reaps_children_when_interrupted():
    """SIGTERM to the launcher: the ranks are terminated and reaped, the launcher dies with the signal's exception."""
    import signal
    import subprocess
    import sys
    import time
    from conftest import ROOT
    prog = ("import sys, os; sys.path.insert(0, %r)\n"
            "from deqsci_amd.distributed import launch_ranks\n"
            "child = 'import os,time; open(os.environ[\"PIDFILE\"] + os.environ[\"RANK\"], \"w\").write(str(os.getpid())); time.sleep(60)'\n"
            "launch_ranks([sys.executable, '-c', child], 2)\n") % ROOT
    import tempfile
    d = tempfile.mkdtemp()
    env = dict(os.environ, PIDFILE=os.path.join(d, "pid"))
    p = subprocess.Popen([sys.executable, "-c", prog], env=env, stderr=subprocess.DEVNULL)
    deadline = time.time() + 60
    while time.time() < deadline and not all(os.path.exists(os.path.join(d, f"pid{r}")) and os.path.getsize(os.path.join(d, f"pid{r}")) for r in range(2)):
        time.sleep(0.1)
    pids = [int(open(os.path.join(d, f"pid{r}")).read()) for r in range(2)]
    p.send_signal(signal.SIGTERM)
    p.wait(timeout=30)
    assert p.returncode != 0
    time.sleep(0.2)
    for pid in pids:
        try:
            os.kill(pid, 0)
            alive = True
        except OSError:
            alive = False
        assert not alive, f"rank process {pid} survived its launcher"


def test_visible_gpu_count_never_touches_hip(monkeypatch):
    from deqsci_amd import distributed
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2,3")
    assert distributed.visible_gpu_count() == 4
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert distributed.visible_gpu_count() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "2")
    assert distributed.visible_gpu_count() == 1
    # the launcher parents (bench.py, cli.py) must not call torch.cuda at all
    from conftest import ROOT
    for f in ("bench.py", os.path.join("deqsci_amd", "cli.py")):
        src = open(os.path.join(ROOT, f)).read()
        main = src[src.index("def main("):]
        launcher = main[:main.index("launch_ranks(")]
        assert "torch.cuda" not in launcher, f


class _FakeDEQ:
    """Stands in for DEQFixedPoint (which needs a GPU): a deterministic function of its inputs with a residual attribute."""
    forward_res = None

    def forward(self, y, Phi, Phi_sum, initial_point=None, train_flag=False):
        self.forward_res = float(y.mean())
        return (0.5 * initial_point + 0.1 * Phi / Phi_sum.unsqueeze(-1)).contiguous()


def _patch_operators():
    """The product's operators are HIP-only; here the oracle's CPU restatement is patched in so that the harness -> sharding
    -> all-gather plumbing can run under gloo."""
    from oracle import deqsci_oracle as orc
    from deqsci_amd import harness
    harness.operators.phi_sum = orc.phi_sum
    harness.operators.initial_point = lambda y, Phi, Phi_sum=None, gt=None: orc.sci_adjoint(y, Phi.expand(y.shape[0], -1, -1, -1))


def _harness_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import deqsci_oracle as orc
    from deqsci_amd import harness
    _patch_operators()
    ds = harness.SCITestDataset(orc.DATA_DIR)
    out = []
    for clip in (ds[0], ds[2]):                                # drop8: 1 scored measurement (rank 1 idles), traffic: 6 (3 per rank)
        r = harness.reconstruct_clip(_FakeDEQ(), clip, device="cpu")
        out.append((r.name, r.rec.clone(), list(r.psnr), list(r.res), r.frames))
    q.put((rank, [(n, rec.numpy(), p, s, f) for n, rec, p, s, f in out]))
    dist.destroy_process_group()


def test_harness_shards_a_clip_over_two_ranks():
    """reconstruct_clip under a process group: the clip's measurements are cut into contiguous slices, each rank reconstructs
    its slice, one all-gather (+ one for the scalars) - same reconstructions, PSNRs and residual list as a single process."""
    import numpy as np
    from oracle import deqsci_oracle as orc
    from deqsci_amd import harness
    _patch_operators()
    ds = harness.SCITestDataset(orc.DATA_DIR)
    want = [harness.reconstruct_clip(_FakeDEQ(), clip, device="cpu", batch=False) for clip in (ds[0], ds[2])]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_harness_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(30)
    for rank, clips in res:
        for (name, rec, psnr, resid, frames), w in zip(clips, want):
            assert name == w.name and frames == w.frames == 8 * len(w.psnr)
            assert np.array_equal(rec, w.rec.numpy()), (rank, name)
            assert psnr == w.psnr and len(resid) == len(w.res)
            # residuals: one value per scored measurement, in measurement order (each rank reports its own slice)
            assert all(isinstance(v, float) for v in resid)
