"""Multi-GPU sharding of independent measurements.

Each (y, Phi) -> reconstruction is independent (the reference loops them serially,
training/sci_equilibrium_training.py:157,171), so a global batch of M measurements is cut into
contiguous slices, one per rank, every rank runs the single-GPU engine on its slice with no
data-path collective, and ONE all_gather_into_tensor (RCCL over xGMI; gloo in the CPU tests) at
the end assembles the (M,H,W,B) result on every rank.  One process per GPU.

    shard_bounds          contiguous slice of a rank
    gather_shards         local shard -> all ranks' shards (THE collective of the path; times itself)
    sharded_reconstruct   global batch on every rank -> slice, reconstruct, gather_shards   (harness: a clip is small)
    reconstruct_shard     the same for a rank that holds only its own slice                    (bench: nothing global materialised)
    launch_ranks          start N fresh child processes (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in their
                          environment) - what `bench.py --gpus N` and `cli --gpu_ids 0,1,..` use when
                          they are not already running under torch.distributed.run
    init_from_env         process-group set-up of one rank from that environment
"""
import os
import socket
import subprocess
import time

import torch
import torch.distributed as dist


def shard_bounds(M, world_size, rank):
    """Contiguous slice [lo, hi) of rank; every rank gets ceil(M/R) slots, the tail is padding."""
    per = -(-M // world_size)
    lo = min(rank * per, M)
    return lo, min(lo + per, M), per


class GatherTimer:
    """Accumulates the time spent in the path's one collective (reported next to the throughput)."""

    def __init__(self):
        self.seconds, self.calls, self._pending = 0.0, 0, []

    def add(self, start, stop):
        self._pending.append((start, stop))
        self.calls += 1

    def total_seconds(self):
        for a, b in self._pending:
            if isinstance(a, float):
                self.seconds += b - a
            else:                                   # torch.cuda.Event pair; caller has synchronised the device
                self.seconds += a.elapsed_time(b) * 1e-3
        self._pending = []
        return self.seconds


def _active(group=None):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def gather_shards(local, group=None, out=None, timer=None):
    """local (per,...) on every rank -> (R*per,...) on every rank, rank-major.  No process group (or a group of
    one) = identity.  `out` may be a preallocated result buffer.  With `timer`, the collective is bracketed by
    events on the current stream (perf_counter for CPU tensors)."""
    if not _active(group):
        return local
    R = dist.get_world_size(group)
    full = out if out is not None else torch.empty((R * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                                                     device=local.device)
    if timer is not None and local.is_cuda:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dist.all_gather_into_tensor(full, local, group=group)
        e1.record()
        timer.add(e0, e1)
    elif timer is not None:
        t0 = time.perf_counter()
        dist.all_gather_into_tensor(full, local, group=group)
        timer.add(t0, time.perf_counter())
    else:
        dist.all_gather_into_tensor(full, local, group=group)
    return full


def sharded_reconstruct(reconstruct_fn, y, Phi, group=None, timer=None, **kw):
    """y (M,H,W), Phi (M|1,H,W,B) hold the GLOBAL batch on every rank (or are generated
    identically); returns the (M,H,W,B) reconstruction on every rank.  `reconstruct_fn(y_local,
    Phi_local, **kw) -> (n,H,W,B)`."""
    if not _active(group):
        return reconstruct_fn(y, Phi, **kw)
    R, r = dist.get_world_size(group), dist.get_rank(group)
    M = y.shape[0]
    lo, hi, per = shard_bounds(M, R, r)
    shared = Phi.dim() == 3 or (Phi.shape[0] == 1 and M > 1)
    out_shape = (per,) + tuple(y.shape[1:]) + (Phi.shape[-1],)
    if hi - lo == per:
        local = reconstruct_fn(y[lo:hi].contiguous(), Phi if shared else Phi[lo:hi].contiguous(), **kw).contiguous()
    else:                                           # ragged tail / idle rank: pad the shard with zeros
        local = torch.zeros(out_shape, dtype=torch.float32, device=y.device)
        if hi > lo:
            local[:hi - lo] = reconstruct_fn(y[lo:hi].contiguous(), Phi if shared else Phi[lo:hi].contiguous(), **kw)
    return gather_shards(local, group=group, timer=timer)[:M]


def reconstruct_shard(reconstruct_fn, y_local, Phi_local, M, group=None, timer=None, **kw):
    """The same path for a rank that holds ONLY its own slice (bench.py: nothing of size M is ever materialised on one GPU):
    y_local (hi - lo, H, W), Phi_local (hi - lo | 1, H, W, B) are rank r's measurements [lo, hi) = shard_bounds(M, R, r); returns
    the (M,H,W,B) reconstruction on every rank through the path's one all-gather."""
    if not _active(group):
        return reconstruct_fn(y_local, Phi_local, **kw)
    R, r = dist.get_world_size(group), dist.get_rank(group)
    lo, hi, per = shard_bounds(M, R, r)
    if y_local.shape[0] != hi - lo:
        raise ValueError(f"rank {r} of {R} owns measurements [{lo}, {hi}) of {M} but was handed {y_local.shape[0]}")
    if hi - lo == per:
        local = reconstruct_fn(y_local, Phi_local, **kw).contiguous()
    else:                                           # ragged tail / idle rank: pad the shard with zeros
        local = torch.zeros((per,) + tuple(y_local.shape[1:]) + (Phi_local.shape[-1],), dtype=torch.float32, device=y_local.device)
        if hi > lo:
            local[:hi - lo] = reconstruct_fn(y_local, Phi_local, **kw)
    return gather_shards(local, group=group, timer=timer)[:M]


def gather_scalars(values, group=None):
    """Per-measurement scalars (PSNR, res, ...) of the local shard -> list over all ranks."""
    if not _active(group):
        return list(values)
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, list(values), group=group)
    return [v for part in out for v in part]


# ----------------------------------------------------------------------------- launching ranks
def visible_gpu_count():
    """Number of GPUs a child process will see, WITHOUT touching HIP in this process (a launcher parent that has initialised the
    GPU must not spawn-and-wait on this pool, and `torch.cuda.device_count()` is one driver call away from that): the visibility
    variables if set, else the KFD topology (nodes with SIMDs are GPUs).  None when neither source exists."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    root = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir(root):
        return None
    n = 0
    for node in os.listdir(root):
        try:
            with open(os.path.join(root, node, "properties")) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    return n


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _stop(procs, grace=5.0):
    """terminate -> wait up to `grace` seconds -> kill -> wait: no rank is left running or as a zombie."""
    for p in procs:
        if p.poll() is None:
            p.terminate()
    deadline = time.time() + grace
    for p in procs:
        try:
            p.wait(timeout=max(0.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
    for p in procs:
        try:
            p.wait(timeout=grace)
        except subprocess.TimeoutExpired:
            pass


def _tail(path, lines=25):
    try:
        with open(path, errors="replace") as fh:
            return "".join(fh.readlines()[-lines:])
    except OSError:
        return ""


def launch_ranks(argv, n, device_ids=None, timeout=None, _attempts=2):
    """Run `argv` (a full command line) as n child processes, rank r with RANK=r, LOCAL_RANK=device_ids[r] (default r),
    WORLD_SIZE=n and a fresh 127.0.0.1 rendezvous port.  Children are NEW processes (never an exec of this one); rank 0 inherits
    stdout; every rank's stderr goes to a scratch file - rank 0's is forwarded LIVE (warnings and progress of long runs), the other
    ranks' when the run ends; when a rank fails, the stderr tail of EVERY rank is printed, the failing one first (the root cause is
    often on another rank than the first to exit).  Returns 0, or the exit code of the first rank that failed (128 + signal number for
    a rank killed by a signal, 124 on timeout); when a rank fails - or this process is interrupted / terminated - the others are
    terminated, then killed, and reaped, and the logs are still echoed and removed (all of it in the finally block).
    A rendezvous that loses the race for its port (EADDRINUSE between free_port() and the bind of rank 0) is retried once on
    a new port; nothing else is ever retried."""
    import shutil
    import signal
    import sys
    import tempfile
    port = free_port()
    procs, logs = [], []
    scratch = tempfile.mkdtemp(prefix="deqsci_ranks_")
    forwarded = [0]                                      # bytes of rank 0's stderr already passed through
    import codecs
    decoder = codecs.getincrementaldecoder("utf-8")(errors="replace")    # (a character split between two polls is completed by the next one)

    def forward_rank0():
        try:
            with open(logs[0], "rb") as fh:
                fh.seek(forwarded[0])
                chunk = fh.read()
        except (OSError, IndexError):
            return
        if chunk:
            forwarded[0] += len(chunk)
            sys.stderr.write(decoder.decode(chunk))
            sys.stderr.flush()

    def on_term(signum, frame):
        raise KeyboardInterrupt(f"signal {signum}")
    old = None
    try:
        old = signal.signal(signal.SIGTERM, on_term)
    except ValueError:                                   # not the main thread: the try/finally below still reaps on exceptions
        pass
    first_bad, code, retry, finished = None, 0, False, False
    try:
        for r in range(n):
            env = dict(os.environ)
            env.update(RANK=str(r), LOCAL_RANK=str(r if device_ids is None else device_ids[r]), WORLD_SIZE=str(n),
                       LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC is the only one this driver supports
            logs.append(os.path.join(scratch, f"rank{r}.stderr"))
            with open(logs[-1], "w") as errf:
                procs.append(subprocess.Popen(list(argv), env=env, stdout=None if r == 0 else subprocess.DEVNULL, stderr=errf))
        deadline = None if timeout is None else time.time() + timeout
        live = list(range(n))
        while live and first_bad is None:
            for r in list(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.remove(r)
                if rc != 0 and first_bad is None:
                    first_bad, code = r, (128 - rc if rc < 0 else rc)
            if deadline is not None and time.time() > deadline and first_bad is None and live:
                first_bad, code = live[0], 124
            forward_rank0()
            if live and first_bad is None:
                time.sleep(0.05)
        finished = True
    finally:
        _stop(procs)
        if old is not None:
            signal.signal(signal.SIGTERM, old)
        try:
            if first_bad is not None:
                text = _tail(logs[first_bad], 60)
                retry = _attempts > 1 and code != 124 and ("EADDRINUSE" in text or "ddress already in use" in text)
                if not retry:
                    forward_rank0()
                    sys.stderr.write(f"[launch_ranks] rank {first_bad} of {n} {'timed out' if code == 124 else 'failed'} (exit code {code}); the other "
                                     "ranks were stopped.  " + ("Its stderr is above.\n" if first_bad == 0 else f"Its stderr tail:\n{text}"))
                    for r in range(1, len(logs)):            # (rank 0's went through live)
                        other = _tail(logs[r], 25) if r != first_bad else ""
                        if other:
                            sys.stderr.write(f"[launch_ranks] stderr tail of rank {r}:\n{other}")
            else:
                forward_rank0()                              # a clean (or interrupted) run: the rest of rank 0, then the other ranks' stderr
                for r in range(1, len(logs)):
                    sys.stderr.write(_tail(logs[r], 10 ** 6))
                if not finished:
                    sys.stderr.write(f"[launch_ranks] interrupted: {len(procs)} rank(s) stopped\n")
        finally:
            shutil.rmtree(scratch, ignore_errors=True)
    if retry:
        sys.stderr.write(f"[launch_ranks] the rendezvous lost the race for port {port} (address already in use; rank {first_bad}'s traceback may be "
                         "above): starting the ranks again on a new port\n")
        return launch_ranks(argv, n, device_ids=device_ids, timeout=timeout, _attempts=_attempts - 1)
    return code


def init_from_env(backend=None):
    """-> (rank, world, local_rank, device).  Creates the default process group when WORLD_SIZE > 1
    (backend "nccl" = RCCL on a GPU box, "gloo" for CPU plumbing tests)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    elif backend == "gloo+cuda0":
        # test rig for a ONE-GPU box: every rank on cuda:0, the collective through gloo (RCCL refuses two ranks on one device).
        # Exercises the real engine + sharding + all-gather plumbing with world size > 1; says nothing about RCCL or speed.
        torch.cuda.set_device(0)
        device, backend = torch.device("cuda", 0), "gloo"
    else:
        device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    return rank, world, local_rank, device


def relaunch_needed(n):
    """True when n > 1 ranks were asked for and this process is not already one of them."""
    return n > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1
