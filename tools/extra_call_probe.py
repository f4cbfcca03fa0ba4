import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from deqsci_amd.engine import DEQSCIEngine
args = bench.parse_args([])
dev = torch.device("cuda:0")
y, Phi, _ = bench.make_batch(0, 8, 256, 256, 8, 1234, dev)
for kw in ({"extra_call": True}, {"extra_call": True, "groups": 1}, {}, {"groups": 1}):
    eng = bench.build_engine(args, dev, **kw)
    eng.reconstruct(y, Phi)
    torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); eng.reconstruct(y, Phi); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    li = eng.last_info
    print(kw, ["%.1f" % (1e3 * t) for t in ts], li["f_calls"], li["stack_launches"], li["stack_timeouts"], li.get("groups"), li["denoiser_path"], flush=True)
