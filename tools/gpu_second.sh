#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/denoiser_bench.py > gpurun_out/denoiser_bench.log 2>&1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --iters 40 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof_bench | head -20
cat gpurun_out/denoiser_bench.log
python - <<'PY'
import time, torch, sys
sys.path.insert(0,'.')
from oracle import deqsci_oracle as orc
g = torch.Generator().manual_seed(1)
Phi=(torch.rand(1,256,256,8,generator=g)<0.5).float(); x=torch.rand(1,256,256,8,generator=g); y=orc.sci_forward(x,Phi); Ps=orc.phi_sum(Phi)
for kind in ("ffdnet","SimpleCNN"):
    for th in (8,16,32,64,128):
        torch.set_num_threads(th)
        f=orc.ProxGradSCI(kind); z=orc.initial_point(y,Phi)
        f(z,y,Phi,Ps)
        t0=time.perf_counter()
        for _ in range(3): f(z,y,Phi,Ps)
        print(kind,"threads",th,"s/call",(time.perf_counter()-t0)/3, flush=True)
PY
