#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python tools/denoiser_bench.py 2>&1 | grep -v amdgpu.ids | head -12 > gpurun_out/denoiser_bench2.log
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_fused.log
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --channels-last 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_fused_cl.log
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fused-epilogue 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_unfused.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 1 --warmup 0 --iters 20 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/bench_torchrun1.log
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/pytest_gpu.log
cat gpurun_out/denoiser_bench2.log; for f in bench_fused bench_fused_cl bench_unfused bench_torchrun1; do echo $f; python - <<PY
import json
for l in open("gpurun_out/$f.log"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d.get("roofline",{}).get("avg_launch_us"), d.get("roofline",{}).get("frac"), d["final_res"])
    elif "rror" in l: print(l.strip()[:300])
PY
done; tail -3 gpurun_out/pytest_gpu.log
