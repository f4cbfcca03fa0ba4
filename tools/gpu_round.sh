#!/bin/bash
# full GPU check: tests, smoke, kernel microbench, bench, rocprof stats (CSV)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids > gpurun_out/smoke.log
timeout 300 python tools/kernel_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/kernel_bench.log
timeout 300 python tools/kernel_bench.py --bsz 8 2>&1 | grep -v amdgpu.ids > gpurun_out/kernel_bench_bsz8.log
timeout 300 python tools/kernel_bench.py --bsz 8 --size 512x512x16 2>&1 | grep -v amdgpu.ids > gpurun_out/kernel_bench_512.log
timeout 900 python bench.py --steps 2 --warmup 1 2>&1 | grep -v amdgpu.ids > gpurun_out/bench.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
cd $R
find gpurun_out/prof_bench -type f | head
tail -6 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; cat gpurun_out/kernel_bench.log; echo; cat gpurun_out/kernel_bench_bsz8.log | grep -E "mix_gap|residual_store|solve|gap_update_bhw"; tail -1 gpurun_out/bench.log
