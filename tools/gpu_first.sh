#!/bin/bash
# first GPU trip: tests, smoke, kernel microbench, short bench
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|gfx9" | head -4 > gpurun_out/rocminfo.txt 2>&1
nproc >> gpurun_out/rocminfo.txt; lscpu | grep "Model name" >> gpurun_out/rocminfo.txt
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py::test_end_to_end_vs_reference_rec 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
timeout 300 python tools/kernel_bench.py > gpurun_out/kernel_bench.log 2>&1
timeout 600 python bench.py --steps 1 --warmup 1 > gpurun_out/bench.log 2>&1
tail -5 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; cat gpurun_out/kernel_bench.log; tail -2 gpurun_out/bench.log
