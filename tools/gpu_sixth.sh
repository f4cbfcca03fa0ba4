#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --denoiser SimpleCNN 2>&1 | grep "^{" > gpurun_out/bench_simplecnn.log
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --denoiser SimpleCNN --no-channels-last 2>&1 | grep "^{" > gpurun_out/bench_simplecnn_nchw.log
timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --size 512x512x16 --batch-per-gpu 2 2>&1 | grep "^{" > gpurun_out/bench_512.log
tail -3 gpurun_out/pytest_gpu.log
for f in bench_simplecnn bench_simplecnn_nchw bench_512; do python -c "
import json; d=json.loads(open('gpurun_out/$f.log').read()); print('$f', round(d['value'],2), round(d['ms_per_step'],1), d['roofline']['avg_launch_us'], round(d['roofline']['frac'],3), d['final_res'])"; done
