#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/pytest_gpu.log
timeout 1500 python tools/parity_report.py 2>&1 | grep -v amdgpu.ids > gpurun_out/parity_report.log
for b in 1 2 4 16; do timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch-per-gpu $b 2>&1 | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('bsz',$b,'fps',round(d['value'],2),'ms/step',round(d['ms_per_step'],1),'mixgap us',round(d['roofline']['avg_launch_us'],2), 'frac', round(d['roofline']['frac'],3))"; done > gpurun_out/bench_bsz_sweep.log 2>&1
tail -3 gpurun_out/pytest_gpu.log; cat gpurun_out/parity_report.log; cat gpurun_out/bench_bsz_sweep.log
