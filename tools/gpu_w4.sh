#!/bin/bash
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for a in "--denoiser SimpleCNN" "--denoiser SimpleCNN --no-fused-edges" ""; do echo "bench $a: $(timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline $a 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],1), d['final_res'], round(d['roofline']['frac'],3))")"; done
