#!/bin/bash
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
for a in "" "--no-winograd" "--denoiser SimpleCNN" "--denoiser SimpleCNN --no-winograd"; do echo "bench $a: $(timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline $a 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],1), d['final_res'])")"; done
