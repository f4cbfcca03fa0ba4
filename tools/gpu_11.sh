#!/bin/bash
timeout 900 python -m pytest tests -m gpu -q -x -k "tail or head or variants or end_to_end" 2>&1 | tail -3
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep "^{" | cut -c1-120
python tools/find_mode_probe.py 2>&1 | grep benchmark
for v in 0 1; do MIOPEN_FIND_MODE=1 python tools/find_mode_probe.py 2>&1 | grep "benchmark True"; done
