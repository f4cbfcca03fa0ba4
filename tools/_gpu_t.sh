#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stack or e2e or graph_replay or batch or config3 or ragged" 2>&1 | tail -4
run() { timeout 400 python bench.py --warmup 1 --no-parity-check --no-cpu-baseline --no-hbm-stream --no-other-kernel "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],2), d['config']['stack_launches_per_step'])"; }
run --steps 4
run --steps 4 --no-slice-edges
run --steps 4
run --steps 4 --no-slice-edges
run --steps 6 --batch-per-gpu 1
run --steps 6 --batch-per-gpu 1 --no-slice-edges
run --steps 4 --batch-per-gpu 2
run --steps 4 --batch-per-gpu 2 --no-slice-edges
run --steps 1 --batch-per-gpu 32
run --steps 1 --batch-per-gpu 32 --no-slice-edges
