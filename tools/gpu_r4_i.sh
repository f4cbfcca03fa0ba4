#!/bin/bash
mkdir -p gpurun_out/r04i
O=gpurun_out/r04i
timeout 900 python -m pytest tests -q -m gpu --tb=short -x -k "config3 or data_scale or ffdnet_tail_kernel or ffdnet_head_kernel or plain_edge or ranges_are_measured or split16" 2>&1 | tail -15 | cut -c1-300 | tee $O/tests_a.log
timeout 1500 python -m pytest tests -q -m gpu --tb=line 2>&1 | tail -8 | cut -c1-300 | tee $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/smoke.log
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-hbm-stream --no-other-kernel 2>&1 | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), 'fps', round(d['roofline']['avg_launch_us'],2), round(d['roofline']['frac'],4), d['parity_spot_check']['rel_l2'])" | tee $O/bench.txt
timeout 600 python bench.py --steps 6 --warmup 2 --batch-per-gpu 1 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-parity-check 2>&1 | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bsz1', round(d['value'],2), 'fps')" | tee -a $O/bench.txt
