#!/bin/bash
mkdir -p gpurun_out/r04g
O=gpurun_out/r04g
timeout 900 python -m pytest tests -q -m gpu --tb=short -s -k "config2 or conv_layout_and_kernel_choice or ranges_are_measured" 2>&1 | grep -E "^E  |FAILED|passed|failed|six chaotic|RMS|harness average|build mean|reference arithmetic|Error" | cut -c1-330 | tee $O/config2_tests.log
timeout 900 python tools/config2_fp64_denoiser.py seeds=100 out=config2_refarith.json variants=refarith > $O/refarith.log 2>&1; grep -E "SUMMARY|m[0-5]:" $O/refarith.log | cut -c1-400; cp gpurun_out/config2_refarith.json $O/
timeout 1500 python -m pytest tests -q -m gpu --tb=line 2>&1 | tail -8 | tee $O/gpu_tests.log
timeout 600 python bench.py --steps 5 --warmup 1 2>&1 | grep "^{" > $O/bench_n1.json; python -c "
import json; d=json.load(open('$O/bench_n1.json')); print(round(d['value'],2), 'fps', {k: d['roofline'].get(k) for k in ('avg_launch_us','frac','frac_useful','share_of_step_time')}, d.get('parity_spot_check',{}).get('rel_l2'), {k: round(v['value'],1) for k,v in d.get('other_conv64_policies',{}).items()}, d['cpu_baseline']['value'])"
