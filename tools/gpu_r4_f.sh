#!/bin/bash
# round 4, pass F: the stack launch (13 layers, grid barriers) - tests, then bench with and without it at bsz 8 and bsz 1; Anderson-arithmetic controls
mkdir -p gpurun_out/r04f
O=gpurun_out/r04f
timeout 600 python -m pytest tests -q -m gpu --tb=short -x -k "stack or ranges_are_measured or split16" 2>&1 | tail -8 | tee $O/stack_tests.log
timeout 1500 python -m pytest tests -q -m gpu --tb=line 2>&1 | tail -15 | tee $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8 | tee $O/smoke.log
timeout 600 python bench.py --steps 4 --warmup 1 2>&1 | grep "^{" > $O/bench_n1.json
python - <<'PY' | tee -a $O/bench_summary.txt
import json
d=json.load(open('gpurun_out/r04f/bench_n1.json'))
print('bsz8 stack:', round(d['value'],2), 'fps', {k: d['roofline'].get(k) for k in ('avg_launch_us','layers_per_launch','avg_layer_us','frac','frac_useful','share_of_step_time')})
print('parity', d.get('parity_spot_check'), 'other', {k: round(v['value'],1) for k,v in d.get('other_conv64_policies',{}).items()})
PY
for extra in "" "--batch-per-gpu 1 --steps 6 --warmup 2"; do
  timeout 600 python bench.py --steps 4 --warmup 1 $extra --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-parity-check 2>&1 | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('stack on  [$extra]', round(d['value'],2), 'fps')" | tee -a $O/bench_summary.txt
done
timeout 900 python tools/config2_anderson_arith.py seeds=50 denoiser=miopen variants=g64s64,g64s32 out=config2_anderson_arith_controls.json > $O/anderson_controls.log 2>&1
grep -E "SUMMARY|m[0-5]:" $O/anderson_controls.log | cut -c1-600; cp gpurun_out/config2_anderson_arith_controls.json $O/ 2>/dev/null
