#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/gputest_d.txt
python bench.py --steps 3 --warmup 1 > $O/bench_d.json 2> $O/bench_d.err
python bench.py --steps 2 --warmup 1 --denoiser SimpleCNN --no-cpu-baseline --no-hbm-stream > $O/bench_d_simplecnn.json 2>/dev/null
python bench.py --steps 3 --warmup 2 --batch-per-gpu 1 --no-cpu-baseline --no-hbm-stream --no-other-kernel > $O/bench_d_bsz1.json 2>/dev/null
cat $O/gputest_d.txt | tail -12; python - <<'PY'
import json
for f in ('bench_d','bench_d_simplecnn','bench_d_bsz1'):
    d=json.load(open(f'gpurun_out/r03/{f}.json'))
    print(f, d['value'], d['ms_per_step'], d['config']['conv64_policy'], d['config'].get('conv64_kernel'))
    for k in ('roofline',):
        if k in d and 'avg_launch_us' in d[k]: print(' ', k, d[k]['kernel'][:40], d[k]['avg_launch_us'], d[k]['frac'], d[k].get('share_of_step_time'))
    print(' ', d.get('other_conv64_policies'))
PY
tail -3 $O/bench_d.err
