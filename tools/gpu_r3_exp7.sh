#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/gputest_c.txt
python bench.py --steps 3 --warmup 1 > $O/bench_c.json 2> $O/bench_c.err
cat $O/gputest_c.txt | tail -12; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench_c.json'))
print(d['value'], d['ms_per_step'], d['config']['conv64_policy'], d['config'].get('conv64_kernel'))
for k in ('roofline',):
    if k in d: print(k, d[k]['kernel'][:40], d[k]['avg_launch_us'], d[k]['frac'], d[k]['share_of_step_time'], d[k]['launches_per_step'])
print(d.get('other_conv64_policies'))
PY
tail -3 $O/bench_c.err
