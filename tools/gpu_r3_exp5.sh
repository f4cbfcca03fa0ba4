#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gputest_b.txt
python bench.py --steps 3 --warmup 1 > $O/bench_b.json 2> $O/bench_b.err
python tools/w44_stamps.py > $O/w44_stamps_b64.txt 2>&1
cat $O/gputest_b.txt; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench_b.json'))
print(d['value'], d['ms_per_step'], d['config']['conv64_policy'], d['config'].get('conv64_kernel'))
for k in ('roofline','roofline_other_form'):
    if k in d: print(k, d[k]['kernel'][:40], d[k]['avg_launch_us'], d[k]['frac'], d[k]['share_of_step_time'], d[k]['launches_per_step'])
print(d.get('other_conv64_policy'))
PY
tail -3 $O/bench_b.err; cat $O/w44_stamps_b64.txt
