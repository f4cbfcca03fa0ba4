#!/bin/bash
# needs the diagnostic build of the library (`make diag`): the shipped one reads no environment variable
# same box, same process layout: every kernel under each cache policy (0 default, 1 nt loads, 2 nt stores, 3 both)
mkdir -p gpurun_out
for rep in 1 2; do
for P in 0 1 2 3; do
  DEQSCI_HIP_LIB=build/diag/libdeqsci_hip_diag.so DEQSCI_FORCE_POLICY=$P python tools/kernel_bench.py --launches 40 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$P', d['kernel'], d['avg_us'], d.get('GBps'))
" >> gpurun_out/policy_sweep.log
done; done
python - <<'PY'
from collections import defaultdict
t = defaultdict(lambda: defaultdict(list))
for l in open('gpurun_out/policy_sweep.log'):
    p, k, us, gb = l.split()
    t[k][int(p)].append(float(us))
print('%-24s %9s %9s %9s %9s' % ('kernel (avg us)', 'default', 'nt-load', 'nt-store', 'both'))
for k, v in t.items():
    print('%-24s' % k, ' '.join('%9.2f' % min(v[p]) for p in range(4)))
PY
