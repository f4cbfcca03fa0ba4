#!/bin/bash
# round 6: variants of csrc/conv_w16.hip (built by tools/w16_variants.sh into build/w16v/) - check, stack launch time, 3 bench steps each
# usage: tools/gpu_round6_w16.sh <out tag> <variant> [<variant> ...]
cd "$(dirname "$0")/.."
TAG=$1; shift
O=gpurun_out/r06_w16_$TAG; mkdir -p $O
for v in "$@"; do
  echo "== $v" | tee -a $O/check.txt $O/stack.txt $O/bench.txt
  DEQSCI_HIP_LIB=$PWD/build/w16v/lib_$v.so timeout 300 python tools/w16_check.py check 2>&1 | grep -v amdgpu.ids | tail -4 >> $O/check.txt
  for rep in 1 2; do
    DEQSCI_HIP_LIB=$PWD/build/w16v/lib_$v.so STACK_IMAGES=64 timeout 300 python tools/stack_bench.py 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/stack.txt
  done
  DEQSCI_HIP_LIB=$PWD/build/w16v/lib_$v.so STACK_IMAGES=8 timeout 300 python tools/stack_bench.py 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/stack.txt
  DEQSCI_HIP_LIB=$PWD/build/w16v/lib_$v.so timeout 600 python bench.py --steps 3 --warmup 1 --no-other-kernel --no-other-configs --no-cpu-baseline --no-hbm-stream 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d.get('roofline', {})
print(json.dumps({'value': round(d['value'], 2), 'ms_per_step': round(d['ms_per_step'], 2), 'stack_us': r.get('avg_launch_us'), 'frac_useful': r.get('frac_useful'), 'parity': d.get('parity_spot_check', {}).get('rel_l2'), 'groups': d['config'].get('groups')}))" >> $O/bench.txt
done
cat $O/check.txt $O/stack.txt $O/bench.txt
