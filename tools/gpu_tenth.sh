#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep "^{" | cut -c1-120
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fused-edges 2>&1 | grep "^{" | cut -c1-120
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tail -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --iters 30 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_tail/t_kernel_stats.csv')))
for r in rows[:12]: print(r['Name'][:90], r['Calls'], round(float(r['AverageNs'])/1e3,2), r['Percentage'])
PY
