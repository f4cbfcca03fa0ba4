#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_scnn -o t -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --iters 20 --denoiser SimpleCNN > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ffd -o t -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --iters 20 > /dev/null 2>&1
cd $R; python - <<'PY'
import csv
for d in ('prof_scnn','prof_ffd'):
    rows=list(csv.DictReader(open(f'gpurun_out/{d}/t_kernel_stats.csv')))
    print(d)
    for r in rows[:9]: print('  ', r['Name'][:86], r['Calls'], round(float(r['AverageNs'])/1e3,2), r['Percentage'])
PY
