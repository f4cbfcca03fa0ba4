#!/bin/bash
run() { echo "$1: $(eval "$2 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline $3" 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],1), d['final_res'])")"; }
for rep in 1 2; do
run "defer, no find" "" ""
run "no defer, no find" "DEQSCI_NO_DEFER=1" ""
run "defer, find" "" "--miopen-find"
run "no defer, find" "DEQSCI_NO_DEFER=1" "--miopen-find"
done
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_find -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --iters 30 --miopen-find > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_find/t_kernel_stats.csv')))
for r in rows[:8]: print(r['Name'][:100], r['Calls'], round(float(r['AverageNs'])/1e3,2), r['Percentage'])
PY
