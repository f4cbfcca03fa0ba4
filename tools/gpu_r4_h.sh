#!/bin/bash
# round 4, pass H: after the head kernel's TRACK template - tests touched, bench line, kernel stats (refreshes r04_bench_n1.json, r04_bench_kernel_stats.csv)
mkdir -p gpurun_out/r04h
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04h
cd $R
timeout 900 python -m pytest tests -q -m gpu -x -k "ffdnet_head_kernel or ranges_are_measured or config3 or plain_edge or data_scale or smoke" 2>&1 | tail -3 | tee $O/tests.log
timeout 900 python bench.py --steps 5 --warmup 1 2>&1 | grep "^{" > $O/r04_bench_n1.json
timeout 900 python bench.py --steps 6 --warmup 2 --batch-per-gpu 1 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-parity-check 2>&1 | grep "^{" > $O/bench_bsz1.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-parity-check > $R/gpurun_out/prof_bench.log 2>&1
cp $(find $R/gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1) $O/r04_bench_kernel_stats.csv
cd $R
python - <<'PY'
import json, csv
d=json.load(open('gpurun_out/r04h/r04_bench_n1.json')); print(round(d['value'],2), 'fps', round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],4))
print('bsz1', round(json.load(open('gpurun_out/r04h/bench_bsz1.json'))['value'],2))
rows=list(csv.DictReader(open('gpurun_out/r04h/r04_bench_kernel_stats.csv'))); tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:8]: print('  ', r['Name'][:60].ljust(60), r['Calls'].rjust(6), '%.1f us'%(float(r['AverageNs'])/1e3), '%.2f%%'%(100*float(r['TotalDurationNs'])/tot))
PY
timeout 300 python tools/s16_fuzz.py 2>&1 | tail -3 | tee $O/s16_fuzz.txt
