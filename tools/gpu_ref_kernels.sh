#!/bin/bash
# per-kernel time of anderson_arith="reference" inside the bench step (one and eight measurements per call): rocprofv3 kernel stats of its kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for b in 1 8; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ref_b$b -o t -- python3 $R/bench.py --batch-per-gpu $b --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-stream --no-kernel-timing --no-other-kernel --no-other-configs --no-parity-check > $R/gpurun_out/prof_ref_b$b.log 2>&1
  python3 - $(find $R/gpurun_out/prof_ref_b$b -name "*kernel_stats.csv" | head -1) $(find $R/gpurun_out/prof_ref_b$b -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('gram_chain', 'gram_round_kernel<5', 'anderson_solve', 'residual_store_kernel<5')):
        print(n[:50].ljust(50), r['Calls'], round(float(r['AverageNs']) / 1e3, 1), r['MinNs'], r['MaxNs'])
d = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(sys.argv[2])) if 'gram_chain_apply' in r['Kernel_Name'])
print('gram_chain_apply percentiles 5/50/90/99 us:', [round(d[int(len(d) * q)], 1) for q in (0.05, 0.5, 0.9, 0.99)])
PY
done
