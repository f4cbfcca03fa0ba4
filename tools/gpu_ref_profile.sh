#!/bin/bash
# per-kernel time of anderson_arith="reference" inside the bench step (one and eight measurements per call) + the step rates
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05a; mkdir -p $O
for b in 1 8; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ref_b$b -o t -- python3 $R/bench.py --batch-per-gpu $b --steps 2 --warmup 1 --anderson-arith reference --no-cpu-baseline --no-hbm-stream --no-kernel-timing --no-other-kernel --no-other-configs --no-parity-check > $O/prof_b$b.log 2>&1
  f=$(find $R/gpurun_out/prof_ref_b$b -name "*kernel_stats.csv" | head -1); grep "gram_\|anderson_solve\|residual_store" $f | cut -d, -f1-4 | sed 's/(.*)"/"/' > $O/r05_ref_kernels_b$b.csv; cat $O/r05_ref_kernels_b$b.csv
done
cd $R
for aa in reference float64; do for b in 8 1; do
  timeout 600 python bench.py --steps 4 --warmup 2 --batch-per-gpu $b --anderson-arith $aa --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-other-configs --no-parity-check 2>&1 | grep "^{"
done; done > $O/r05_bench_anderson_arith.jsonl
python3 - <<'PY'
import json
for l in open('gpurun_out/r05a/r05_bench_anderson_arith.jsonl'):
    d=json.loads(l); print(d['config'].get('anderson_arith'), d['config'].get('global_batch'), round(d['value'],1), round(d['ms_per_step'],1))
PY
