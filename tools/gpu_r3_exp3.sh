#!/bin/bash
# round 3: (a) bench with the spill-free F(4x4,3x3) kernel; (b) rounding of each conv form along the iteration; (c) config-2 ensembles:
# hybrid kernel schedules, and the bordered system solved in fp32 as torch.solve does
mkdir -p gpurun_out/r03
O=gpurun_out/r03
python bench.py --steps 2 --warmup 1 --conv64 fast --no-cpu-baseline --no-hbm-stream --no-other-kernel > $O/bench_fast_nospill.json 2>/dev/null
timeout 900 python tools/conv_error_real.py > $O/conv_error_real_iter.jsonl 2> /dev/null
cp gpurun_out/conv_error_real.json $O/conv_error_real.json
DEQSCI_ENSEMBLE_HYBRID=10,20,40 DEQSCI_ENSEMBLE_CONFIGS="hybrid,conv64='f22'" DEQSCI_ENSEMBLE_SEEDS=25 DEQSCI_ENSEMBLE_TRAFFIC_ONLY=1 timeout 1500 python tools/config2_ensemble.py > $O/ensemble25_hybrid.txt 2>&1
cp gpurun_out/config2_ensemble.json $O/config2_ensemble25_hybrid.json
DEQSCI_HIP_LIB=build/diag/libdeqsci_hip_diag.so DEQSCI_SOLVE_F32=1 DEQSCI_ENSEMBLE_CONFIGS="conv64='f22'" DEQSCI_ENSEMBLE_SEEDS=25 DEQSCI_ENSEMBLE_TRAFFIC_ONLY=1 timeout 1500 python tools/config2_ensemble.py > $O/ensemble25_solvef32.txt 2>&1
cp gpurun_out/config2_ensemble.json $O/config2_ensemble25_solvef32.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench_fast_nospill.json')); print('bench fast', d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'])
rows=[json.loads(l) for l in open('gpurun_out/r03/conv_error_real_iter.jsonl')]
import collections
by=collections.OrderedDict()
for r in rows: by.setdefault(r['input'],[]).append(r)
for k,v in by.items():
    import statistics
    print(k, 'f22 %.2e f44 %.2e miopen %.2e (median over layers)'%tuple(statistics.median([r[n] for r in v]) for n in ('f22','f44','miopen')))
PY
grep SUMMARY $O/ensemble25_hybrid.txt; grep SUMMARY $O/ensemble25_solvef32.txt
