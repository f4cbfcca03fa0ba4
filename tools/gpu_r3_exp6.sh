#!/bin/bash
# round 3: split-fp16 direct convolution: rounding on the network's own data, ensembles, bench, counters
mkdir -p gpurun_out/r03 gpurun_out/pmc_s16
O=gpurun_out/r03
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "config2 or kernel_choice or ffdnet" 2>&1 | tail -4 > $O/s16_regress.txt
timeout 900 python tools/conv_error_real.py > $O/conv_error_real_s16.jsonl 2>/dev/null
timeout 900 python tools/fcall_error_along_loop.py > $O/fcall_error_along_loop_s16.jsonl 2>/dev/null
python bench.py --steps 2 --warmup 1 --conv64 s16 --no-cpu-baseline --no-hbm-stream --no-other-kernel > $O/bench_s16.json 2>$O/bench_s16.err
DEQSCI_ENSEMBLE_HYBRID_S16=40 DEQSCI_ENSEMBLE_CONFIGS="s16" DEQSCI_ENSEMBLE_SEEDS=25 DEQSCI_ENSEMBLE_TRAFFIC_ONLY=1 timeout 1500 python tools/config2_ensemble.py > $O/ensemble25_s16.txt 2>&1
cp gpurun_out/config2_ensemble.json $O/config2_ensemble25_s16.json
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_s16/p1 -o k -- python3 $R/tools/s16_check.py time > $R/gpurun_out/pmc_s16/p1.log 2>&1
cd $R
python - <<'PY'
import csv, collections, glob, json, statistics
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_s16/p1/**/k_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_s16_kernelILi0" in r["Kernel_Name"] and r["Grid_Size"] == str(256*512):
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
med={k: sorted(x)[len(x)//2] for k,x in agg.items()}
print('s16 counters (64x128x128):', json.dumps(med))
if med: print('mfma busy fraction', med['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*med['GRBM_GUI_ACTIVE']/8) if 'GRBM_GUI_ACTIVE' in med else None)
json.dump(med, open('gpurun_out/r03/s16_counters.json','w'))
rows=[json.loads(l) for l in open('gpurun_out/r03/conv_error_real_s16.jsonl')]
by=collections.OrderedDict()
for r in rows: by.setdefault(r['input'],[]).append(r)
for k,v in by.items():
    print(k, ' '.join('%s %.2e'%(n, statistics.median([r[n] for r in v])) for n in ('f22','f44','s16','miopen')))
for l in open('gpurun_out/r03/fcall_error_along_loop_s16.jsonl'):
    d=json.loads(l); print('call',d['call'],' '.join('%s %.2e'%(n,d[n]) for n in ('f22','f44','s16','miopen')))
d=json.load(open('gpurun_out/r03/bench_s16.json')); print('bench s16', d['value'], d['ms_per_step'])
PY
cat $O/s16_regress.txt; tail -2 $O/bench_s16.err; grep SUMMARY $O/ensemble25_s16.txt | cut -c1-1500
