#!/bin/bash
mkdir -p gpurun_out/r03 gpurun_out/pmc_s16
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_s16/p1 -o k -- python3 $R/tools/s16_check.py time > $R/gpurun_out/pmc_s16/p1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_s16/p2 -o k -- python3 $R/tools/s16_check.py time > $R/gpurun_out/pmc_s16/p2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmc_s16/p3 -o k -- python3 $R/tools/s16_check.py time > $R/gpurun_out/pmc_s16/p3.log 2>&1
cd $R
python - <<'PY'
import csv, collections, glob, json
agg=collections.defaultdict(list)
names=set()
for f in glob.glob('gpurun_out/pmc_s16/p[12]/**/k_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        names.add(r["Kernel_Name"][:60])
        if "conv_s16_kernel<0>" in r["Kernel_Name"] or "conv_s16_kernelILi0" in r["Kernel_Name"]:
            agg[(r['Counter_Name'])].append(float(r['Counter_Value']))
print(sorted(names)[:12])
med={k: max(x) for k,x in agg.items()}      # the 64x128x128 launches are the largest
print(json.dumps(med))
json.dump(med, open('gpurun_out/r03/s16_counters.json','w'))
for f in glob.glob('gpurun_out/pmc_s16/p3/**/k_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MaxNs'], r['MinNs'])
PY
