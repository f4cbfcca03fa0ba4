#!/bin/bash
# SQ counters for the Winograd conv kernel (one pass, 8 SQ slots; no trace domains)
mkdir -p gpurun_out/pmc_wg
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_wg/p1 -o k -- python3 $R/tools/conv_bench.py --shape 64 128 128 > $R/gpurun_out/pmc_wg/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmc_wg/p2 -o k -- python3 $R/tools/conv_bench.py --shape 64 128 128 > $R/gpurun_out/pmc_wg/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_wg/p3 -o k -- python3 $R/tools/conv_bench.py --shape 64 128 128 > $R/gpurun_out/pmc_wg/p3.log 2>&1
cd $R
python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/pmc_wg/p*/k_counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "winograd" in r["Kernel_Name"]:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(f.split('/')[2], k, sorted(v)[len(v)//2], len(v))
PY
tail -2 gpurun_out/pmc_wg/p3.log | cut -c1-200
