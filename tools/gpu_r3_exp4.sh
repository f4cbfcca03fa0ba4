#!/bin/bash
# round 3: (a) f-call rounding along the real loop; (b) F(4x4,3x3) kernel A/B: round-2 source, spill-free, spill-free + b64 patch reads:
# time (NHWC and blk32) and LDS counters
mkdir -p gpurun_out/r03 gpurun_out/pmc_w44ab
O=gpurun_out/r03
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python tools/fcall_error_along_loop.py > $O/fcall_error_along_loop.jsonl 2>/dev/null
for rep in 1 2; do
for v in r02 nospill b64reads; do
  W44_LIB=build/w44v/lib_$v.so W44_SHAPES=2 python tools/w44_check.py time 2>/dev/null | grep "^{" >> $O/w44_ab_time.jsonl
  W44_LIB=build/w44v/lib_$v.so python tools/w44_blk_time.py 2>/dev/null | grep "^{" >> $O/w44_ab_time.jsonl
done; done
W44_LIB=build/w44v/lib_b64reads.so W44_SHAPES=8 python tools/w44_check.py check 2>/dev/null | tail -3 > $O/w44_b64reads_check.txt
cd /tmp
for v in nospill b64reads; do
  export DEQSCI_HIP_LIB=$R/build/w44v/lib_$v.so
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_w44ab/$v -o k -- python3 $R/tools/conv_bench.py --shape 64 128 128 > $R/gpurun_out/pmc_w44ab/$v.log 2>&1
done
unset DEQSCI_HIP_LIB
cd $R
python - <<'PY'
import csv, collections, glob, json
out={}
for v in ("nospill","b64reads"):
    agg=collections.defaultdict(list)
    for f in glob.glob(f'gpurun_out/pmc_w44ab/{v}/**/k_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if "winograd44_conv64_kernel" in r["Kernel_Name"]:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    med={k: sorted(x)[len(x)//2] for k,x in agg.items()}
    if med:
        med["lds_conflict_share"]=round(med["SQ_LDS_BANK_CONFLICT"]/med["SQ_LDS_IDX_ACTIVE"],3)
    out[v]=med
json.dump(out, open('gpurun_out/r03/w44_ab_lds_counters.json','w'), indent=1)
print(json.dumps(out))
PY
cat $O/w44_ab_time.jsonl; cat $O/w44_b64reads_check.txt; cat $O/fcall_error_along_loop.jsonl | cut -c1-220
