#!/bin/bash
# Regenerates the profiles/r06_* files of profiles/README.md's round-6 table (run on the GPU box through gpurun).  Stages can be skipped:
# R6_SKIP="tests pmc shapes anderson".  The stamp file needs the profiling build first: bash tools/w16_variants.sh "stamp:-DW16_STAMP"
mkdir -p gpurun_out/r06p gpurun_out/pmc_w16
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06p
cd $R
skip() { [[ " $R6_SKIP " == *" $1 "* ]]; }
if ! skip tests; then ( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3 ) > $O/r06_gputest_summary.txt; fi
timeout 900 python bench.py --steps 5 --warmup 1 2>&1 | grep "^{" > $O/r06_bench_n1.json
if ! skip shapes; then
( timeout 900 python bench.py --steps 6 --warmup 2 --batch-per-gpu 1 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-other-configs --no-parity-check 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 6 --warmup 2 --batch-per-gpu 1 --no-graph --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-other-configs --no-parity-check 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 6 --warmup 2 --batch-per-gpu 1 --stack-kernel s16 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-other-configs --no-parity-check 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 4 --warmup 2 --batch-per-gpu 2 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-other-configs --no-parity-check 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 4 --warmup 2 --batch-per-gpu 2 --stack-kernel s16 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-other-configs --no-parity-check 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 1 --warmup 1 --batch-per-gpu 32 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-other-configs --no-parity-check 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 1 --warmup 1 --batch-per-gpu 32 --stack-kernel s16 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-other-configs --no-parity-check 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 1 --warmup 1 --global-batch 64 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-other-configs --no-parity-check 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 3 --warmup 1 --stack-kernel s16 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-other-configs --no-parity-check 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 3 --warmup 1 --no-stack --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-other-configs --no-parity-check 2>&1 | grep "^{" ) > $O/r06_bench_other_shapes.jsonl
fi
( for k in w16 s16; do STACK_KERNEL=$k timeout 300 python tools/stack_bench.py 2>&1 | grep "^{"; STACK_KERNEL=$k STACK_IMAGES=8 timeout 300 python tools/stack_bench.py 2>&1 | grep "^{"; done ) > $O/r06_stack_bench.jsonl
timeout 300 python tools/w16_check.py both 2>&1 | grep -v amdgpu > $O/r06_w16_check.txt
if [ -f build/w16v/lib_stamp.so ]; then
  for n in 64 32 8; do W16_IMAGES=$n timeout 200 python tools/w16_stamps.py 2>&1 | grep -v amdgpu; done > $O/r06_w16_stamps.txt
fi
( PROBE_IMAGES=32 PROBE_KERNEL=w16stack timeout 200 python tools/power_probe.py 2>&1 | grep "^{" | tail -1
  PROBE_IMAGES=32 PROBE_KERNEL=stack timeout 200 python tools/power_probe.py 2>&1 | grep "^{" | tail -1 ) > $O/r06_power_probe.jsonl
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench6 -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-other-configs --no-parity-check > $R/gpurun_out/prof_bench6.log 2>&1
cp $(find $R/gpurun_out/prof_bench6 -name "*kernel_stats.csv" | head -1) $O/r06_bench_kernel_stats.csv
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bsz1_6 -o b1 -- python3 $R/bench.py --batch-per-gpu 1 --steps 2 --warmup 2 --no-cpu-baseline --no-hbm-stream --no-kernel-timing --no-other-kernel --no-other-configs --no-parity-check > $R/gpurun_out/prof_bsz1_6.log 2>&1
cp $(find $R/gpurun_out/prof_bsz1_6 -name "*kernel_stats.csv" | head -1) $O/r06_bench_bsz1_graph_kernel_stats.csv
cd $R
if ! skip anderson; then
  ( timeout 300 python tools/gram_fused_time.py 2>&1 | grep "^{" ) > $O/r06_gram_fused_time.jsonl
  ( timeout 300 python tools/edge_p32_bench.py 2>&1 | grep "^{" ) > $O/r06_edge_p32_bench.jsonl
  ( for aa in reference float64; do for b in 8 1; do
      timeout 900 python bench.py --steps 4 --warmup 2 --batch-per-gpu $b --anderson-arith $aa --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-other-configs --no-parity-check 2>&1 | grep "^{"
    done; done
    timeout 900 python bench.py --steps 4 --warmup 2 --groups 2 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-other-configs --no-parity-check 2>&1 | grep "^{" ) > $O/r06_bench_anderson_arith.jsonl
fi
cd /tmp
if ! skip pmc; then
  i=0
  for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/pmc_w16/p$i -o k -- python3 $R/tools/stack_bench.py > $R/gpurun_out/pmc_w16/p$i.log 2>&1
  done
  cd $R
  python - <<'PY' > $O/r06_pmc_conv_w16_stack.json
import csv, collections, glob, json
agg = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/pmc_w16/p*/**/k_counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if "conv_w16_kernel" in r["Kernel_Name"]:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
med = {k: sorted(v)[len(v) // 2] for k, v in agg.items()}
n, L, H, W = 32, 13, 128, 128
rd, wr = med.get("FETCH_SIZE", 0) * 1024 * 2.0, med.get("WRITE_SIZE", 0) * 1024 * 1.0      # (units and corrections as tools/pmc_summarize.py calibrates them)
alg = L * (2 * n * H * W * 256 + 196608)
mfma = L * n * H * W * 3 * 6 * 64 * 64 * 2 / 32768
out = {"kernel": "deqsci::w16::conv_w16_kernel<1> (stack launch: 13 layers over a slice of 32 images)", "shape": [n, 64, H, W], "layers": L,
       "launches": len(agg.get("FETCH_SIZE", [])), "counters_median": med, "hbm_read_bytes": int(rd), "hbm_write_bytes": int(wr), "hbm_bytes_per_launch": int(rd + wr),
       "algorithmic_hbm_bytes": alg, "traffic_over_algorithmic": round((rd + wr) / alg, 3) if alg else None,
       "mfma_instructions_per_launch": mfma,
       "mfma_busy_fraction": round(med["SQ_VALU_MFMA_BUSY_CYCLES"] / (med["GRBM_GUI_ACTIVE"] / 8 * 1024), 3) if "SQ_VALU_MFMA_BUSY_CYCLES" in med and "GRBM_GUI_ACTIVE" in med else None,
       "non_mfma_valu_per_mfma": round((med["SQ_INSTS_VALU"] - mfma) / mfma, 2) if "SQ_INSTS_VALU" in med else None,
       "lds_bank_conflict_share": round(med["SQ_LDS_BANK_CONFLICT"] / med["SQ_LDS_IDX_ACTIVE"], 4) if med.get("SQ_LDS_IDX_ACTIVE") else None,
       "note": "FETCH_SIZE / WRITE_SIZE count at the L2's fabric side, in front of the Infinity Cache: what that cache serves of the slice's activations is not "
               "subtracted; mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)"}
print(json.dumps(out, indent=1))
PY
fi
cd $R
ls -la $O; cat $O/r06_gputest_summary.txt 2>/dev/null; python - <<PY
import json
d=json.loads(open('$O/r06_bench_n1.json').read().strip().splitlines()[-1]); r=d['roofline']
print('bench', round(d['value'],1), 'frames/s', r['kernel'][:40], 'frac', round(r['frac'],3), 'useful', round(r['frac_useful'],3), 'us', round(r['avg_launch_us'],1))
print('other_configs', {k: round(v['value'],1) for k,v in d.get('other_configs',{}).items()})
print('other_policies', {k: round(v['value'],1) for k,v in d.get('other_conv64_policies',{}).items()})
try:
    for ln in open('$O/r06_bench_other_shapes.jsonl'):
        x=json.loads(ln); print(round(x['value'],1), x['config']['workload'][:60], x['config'].get('launch_mode','')[:12], x['config'].get('stack_kernel'), x['config'].get('stack_launches_per_step'))
except Exception as e: print(e)
PY
cat $O/r06_stack_bench.jsonl; cat $O/r06_power_probe.jsonl | cut -c1-300
