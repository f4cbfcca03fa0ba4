#!/bin/bash
# Regenerates everything committed under profiles/r03_* (run on the GPU box through gpurun).  Stages can be skipped: R3_SKIP="parity ensembles"
mkdir -p gpurun_out/pmc gpurun_out/r03p
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03p
cd $R
skip() { [[ " $R3_SKIP " == *" $1 "* ]]; }
timeout 900 python bench.py --steps 3 --warmup 1 2>&1 | grep "^{" > $O/r03_bench_n1.json
timeout 900 python bench.py --steps 2 --warmup 1 --denoiser SimpleCNN --no-cpu-baseline --no-hbm-stream 2>&1 | grep "^{" > $O/r03_bench_n1_simplecnn.json
( timeout 900 python bench.py --steps 1 --warmup 1 --size 512x512x16 --no-cpu-baseline --no-hbm-stream --no-other-kernel 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 3 --warmup 2 --batch-per-gpu 1 --no-cpu-baseline --no-hbm-stream --no-other-kernel 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 3 --warmup 2 --batch-per-gpu 1 --no-graph --no-cpu-baseline --no-hbm-stream --no-other-kernel 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 1 --warmup 1 --batch-per-gpu 32 --no-cpu-baseline --no-hbm-stream --no-other-kernel 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 1 --warmup 1 --global-batch 64 --no-cpu-baseline --no-hbm-stream --no-other-kernel 2>&1 | grep "^{" ) > $O/r03_bench_other_shapes.jsonl
timeout 300 python tools/kernel_bench.py 2>&1 | grep "^{" > $O/r03_kernel_bench_bsz64.jsonl
timeout 300 python tools/conv_bench.py 2>&1 | grep "^{" > $O/r03_conv_bench.jsonl
timeout 300 python tools/s16_check.py both 2>&1 | grep -v amdgpu > $O/r03_s16_check.txt
timeout 600 python tools/conv_error_real.py 2>/dev/null > /dev/null; cp gpurun_out/conv_error_real.json $O/r03_conv_error_real.json
timeout 600 python tools/fcall_error_along_loop.py 2>/dev/null > /dev/null; cp gpurun_out/fcall_error_along_loop.json $O/r03_fcall_error_along_loop.json
./build/ub/mfma_f16_numerics > $O/r03_mfma_f16_numerics.txt 2>&1
for k in s16 f44 f22; do PROBE_KERNEL=$k timeout 200 python tools/power_probe.py 2>&1 | grep "^{" | tail -1; done > $O/r03_power_probe_kernels.jsonl
if ! skip ensembles; then
  DEQSCI_ENSEMBLE_HYBRID=40 DEQSCI_ENSEMBLE_SEEDS=25 DEQSCI_ENSEMBLE_TRAFFIC_ONLY=1 timeout 2400 python tools/config2_ensemble.py > $O/r03_config2_ensembles.log 2>&1
  cp gpurun_out/config2_ensemble.json $O/r03_config2_ensembles.json
fi
if ! skip parity; then
  timeout 2400 python tools/parity_report.py 3 > $O/parity_report.log 2>&1
  cp gpurun_out/parity_report.md $O/r03_parity_report.md
fi
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-hbm-stream --no-other-kernel > $R/gpurun_out/prof_bench.log 2>&1
cp $(find $R/gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1) $O/r03_bench_kernel_stats.csv
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bsz1 -o b1 -- python3 $R/bench.py --batch-per-gpu 1 --steps 2 --warmup 2 --no-cpu-baseline --no-hbm-stream --no-kernel-timing --no-other-kernel > $R/gpurun_out/prof_bsz1.log 2>&1
cp $(find $R/gpurun_out/prof_bsz1 -name "*kernel_stats.csv" | head -1) $O/r03_bench_bsz1_graph_kernel_stats.csv
cd $R
bash tools/pmc_winograd.sh > /dev/null 2>&1
cp gpurun_out/pmc_winograd.json $O/r03_pmc_winograd.json
cp gpurun_out/pmc_winograd44.json $O/r03_pmc_winograd44.json
cp gpurun_out/pmc_conv_s16.json $O/r03_pmc_conv_s16.json
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  for B in 64 8; do
    timeout 600 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc/${C}_b$B -o k -- python3 $R/tools/kernel_bench.py --bsz $B --launches 9 --sets 1 > $R/gpurun_out/pmc/${C}_b$B.log 2>&1
  done
done
cd $R
python tools/pmc_summarize.py gpurun_out/pmc 64 8 > $O/r03_pmc_hbm_traffic.json
ls -la $O; head -c 900 $O/r03_bench_n1.json; echo; tail -5 $O/parity_report.log 2>/dev/null; python -c "
import json
for f in ('r03_pmc_winograd.json', 'r03_pmc_winograd44.json', 'r03_pmc_conv_s16.json'):
    d=json.load(open('$O/' + f)); print(f, {k: d.get(k) for k in ('mfma_busy_fraction','non_mfma_valu_per_mfma','traffic_over_algorithmic','lds_bank_conflict_share','hbm_bytes_per_launch')})"
grep SUMMARY $O/r03_config2_ensembles.log | cut -c1-600
