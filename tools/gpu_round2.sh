#!/bin/bash
# Regenerates everything committed under profiles/r02_* (run on the GPU box through gpurun).
mkdir -p gpurun_out/pmc gpurun_out/r02
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
cd $R
timeout 900 python bench.py --steps 2 --warmup 1 2>&1 | grep "^{" > $O/r02_bench_n1.json
timeout 900 python bench.py --steps 1 --warmup 1 --denoiser SimpleCNN --no-cpu-baseline --no-hbm-stream 2>&1 | grep "^{" > $O/r02_bench_n1_simplecnn.json
( timeout 900 python bench.py --steps 1 --warmup 1 --size 512x512x16 --no-cpu-baseline --no-hbm-stream 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 3 --warmup 2 --batch-per-gpu 1 --no-cpu-baseline --no-hbm-stream 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 3 --warmup 2 --batch-per-gpu 1 --no-graph --no-cpu-baseline --no-hbm-stream 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 1 --warmup 1 --batch-per-gpu 32 --no-cpu-baseline --no-hbm-stream 2>&1 | grep "^{" ) > $O/r02_bench_other_shapes.jsonl
timeout 300 python tools/kernel_bench.py 2>&1 | grep "^{" > $O/r02_kernel_bench_bsz64.jsonl
timeout 300 python tools/conv_bench.py 2>&1 | grep "^{" > $O/r02_conv_bench.jsonl
timeout 2400 python tools/parity_report.py 2 > $O/parity_report.log 2>&1
cp gpurun_out/parity_report.md $O/r02_parity_report.md
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-hbm-stream > $R/gpurun_out/prof_bench.log 2>&1
cp $(find $R/gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1) $O/r02_bench_kernel_stats.csv
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bsz1 -o b1 -- python3 $R/bench.py --batch-per-gpu 1 --steps 2 --warmup 2 --no-cpu-baseline --no-hbm-stream --no-kernel-timing > $R/gpurun_out/prof_bsz1.log 2>&1
cp $(find $R/gpurun_out/prof_bsz1 -name "*kernel_stats.csv" | head -1) $O/r02_bench_bsz1_graph_kernel_stats.csv
cd $R
bash tools/pmc_winograd.sh > /dev/null 2>&1
cp gpurun_out/pmc_winograd.json $O/r02_pmc_winograd.json
cp gpurun_out/pmc_winograd44.json $O/r02_pmc_winograd44.json
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  for B in 64 8; do
    timeout 600 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc/${C}_b$B -o k -- python3 $R/tools/kernel_bench.py --bsz $B --launches 9 --sets 1 > $R/gpurun_out/pmc/${C}_b$B.log 2>&1
  done
done
cd $R
python tools/pmc_summarize.py gpurun_out/pmc 64 8 > $O/r02_pmc_hbm_traffic.json
ls -la $O; head -c 600 $O/r02_bench_n1.json; echo; cat $O/parity_report.log | tail -9; python -c "
import json; 
for f in ('r02_pmc_winograd.json', 'r02_pmc_winograd44.json'):
    d=json.load(open('$O/' + f)); print(f, {k: d[k] for k in ('mfma_busy_fraction','non_mfma_valu_per_mfma','traffic_over_algorithmic')})"
