#!/bin/bash
# HBM traffic per launch from PMC counters (separate passes for FETCH_SIZE / WRITE_SIZE, no trace domains)
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  for B in 64 8; do
    timeout 600 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc/${C}_b$B -o k -- python3 $R/tools/kernel_bench.py --bsz $B --launches 10 > $R/gpurun_out/pmc/${C}_b$B.log 2>&1
  done
done
cd $R
find gpurun_out/pmc -name "*.csv" | head -20
head -3 $(find gpurun_out/pmc -name "*counter_collection.csv" | head -1)
