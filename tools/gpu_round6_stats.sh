#!/bin/bash
# round 6: in-loop kernel stats of the bench step (one stream), top kernels
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_stats_$1; mkdir -p $O
Q="--no-other-kernel --no-other-configs --no-cpu-baseline --no-hbm-stream --no-parity-check --no-kernel-timing"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --groups ${2:-1} $Q > $O/tr.log 2>&1
python3 $R/tools/trace_window.py $(find $O/tr -name "*kernel_trace.csv" | head -1) 0 | head -12
cp $(find $O/tr -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -rf $O/tr
