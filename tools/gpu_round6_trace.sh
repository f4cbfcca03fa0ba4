#!/bin/bash
# round 6: kernel traces of the bench step, grouped and on one stream; tools/trace_window.py reads them
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_trace; mkdir -p $O
Q="--no-other-kernel --no-other-configs --no-cpu-baseline --no-hbm-stream --no-parity-check --no-kernel-timing"
cd /tmp
for g in auto 1; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/g$g -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --groups $g $Q > $O/g$g.log 2>&1
  f=$(find $O/g$g -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/trace_window.py $f 2 > $O/window_g$g.txt
  cp $(find $O/g$g -name "*kernel_stats.csv" | head -1) $O/kernel_stats_g$g.csv
  rm -rf $O/g$g
done
cat $O/window_gauto.txt; cat $O/window_g1.txt
