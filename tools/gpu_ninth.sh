#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "tail or variants or picard_180 or end_to_end" 2>&1 | tail -4
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep "^{" | cut -c1-120
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fused-edges 2>&1 | grep "^{" | cut -c1-120
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tail -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --iters 30 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; head -8 gpurun_out/prof_tail/t_kernel_stats.csv | cut -c1-110,300-420 | awk -F, '{print $1, $2, $(NF-4), $(NF-3)}' | cut -c1-200
