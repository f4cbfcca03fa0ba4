#!/bin/bash
# refresh everything that is committed under profiles/
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 > gpurun_out/pytest_gpu.log
timeout 300 python tools/kernel_bench.py 2>&1 | grep "^{" > gpurun_out/kernel_bench.log
timeout 300 python tools/kernel_bench.py --bsz 8 2>&1 | grep "^{" > gpurun_out/kernel_bench_bsz8.log
timeout 300 python tools/kernel_bench.py --bsz 8 --size 512x512x16 2>&1 | grep "^{" > gpurun_out/kernel_bench_512.log
timeout 900 python bench.py --steps 2 --warmup 1 2>&1 | grep "^{" > gpurun_out/bench.log
timeout 900 python bench.py --steps 1 --warmup 1 --denoiser SimpleCNN --no-cpu-baseline 2>&1 | grep "^{" > gpurun_out/bench_simplecnn.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  for B in 64 8; do
    timeout 600 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc/${C}_b$B -o k -- python3 $R/tools/kernel_bench.py --bsz $B --launches 9 --sets 1 > $R/gpurun_out/pmc/${C}_b$B.log 2>&1
  done
done
cd $R
python tools/pmc_summarize.py gpurun_out/pmc 64 8 > gpurun_out/pmc_hbm_traffic.json
tail -2 gpurun_out/pytest_gpu.log; cat gpurun_out/kernel_bench.log | cut -c1-160; cat gpurun_out/bench.log | cut -c1-400; grep -o '"roofline".*' gpurun_out/bench.log | cut -c1-500
