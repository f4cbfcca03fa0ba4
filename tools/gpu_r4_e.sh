#!/bin/bash
mkdir -p gpurun_out/r04e
O=gpurun_out/r04e
timeout 2400 python tools/config2_anderson_arith.py seeds=50 denoiser=miopen variants=g32s32,g64s32,g64s64 > $O/anderson_arith_miopen.log 2>&1
grep -E "SUMMARY|Error|error" $O/anderson_arith_miopen.log | cut -c1-1200; tail -3 $O/anderson_arith_miopen.log | cut -c1-300
cp gpurun_out/config2_anderson_arith_miopen.json $O/ 2>/dev/null
bash tools/gpu_r4_d.sh
