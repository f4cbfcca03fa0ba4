#!/bin/bash
# round 4, pass B: (1) the 4-row wave geometry of conv_s16 against the shipped one, same box; (2) the scatter of the config-2 pooled mean
# between equivalent arithmetics; (3) the re-stated scale tests
mkdir -p gpurun_out/r04b
O=gpurun_out/r04b
bash tools/s16_variants.sh "base:" "rows4:-DS16_ROWS4=1" > $O/variants_build.log 2>&1
cat $O/variants_build.log | tail -3
for v in base rows4; do
  DEQSCI_HIP_LIB=build/s16v/lib_$v.so timeout 600 python tools/s16_check.py both 2>&1 | grep -v amdgpu > $O/s16_check_$v.txt
  tail -4 $O/s16_check_$v.txt | cut -c1-300
done
for v in base rows4 base rows4; do
  DEQSCI_HIP_LIB=build/s16v/lib_$v.so timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-parity-check 2>&1 | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), 'fps', round(d['roofline']['avg_launch_us'],2), 'us', round(d['roofline']['frac'],4))" | tee -a $O/bench_ab.txt
done
DEQSCI_HIP_LIB=build/s16v/lib_rows4.so timeout 900 python -m pytest tests -q -m gpu -x -k "split16 or s16 or data_scale or engine_split16" 2>&1 | tail -3 | tee $O/rows4_tests.log
timeout 1200 python -m pytest tests -q -m gpu --tb=short -s -k "data_scale or scaled_measurements or rounding_along or overflows" 2>&1 | grep -E "^\{|^[0-9]+ \{|^E  |FAILED|passed|failed|vs reference" | cut -c1-400 | tee $O/scale_tests.log
timeout 1500 python tools/config2_fp64_denoiser.py seeds=50 out=config2_family.json variants=fast+1,fast+2,fast+3,fast+4,fast+5,fast+6,fast+7,fast+8,fast+10,fast+12,fast+16,fast+20 > $O/family.log 2>&1
grep -E "FAMILY|SUMMARY" $O/family.log | cut -c1-1500; cp gpurun_out/config2_family.json $O/
