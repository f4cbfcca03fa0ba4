#!/bin/bash
mkdir -p gpurun_out/r04c
O=gpurun_out/r04c
timeout 900 python -m pytest tests -q -m gpu --tb=short -x -k "ranges_are_measured or cli_end_to_end or data_scale" 2>&1 | tail -5 | tee $O/tests.log
timeout 1500 python tools/config2_fp64_denoiser.py seeds=50 out=config2_family.json variants=fast+1,fast+2,fast+3,fast+4,fast+5,fast+6,fast+7,fast+8,fast+10,fast+12,fast+16,fast+20 > $O/family.log 2>&1
grep -E "FAMILY|SUMMARY" $O/family.log | cut -c1-1500; cp gpurun_out/config2_family.json $O/
for v in base rows4; do for n in 64 8; do DEQSCI_HIP_LIB=build/s16v/lib_$v.so PROBE_IMAGES=$n timeout 300 python tools/power_probe.py 2>&1 | grep "^{" | tail -1; done; done | tee $O/power_probe.jsonl
