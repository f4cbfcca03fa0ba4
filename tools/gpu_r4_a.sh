#!/bin/bash
mkdir -p gpurun_out/r04a
O=gpurun_out/r04a
timeout 600 python -m pytest tests -q -m gpu --tb=short -s -k "data_scale or rounding_along" 2>&1 | grep -E "^\{|^[0-9]+ \{|^E  |FAILED|passed|failed" > $O/fail_tests.log
cat $O/fail_tests.log | cut -c1-400
timeout 900 python tools/r4_scaled_diag.py 2>&1 | grep -v Warning > $O/scaled_diag.log; cat $O/scaled_diag.log
timeout 2400 python tools/config2_fp64_denoiser.py seeds=100 variants=default,fixed,fast32,f22 > $O/ens100.log 2>&1; cp gpurun_out/config2_fp64_denoiser.json $O/ens100_engines.json; grep -E "SUMMARY|DIFFS" $O/ens100.log
timeout 3000 python tools/config2_fp64_denoiser.py seeds=40 variants=fp64 > $O/ens_fp64.log 2>&1; cp gpurun_out/config2_fp64_denoiser.json $O/ens_fp64.json; tail -8 $O/ens_fp64.log
