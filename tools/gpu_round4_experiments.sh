#!/bin/bash
# The experiments of round 4 whose logs are committed under profiles/r04_* (second table of profiles/README.md), as stages:
#   R4_EXP="fp64 config2 anderson scaled variants" bash tools/gpu_round4_experiments.sh   (default: all; run on the GPU box through gpurun)
# fp64      BASELINE config 2 with the denoiser evaluated in float64 (40 starts; 13 minutes)
# config2   BASELINE config 2 ensembles on the engine: default / fixed scales / fast32 / f22 / the reference's Anderson arithmetic (100
#           starts), the family of 12 equivalent arithmetics (50 starts each)
# anderson  the reference's Anderson step as torch ops on the GPU around the build's f: fp32 / float64 Gram, fp32 / float64 LU (50 starts)
# scaled    FFDNet @30 / SimpleCNN @180 on measurements scaled by 1e-2 ... 1e2 under every policy (conditioning vs arithmetic)
# variants  conv_s16 variants on one box, interleaved rounds (needs the libraries of tools/s16_variants.sh, built in the container):
#           bash tools/s16_variants.sh "rows2:-DS16_ROWS4=0" "r4p:-DS16_ZEROC=0 -DS16_TILE_VOFF=0" "r4z:-DS16_TILE_VOFF=0" "r4v:-DS16_ZEROC=0" "r4zv:" "r4x:-DS16_ABL=16"
mkdir -p gpurun_out/r04x
O=gpurun_out/r04x
want() { [[ -z "$R4_EXP" || " $R4_EXP " == *" $1 "* ]]; }
if want fp64; then
  timeout 3000 python tools/config2_fp64_denoiser.py seeds=40 variants=fp64 out=r04_config2_fp64_denoiser.json > $O/fp64.log 2>&1
  grep -h -E "SUMMARY" $O/fp64.log | cut -c1-400; cp gpurun_out/r04_config2_fp64_denoiser.json $O/
fi
if want config2; then
  timeout 1500 python tools/config2_fp64_denoiser.py seeds=100 variants=default,fixed,fast32,f22 out=r04_config2_ensembles_100seeds.json > $O/ens100.log 2>&1
  timeout 1500 python tools/config2_fp64_denoiser.py seeds=100 variants=refarith out=r04_config2_reference_arithmetic_100seeds.json > $O/refarith.log 2>&1
  timeout 1500 python tools/config2_fp64_denoiser.py seeds=50 out=r04_config2_family_scatter.json variants=fast+1,fast+2,fast+3,fast+4,fast+5,fast+6,fast+7,fast+8,fast+10,fast+12,fast+16,fast+20 > $O/family.log 2>&1
  grep -h -E "SUMMARY|FAMILY" $O/ens100.log $O/refarith.log $O/family.log | cut -c1-900
  cp gpurun_out/r04_config2_*.json $O/
fi
if want anderson; then
  timeout 1500 python tools/config2_anderson_arith.py seeds=50 denoiser=miopen variants=g32s32,g64s32,g64s64 out=r04_config2_anderson_arith.json > $O/anderson_arith.log 2>&1
  grep -E "SUMMARY" $O/anderson_arith.log | cut -c1-600; cp gpurun_out/r04_config2_anderson_arith.json $O/
fi
if want scaled; then
  timeout 900 python tools/r4_scaled_diag.py 2>&1 | grep -v Warning > $O/r04_scaled_measurements_by_policy.txt; cat $O/r04_scaled_measurements_by_policy.txt | cut -c1-300
fi
if want variants; then
  for rnd in 1 2 3; do
    for v in rows2 r4p r4z r4v r4zv r4x; do
      [ -f build/s16v/lib_$v.so ] && DEQSCI_HIP_LIB=build/s16v/lib_$v.so timeout 120 python tools/s16_time.py 2>&1 | grep "^{"
    done
  done | tee $O/r04_s16_variants.jsonl
  for v in rows2 r4zv rows2 r4zv; do
    [ -f build/s16v/lib_$v.so ] && DEQSCI_HIP_LIB=build/s16v/lib_$v.so timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-parity-check 2>&1 | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), 'fps', round(d['roofline']['avg_launch_us'],2), 'us', round(d['roofline']['frac'],4))"
  done | tee $O/r04_s16_rows4_bench_ab.txt
  for v in rows2 r4zv; do for n in 64 8; do [ -f build/s16v/lib_$v.so ] && DEQSCI_HIP_LIB=build/s16v/lib_$v.so PROBE_IMAGES=$n timeout 300 python tools/power_probe.py 2>&1 | grep "^{" | tail -1; done; done | tee $O/r04_power_probe_rows4.jsonl
fi
