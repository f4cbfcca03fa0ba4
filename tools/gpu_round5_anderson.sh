#!/bin/bash
# Round 5, VERDICT item 2 (sourced by tools/gpu_round5.sh; or alone through gpurun): anderson_arith="reference" on the build's own kernels - numerics
# against float64 / torch.bmm / the CPU restatement, the two-pass form against the serial chains, cost per kernel, the step in the three arithmetics
# at one and eight measurements per call, WHICH fp32 rounding moves alpha on a real history, and the 100-start config-2 ensemble
# (FFDNet + Anderson @ 180 on the six chaotic measurements).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05p
mkdir -p $O
cd $R
timeout 600 python tools/anderson_ref_check.py 2> /dev/null | grep "^{" > $O/r05_anderson_ref_check.jsonl
for aa in reference float64 reference-bmm; do for b in 8 1; do
  timeout 600 python bench.py --steps 4 --warmup 2 --batch-per-gpu $b --anderson-arith $aa --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-other-configs --no-parity-check 2>&1 | grep "^{"
done; done > $O/r05_bench_anderson_arith.jsonl
timeout 400 python tools/gram_on_real_history.py 2> /dev/null | grep "^{\|^SUMMARY" > $O/r05_gram_on_real_history.jsonl
if [ -f build/diag/libdeqsci_hip_diag.so ]; then DEQSCI_HIP_LIB=build/diag/libdeqsci_hip_diag.so timeout 400 python tools/apply_stamps.py 2> /dev/null | grep "^[0-9]" > $O/r05_ref_apply_stamps.txt; fi
timeout 2400 python tools/config2_fp64_denoiser.py seeds=${SEEDS:-100} variants=${VARIANTS:-refarith} out=r05p/r05_config2_reference_arithmetic_100seeds.json > $R/gpurun_out/refarith.log 2>&1
grep "SUMMARY\|DIFFS" $R/gpurun_out/refarith.log
python - <<'PY'
import json
for l in open('gpurun_out/r05p/r05_bench_anderson_arith.jsonl'):
    d = json.loads(l); print(d['config'].get('anderson_arith'), d['config'].get('global_batch'), round(d['value'], 1), 'frames/s', round(d['ms_per_step'], 1), 'ms')
PY
