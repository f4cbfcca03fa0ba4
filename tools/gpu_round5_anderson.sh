#!/bin/bash
# Round 5, VERDICT item 2: anderson_arith="reference" on the build's own fp32 Gram kernels - numerics, cost, the 100-start config-2 ensemble
# (FFDNet + Anderson @ 180 on the six chaotic measurements) beside round 4's torch.bmm form of the same arithmetic on the same box.
O=gpurun_out/r05a
mkdir -p $O
timeout 400 python tools/anderson_ref_check.py > $O/r05_anderson_ref_check.jsonl 2> $O/check_err.log
timeout 900 python -m pytest tests -m gpu -x -q -k "anderson or deq_loop or plugin or training or end_to_end or admm" > $O/tests_anderson.log 2>&1; tail -3 $O/tests_anderson.log
for aa in reference float64 reference-bmm; do for b in 8 1; do
  timeout 600 python bench.py --steps 4 --warmup 2 --batch-per-gpu $b --anderson-arith $aa --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-other-configs --no-parity-check 2>&1 | grep "^{"
done; done > $O/r05_bench_anderson_arith.jsonl
timeout 300 python tools/gram_on_real_history.py > $O/r05_gram_on_real_history.jsonl 2> $O/hist_err.log
timeout 2400 python tools/config2_fp64_denoiser.py seeds=${SEEDS:-100} variants=${VARIANTS:-refarith} out=r05a/r05_config2_reference_arithmetic_100seeds.json > $O/refarith.log 2>&1
grep "SUMMARY\|DIFFS" $O/refarith.log
