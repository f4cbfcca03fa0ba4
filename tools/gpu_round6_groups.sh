#!/bin/bash
# round 6: the grouped (two half batches on two streams) reconstruction - parity, then A/B against one stream
set -x
cd "$(dirname "$0")/.."
O=gpurun_out/r06_groups; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -k "grouped or config3 or engine_512x512x16_vs_oracle or deterministic_and_options or two_ranks" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
Q="--no-other-kernel --no-other-configs --no-cpu-baseline --no-hbm-stream --no-parity-check"
for g in 1 auto 1 auto; do
  python bench.py --steps 3 --warmup 1 --groups $g $Q 2>&1 | tail -1 >> $O/bench_ab.jsonl
done
python bench.py --steps 3 --warmup 1 --groups auto --anderson-arith float64 $Q 2>&1 | tail -1 >> $O/bench_ab.jsonl
python bench.py --steps 3 --warmup 1 --groups auto --denoiser SimpleCNN $Q 2>&1 | tail -1 >> $O/bench_ab.jsonl
python bench.py --steps 2 --warmup 1 --groups auto --batch-per-gpu 32 $Q 2>&1 | tail -1 >> $O/bench_ab.jsonl
python bench.py --steps 2 --warmup 1 --groups 1 --batch-per-gpu 32 $Q 2>&1 | tail -1 >> $O/bench_ab.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r06_groups/bench_ab.jsonl"):
    try: d=json.loads(l)
    except Exception: print("??", l[:200]); continue
    r=d.get("roofline",{})
    print(d["config"].get("groups"), d["config"]["anderson_arith"], d["config"]["batch_per_gpu"], "%.1f f/s"%d["value"], "%.1f ms"%d["ms_per_step"], "stack us", r.get("avg_launch_us"), "frac_useful", r.get("frac_useful"))
PY
tail -5 $O/tests.log
