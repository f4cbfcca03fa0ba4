#!/bin/bash
# Regenerates everything committed under profiles/r04_* that profiles/README.md lists in its first round-4 table (run on the GPU box through
# gpurun).  Stages can be skipped: R4_SKIP="parity pmc"
# The two stamp files need the profiling build first: bash tools/s16_variants.sh "stamp:-DS16_STAMP"  (build/s16v/lib_stamp.so)
mkdir -p gpurun_out/pmc gpurun_out/r04p
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04p
cd $R
skip() { [[ " $R4_SKIP " == *" $1 "* ]]; }
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 ) > $O/r04_gputest_summary.txt
timeout 900 python bench.py --steps 5 --warmup 1 2>&1 | grep "^{" > $O/r04_bench_n1.json
timeout 900 python bench.py --steps 3 --warmup 1 --denoiser SimpleCNN --no-cpu-baseline --no-hbm-stream 2>&1 | grep "^{" > $O/r04_bench_n1_simplecnn.json
( timeout 900 python bench.py --steps 1 --warmup 1 --size 512x512x16 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-parity-check 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 6 --warmup 2 --batch-per-gpu 1 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-parity-check 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 6 --warmup 2 --batch-per-gpu 1 --no-graph --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-parity-check 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 6 --warmup 2 --batch-per-gpu 1 --no-stack --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-parity-check 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 4 --warmup 2 --batch-per-gpu 2 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-parity-check 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 4 --warmup 2 --batch-per-gpu 2 --no-stack --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-parity-check 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 1 --warmup 1 --batch-per-gpu 32 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-parity-check 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 1 --warmup 1 --global-batch 64 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-parity-check 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 3 --warmup 1 --act-range fixed --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-parity-check 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 3 --warmup 1 --no-stack --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-parity-check 2>&1 | grep "^{";
  timeout 900 python bench.py --steps 1 --warmup 1 --batch-per-gpu 32 --no-stack --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-parity-check 2>&1 | grep "^{" ) > $O/r04_bench_other_shapes.jsonl
timeout 300 python tools/kernel_bench.py 2>&1 | grep "^{" > $O/r04_kernel_bench_bsz64.jsonl
timeout 300 python tools/conv_bench.py 2>&1 | grep "^{" > $O/r04_conv_bench.jsonl
( timeout 300 python tools/stack_bench.py 2>&1 | grep "^{"; STACK_IMAGES=8 timeout 300 python tools/stack_bench.py 2>&1 | grep "^{" ) > $O/r04_stack_bench.jsonl
timeout 300 python tools/s16_check.py both 2>&1 | grep -v amdgpu > $O/r04_s16_check.txt
( for n in 64 8; do PROBE_IMAGES=$n timeout 200 python tools/power_probe.py 2>&1 | grep "^{" | tail -1; done
  PROBE_IMAGES=8 PROBE_KERNEL=stack timeout 200 python tools/power_probe.py 2>&1 | grep "^{" | tail -1 ) > $O/r04_power_probe.jsonl
if [ -f build/s16v/lib_stamp.so ]; then
  for n in 64 8; do echo "== $n images of 128 x 128"; S16_IMAGES=$n timeout 200 python tools/s16_stamps.py 2>&1 | grep -v amdgpu; done > $O/r04_s16_stamps.txt
  timeout 200 python tools/s16_stack_stamps.py 2>&1 | grep -v amdgpu > $O/r04_s16_stack_stamps.txt
fi
python - > $O/r04_act_ranges.json <<'PY'
import json, os, sys, torch
sys.path.insert(0, os.getcwd())
from deqsci_amd import checkpoint, _hip
from deqsci_amd.cli import build_pipeline
from deqsci_amd.engine import DEQSCIEngine
from deqsci_amd.harness import SCITestDataset, as_clip
out = {}
for kind, w in (("ffdnet", "ffdnet_gray"), ("SimpleCNN", "cnn")):
    net = build_pipeline(kind, checkpoint.shipped(w), 30)[0].nonlinear_op
    eng = DEQSCIEngine(net, max_iter=30, use_graph=False)
    for clip in (as_clip(c) for c in SCITestDataset("data/test_gray")):
        eng.reconstruct(clip["meas"].permute(2, 0, 1).contiguous().cuda(), clip["mask"][None].cuda())
        r = eng.last_info["act_ranges"]
        out[f"{kind}:{clip['file']}"] = {"max_abs_per_layer_input": [round(v, 4) for v in r], "exponents": [_hip.act_exp(v) for v in r]}
print(json.dumps(out, indent=1))
PY
if ! skip parity; then
  timeout 1500 python tools/parity_report.py 4 > $O/parity_report.log 2>&1
  cp gpurun_out/parity_report.md $O/r04_parity_report.md
fi
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-parity-check > $R/gpurun_out/prof_bench.log 2>&1
cp $(find $R/gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1) $O/r04_bench_kernel_stats.csv
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bsz1 -o b1 -- python3 $R/bench.py --batch-per-gpu 1 --steps 2 --warmup 2 --no-cpu-baseline --no-hbm-stream --no-kernel-timing --no-other-kernel --no-parity-check > $R/gpurun_out/prof_bsz1.log 2>&1
cp $(find $R/gpurun_out/prof_bsz1 -name "*kernel_stats.csv" | head -1) $O/r04_bench_bsz1_graph_kernel_stats.csv
cd $R
if ! skip pmc; then
  bash tools/pmc_winograd.sh > /dev/null 2>&1
  cp gpurun_out/pmc_conv_s16.json $O/r04_pmc_conv_s16.json
  cp gpurun_out/pmc_conv_s16_stack.json $O/r04_pmc_conv_s16_stack.json
  cp gpurun_out/pmc_winograd44.json $O/r04_pmc_winograd44.json
  cd /tmp
  for C in FETCH_SIZE WRITE_SIZE; do
    for B in 64 8; do
      timeout 600 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc/${C}_b$B -o k -- python3 $R/tools/kernel_bench.py --bsz $B --launches 9 --sets 1 > $R/gpurun_out/pmc/${C}_b$B.log 2>&1
    done
  done
  cd $R
  python tools/pmc_summarize.py gpurun_out/pmc 64 8 > $O/r04_pmc_hbm_traffic.json
fi
ls -la $O; cat $O/r04_gputest_summary.txt; head -c 700 $O/r04_bench_n1.json; echo; python -c "
import json
for ln in open('$O/r04_bench_other_shapes.jsonl'): d=json.loads(ln); print(round(d['value'],2), d['config']['workload'][:70], d['config'].get('launch_mode','')[:20], d['config'].get('stack_launches_per_step'))
d=json.load(open('$O/r04_pmc_conv_s16.json')); print({k: d.get(k) for k in ('mfma_busy_fraction','non_mfma_valu_per_mfma','traffic_over_algorithmic','lds_bank_conflict_share','hbm_bytes_per_launch')})"
cat $O/r04_power_probe.jsonl | cut -c1-400; cat $O/r04_s16_stamps.txt | head -30
