#!/bin/bash
# round 6: K4 fused with the reference Gram's first pass - parity, then the step's kernel stats
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_gram; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "reference_gram or anderson or graph_replay or grouped or deterministic or end_to_end or config3" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
Q="--no-other-kernel --no-other-configs --no-cpu-baseline --no-hbm-stream"
for g in auto 1; do
  python bench.py --steps 3 --warmup 1 --groups $g $Q 2>&1 | tail -1 > $O/bench_g$g.json
  python - <<PY
import json
d=json.load(open("$O/bench_g$g.json")); r=d.get("roofline",{})
print("groups $g: %.1f f/s %.1f ms stack %.1f us parity %s" % (d["value"], d["ms_per_step"], r.get("avg_launch_us",0), d.get("parity_spot_check",{}).get("rel_l2")))
PY
done
python bench.py --steps 3 --warmup 1 --batch-per-gpu 1 $Q 2>&1 | tail -1 > $O/bench_b1.json
python -c "
import json; d=json.load(open('$O/bench_b1.json')); print('bsz 1: %.1f f/s' % d['value'], d['config']['launch_mode'])"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --groups 1 $Q --no-parity-check --no-kernel-timing > $O/tr.log 2>&1
python3 $R/tools/trace_window.py $(find $O/tr -name "*kernel_trace.csv" | head -1) 0 | head -14
cp $(find $O/tr -name "*kernel_stats.csv" | head -1) $O/kernel_stats_g1.csv; rm -rf $O/tr
