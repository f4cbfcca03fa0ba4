export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "head or tail or edge or ffdnet or conv" 2>&1 | tail -2
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tmp -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_tmp.log 2>&1
cd $R
python - <<'PY'
import csv
for r in list(csv.DictReader(open('gpurun_out/prof_tmp/bench_kernel_stats.csv')))[:6]:
    print(r['Name'][:50], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
grep "^{" gpurun_out/prof_tmp.log | cut -c1-120
