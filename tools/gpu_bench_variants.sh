export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --steps 2 --warmup 1 2>&1 | grep "^{" > gpurun_out/bench.log
timeout 900 python bench.py --steps 1 --warmup 1 --denoiser SimpleCNN --no-cpu-baseline 2>&1 | grep "^{" > gpurun_out/bench_simplecnn.log
timeout 900 python bench.py --steps 1 --warmup 1 --size 512x512x16 --no-cpu-baseline 2>&1 | grep "^{" > gpurun_out/bench_512.log
timeout 900 python bench.py --steps 2 --warmup 1 --batch-per-gpu 1 --no-cpu-baseline 2>&1 | grep "^{" > gpurun_out/bench_bsz1.log
timeout 900 python bench.py --steps 1 --warmup 1 --batch-per-gpu 32 --no-cpu-baseline 2>&1 | grep "^{" > gpurun_out/bench_bsz32.log
timeout 1500 python tools/parity_report.py > gpurun_out/parity_report.log 2>&1

cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
cd $R
for f in bench bench_simplecnn bench_512 bench_bsz1 bench_bsz32; do python -c "
import json,sys
d=json.loads(open('gpurun_out/$f.log').read()); print('$f', round(d['value'],2), round(d['ms_per_step'],1), {k:round(v,3) if isinstance(v,float) else v for k,v in d.get('roofline',{}).items() if k in ('frac','avg_launch_us','traffic','share_of_step_time')})"; done
tail -3 gpurun_out/parity_report.log
