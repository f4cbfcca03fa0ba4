#!/bin/bash
# round 4, pass D: conv_s16 variants on one box, interleaved rounds (box drift): 4-row geometry plain / + zero-C / + per-tile DMA offsets / both / + extra reads
mkdir -p gpurun_out/r04d
O=gpurun_out/r04d
for v in r4zv r4x; do DEQSCI_HIP_LIB=build/s16v/lib_$v.so timeout 300 python tools/s16_check.py check 2>&1 | tail -1; done | tee $O/check.txt
for rnd in 1 2 3; do
  for v in base r4p r4z r4v r4zv r4x; do
    DEQSCI_HIP_LIB=build/s16v/lib_$v.so timeout 120 python tools/s16_time.py 2>&1 | grep "^{"
  done
done | tee $O/s16_time.jsonl
for v in base r4zv base r4zv; do
  DEQSCI_HIP_LIB=build/s16v/lib_$v.so timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-parity-check 2>&1 | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), 'fps', round(d['roofline']['avg_launch_us'],2), 'us', round(d['roofline']['frac'],4))" | tee -a $O/bench_ab.txt
done
for v in base r4zv base r4zv; do
  DEQSCI_HIP_LIB=build/s16v/lib_$v.so timeout 600 python bench.py --steps 6 --warmup 2 --batch-per-gpu 1 --no-cpu-baseline --no-hbm-stream --no-other-kernel --no-parity-check 2>&1 | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v bsz1', round(d['value'],2), 'fps')" | tee -a $O/bench_ab.txt
done
