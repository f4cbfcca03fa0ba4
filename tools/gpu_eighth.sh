#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x -k "deterministic or bias_relu" 2>&1 | tail -3
for P in 0 1 2 3 0 3; do echo "policy $P"; DEQSCI_FORCE_POLICY=$P python tools/denoiser_bench.py 2>&1 | grep "channels_last+fused" ; done
