#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/gap_variants tools/ubench/gap_variants.hip && timeout 300 /tmp/gap_variants > gpurun_out/gap_variants.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -x -k "bias_relu or variants or channels_last" 2>&1 | tail -8 > gpurun_out/pytest_gpu2.log
cat gpurun_out/gap_variants.log; tail -8 gpurun_out/pytest_gpu2.log
