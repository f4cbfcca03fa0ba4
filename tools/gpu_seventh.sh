#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/pytest_gpu.log
timeout 1500 python tools/parity_report.py 2>&1 | grep -v amdgpu.ids > gpurun_out/parity_report.log
tail -3 gpurun_out/pytest_gpu.log; cat gpurun_out/parity_report.log
