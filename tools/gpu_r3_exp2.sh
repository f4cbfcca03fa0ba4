#!/bin/bash
# round 3: spill-free F(4x4,3x3) kernel - correctness + time; config-2 ensembles with the bordered system solved in fp32 (as torch.solve)
mkdir -p gpurun_out/r03
O=gpurun_out/r03
python -m pytest tests/test_gpu_parity.py -q -x -k "winograd44 or conv64 or conv_layout or blk32" 2>&1 | tail -5 > $O/w44_tests.txt
python tools/w44_check.py time > $O/w44_time_nospill.txt 2>&1
python tools/w44_fuzz.py > $O/w44_fuzz.txt 2>&1
DEQSCI_HIP_LIB=build/diag/libdeqsci_hip_diag.so DEQSCI_SOLVE_F32=1 DEQSCI_ENSEMBLE_CONFIGS="F(2x2)" DEQSCI_ENSEMBLE_SEEDS=25 DEQSCI_ENSEMBLE_TRAFFIC_ONLY=1 timeout 1500 python tools/config2_ensemble.py > $O/ensemble25_solvef32.txt 2>&1
cp gpurun_out/config2_ensemble.json $O/config2_ensemble25_solvef32.json
DEQSCI_HIP_LIB=build/diag/libdeqsci_hip_diag.so DEQSCI_SOLVE_F32=1 DEQSCI_GRAM_NOISE=2e-6 DEQSCI_ENSEMBLE_CONFIGS="F(2x2)" DEQSCI_ENSEMBLE_SEEDS=25 DEQSCI_ENSEMBLE_TRAFFIC_ONLY=1 timeout 1500 python tools/config2_ensemble.py > $O/ensemble25_solvef32_noise2e-6.txt 2>&1
cp gpurun_out/config2_ensemble.json $O/config2_ensemble25_solvef32_noise2e-6.json
cat $O/w44_tests.txt; tail -4 $O/w44_time_nospill.txt; tail -2 $O/w44_fuzz.txt; grep -v SUMMARY $O/ensemble25_solvef32.txt | cut -c1-400; grep SUMMARY $O/ensemble25_solvef32.txt;  grep -v SUMMARY $O/ensemble25_solvef32_noise2e-6.txt | cut -c1-400; grep SUMMARY $O/ensemble25_solvef32_noise2e-6.txt
