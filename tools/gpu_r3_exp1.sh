#!/bin/bash
# round 3, first look: fp16 MFMA numerics; error of each 64->64 conv kernel on the network's own data; larger config-2 ensembles
mkdir -p gpurun_out/r03
O=gpurun_out/r03
./build/ub/mfma_f16_numerics > $O/mfma_f16_numerics.txt 2>&1
timeout 600 python tools/conv_error_real.py > $O/conv_error_real.jsonl 2> $O/conv_error_real.err
DEQSCI_ENSEMBLE_SEEDS=25 DEQSCI_ENSEMBLE_TRAFFIC_ONLY=1 timeout 1500 python tools/config2_ensemble.py > $O/ensemble25.txt 2> $O/ensemble25.err
cp gpurun_out/config2_ensemble.json $O/config2_ensemble25.json
cat $O/mfma_f16_numerics.txt; tail -3 $O/conv_error_real.err; python - <<'PY'
import json
for l in open('gpurun_out/r03/conv_error_real.jsonl'):
    d=json.loads(l); print(d['input'],d['layer'],'f22 %.2e f44 %.2e miopen %.2e'%(d['f22'],d['f44'],d['miopen']),d['f44_by_row_mod4'],d['f44_by_col_mod4'],d['f44_border_share'],'mean/rms %.3f %.3f'%(d['f44_mean_err_over_rms'],d['f22_mean_err_over_rms']))
PY
grep SUMMARY $O/ensemble25.txt; tail -3 $O/ensemble25.err
