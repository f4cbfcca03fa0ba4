python -m pytest tests -m gpu -q -x -k "winograd or conv" 2>&1 | tail -3
python tools/conv_bench.py 2>&1 | grep "^{"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Ideqsci_amd/csrc -DWG_STAMP -DWG_STAMP_TID=0 -shared -o deqsci_amd/lib/libdeqsci_hip.so deqsci_amd/csrc/*.hip && python tools/ubench/winograd_stamps.py --timeline 2>&1 | grep -E "phases"
