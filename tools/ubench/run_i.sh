for P in 0 32; do
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Ideqsci_amd/csrc -DWG_RAW_PAD=$P -shared -o deqsci_amd/lib/libdeqsci_hip.so deqsci_amd/csrc/*.hip || exit 1
echo "PAD=$P"; for r in 1 2; do python tools/conv_bench.py 2>&1 | grep "^{" | python -c "import sys,json; print([json.loads(l)['winograd_mfma_fused_us'] for l in sys.stdin])"; done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Ideqsci_amd/csrc -DWG_RAW_PAD=$P -DWG_STAMP -DWG_STAMP_TID=0 -DWG_STAMP_SKIP=194 -shared -o deqsci_amd/lib/libdeqsci_hip.so deqsci_amd/csrc/*.hip && python tools/ubench/winograd_stamps.py --timeline 2>&1 | grep -E "phases"
done
