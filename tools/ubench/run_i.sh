for cfg in "-DWG_PF=2" "-DWG_PF=3" "-DWG_PF=4" "-DWG_PF=1"; do
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Ideqsci_amd/csrc $cfg -shared -o deqsci_amd/lib/libdeqsci_hip.so deqsci_amd/csrc/*.hip || exit 1
echo "== $cfg"
for r in 1 2; do python tools/conv_bench.py 2>&1 | grep "^{" | python -c "import sys,json; print([json.loads(l)['winograd_mfma_fused_us'] for l in sys.stdin])"; done
done
