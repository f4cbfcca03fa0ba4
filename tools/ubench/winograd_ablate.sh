#!/bin/bash
# builds three variants of the library (full / no MFMA / no loads+transform after chunk 0) and times the conv
set -e
cd $GRAFT_REPO_ROOT
for v in 0 1 3 4; do
  mkdir -p /tmp/wgv$v/deqsci_amd/lib
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Ideqsci_amd/csrc -DWG_ABLATE=$v -shared -o /tmp/wgv$v/libdeqsci_hip.so deqsci_amd/csrc/*.hip
done
for v in 0 1 3 4; do
  cp /tmp/wgv$v/libdeqsci_hip.so deqsci_amd/lib/libdeqsci_hip.so
  echo "WG_ABLATE=$v: $(python tools/conv_bench.py 2>&1 | grep '^{' | head -1 | cut -c1-140)"
done
cp /tmp/wgv0/libdeqsci_hip.so deqsci_amd/lib/libdeqsci_hip.so
