#!/bin/bash
set -e
cd $GRAFT_REPO_ROOT
cp deqsci_amd/lib/libdeqsci_hip.so /tmp/orig.so
for v in 0 1 2 4; do
  if [ $v = 0 ]; then D=""; else D="-DWG_STAGGER=$v"; fi
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Ideqsci_amd/csrc $D -shared -o deqsci_amd/lib/libdeqsci_hip.so deqsci_amd/csrc/*.hip
  echo "WG_STAGGER=$v: $(python tools/conv_bench.py 2>&1 | grep '^{' | head -1 | cut -c60-140)"
done
cp /tmp/orig.so deqsci_amd/lib/libdeqsci_hip.so
