#!/bin/bash
# Where do the cycles of a Winograd pipeline stage go?  Builds the library with s_memtime stamps (-DWG_STAMP) and prints the
# per-phase timeline of one CU (tools/ubench/winograd_stamps.py), once as shipped and once per timing ablation
# (WG_ABL 1 = no patch reads, 2 = no weight DMA / raw fetch inside the MFMA phase, 3 = both; results are wrong then).
# WG_STAMP_SKIP skips that many marks first (3 per stage), WG_STAMP_TID picks the stamping lane (0 = wave 0, 256 = wave 4).
set -e
cd $GRAFT_REPO_ROOT
SKIP=${WG_STAMP_SKIP:-194}
for A in 0 1 2 3; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Ideqsci_amd/csrc -DWG_STAMP -DWG_STAMP_TID=${WG_STAMP_TID:-0} \
        -DWG_STAMP_SKIP=$SKIP -DWG_ABL=$A -shared -o deqsci_amd/lib/libdeqsci_hip.so deqsci_amd/csrc/*.hip
  echo "WG_ABL=$A"
  python tools/ubench/winograd_stamps.py --timeline 2>&1 | grep -E "phases|lifetime"
done
make -B deqsci_amd/lib/libdeqsci_hip.so > /dev/null     # back to the product build
