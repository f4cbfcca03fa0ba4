#!/bin/bash
# Builds one copy of the HIP library per Winograd-kernel variant (compile-time knobs of csrc/winograd.hip) into
# build/wgv/lib_<name>.so; tools/wg_variant_bench.py then checks and times each on the GPU box.
#   tools/wg_variants.sh "base:" "pf3:-DWG_PF=3" ...            (WG_SRC=tools/ubench/variants/winograd_v3_lds_dma.hip builds the
#   LDS-DMA staging variant of round 2 instead of the product kernel)
set -e
cd "$(dirname "$0")/.."
mkdir -p build/wgv
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Ideqsci_amd/csrc -Wall -Wno-unused-function"
for f in sci_ops anderson epilogue ffdnet_edges; do
  if [ ! -f build/wgv/$f.o ] || [ deqsci_amd/csrc/$f.hip -nt build/wgv/$f.o ]; then
    /opt/rocm/bin/hipcc $FLAGS -c -o build/wgv/$f.o deqsci_amd/csrc/$f.hip &
  fi
done
wait
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c -o build/wgv/wg_$name.o deqsci_amd/csrc/winograd.hip &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/wgv/lib_$name.so build/wgv/sci_ops.o build/wgv/anderson.o build/wgv/epilogue.o build/wgv/ffdnet_edges.o build/wgv/wg_$name.o &&
    echo "built $name ($defs)" ) &
done
wait
