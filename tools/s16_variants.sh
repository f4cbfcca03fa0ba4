#!/bin/bash
# Builds one copy of the HIP library per variant of csrc/conv_s16.hip (compile-time knobs) into build/s16v/lib_<name>.so;
# `DEQSCI_HIP_LIB=build/s16v/lib_<name>.so python tools/s16_check.py time` times it.   tools/s16_variants.sh "base:" "nodma:-DS16_ABL=1" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p build/s16v
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Ideqsci_amd/csrc -Wall -Wno-unused-function -Wno-inline-asm"
for f in sci_ops anderson epilogue ffdnet_edges winograd winograd44; do
  if [ ! -f build/s16v/$f.o ] || [ deqsci_amd/csrc/$f.hip -nt build/s16v/$f.o ]; then
    /opt/rocm/bin/hipcc $FLAGS -c -o build/s16v/$f.o deqsci_amd/csrc/$f.hip &
  fi
done
wait
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  ( mkdir -p build/s16v/$name &&
    /opt/rocm/bin/hipcc $FLAGS $defs -c -save-temps=obj -o build/s16v/$name/s16.o ${S16_SRC:-deqsci_amd/csrc/conv_s16.hip} 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/s16v/lib_$name.so build/s16v/{sci_ops,anderson,epilogue,ffdnet_edges,winograd,winograd44}.o build/s16v/$name/s16.o &&
    echo "built $name ($defs) $(grep -h -E 'vgpr_count|vgpr_spill_count' build/s16v/$name/*gfx950.s 2>/dev/null | tr '\n' ' ' | tr -s ' ')" ) &
done
wait
