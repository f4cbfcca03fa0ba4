#!/bin/bash
# Builds one copy of the HIP library per variant of csrc/anderson.hip (compile-time knobs) into build/andv/lib_<name>.so
# tools/anderson_variants.sh "base:" "noterms:-DRSR_ABL=1" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p build/andv
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Ideqsci_amd/csrc -Wall -Wno-unused-function -Wno-inline-asm"
for f in sci_ops conv_w16 epilogue ffdnet_edges winograd winograd44 conv_s16; do
  if [ ! -f build/andv/$f.o ] || [ deqsci_amd/csrc/$f.hip -nt build/andv/$f.o ]; then
    /opt/rocm/bin/hipcc $FLAGS -c -o build/andv/$f.o deqsci_amd/csrc/$f.hip 2>/dev/null &
  fi
done
wait
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  ( mkdir -p build/andv/$name &&
    /opt/rocm/bin/hipcc $FLAGS $defs -c -o build/andv/$name/anderson.o deqsci_amd/csrc/anderson.hip 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/andv/lib_$name.so build/andv/{sci_ops,conv_w16,epilogue,ffdnet_edges,winograd,winograd44,conv_s16}.o build/andv/$name/anderson.o &&
    echo "built $name ($defs)" ) &
done
wait
