#!/bin/bash
# Builds one copy of the HIP library per variant of csrc/conv_w16.hip (compile-time knobs) into build/w16v/lib_<name>.so;
# `DEQSCI_HIP_LIB=build/w16v/lib_<name>.so python tools/w16_check.py time` times it.   tools/w16_variants.sh "base:" "nodma:-DW16_ABL=1" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p build/w16v
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Ideqsci_amd/csrc -Wall -Wno-unused-function -Wno-inline-asm"
for f in sci_ops anderson epilogue ffdnet_edges winograd winograd44 conv_s16; do
  if [ ! -f build/w16v/$f.o ] || [ deqsci_amd/csrc/$f.hip -nt build/w16v/$f.o ]; then
    /opt/rocm/bin/hipcc $FLAGS -c -o build/w16v/$f.o deqsci_amd/csrc/$f.hip &
  fi
done
wait
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  ( mkdir -p build/w16v/$name &&
    /opt/rocm/bin/hipcc $FLAGS $defs -c -o build/w16v/$name/w16.o ${W16_SRC:-deqsci_amd/csrc/conv_w16.hip} 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/w16v/lib_$name.so build/w16v/{sci_ops,anderson,epilogue,ffdnet_edges,winograd,winograd44,conv_s16}.o build/w16v/$name/w16.o &&
    echo "built $name ($defs)" ) &
done
wait
