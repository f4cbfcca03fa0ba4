#!/bin/bash
# Builds one copy of the HIP library per variant of ONE source file of csrc/ (compile-time knobs) into build/var_<file>/lib_<name>.so
# tools/lib_variants.sh conv_s16 "base:" "rows16:-DS16_HEAD_ROWS=16" ...      (DEQSCI_HIP_LIB=<that .so> selects it)
set -e
cd "$(dirname "$0")/.."
SRC=$1; shift
D=build/var_$SRC
mkdir -p $D
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Ideqsci_amd/csrc -Wall -Wno-unused-function -Wno-inline-asm"
OTHERS=""
for f in sci_ops anderson conv_w16 epilogue ffdnet_edges winograd winograd44 conv_s16; do
  [ $f = $SRC ] && continue
  OTHERS="$OTHERS $D/$f.o"
  if [ ! -f $D/$f.o ] || [ deqsci_amd/csrc/$f.hip -nt $D/$f.o ]; then
    /opt/rocm/bin/hipcc $FLAGS -c -o $D/$f.o deqsci_amd/csrc/$f.hip 2>/dev/null &
  fi
done
wait
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  ( mkdir -p $D/$name &&
    /opt/rocm/bin/hipcc $FLAGS $defs -c -o $D/$name/$SRC.o deqsci_amd/csrc/$SRC.hip 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $D/lib_$name.so $OTHERS $D/$name/$SRC.o &&
    echo "built $name ($defs)" ) &
done
wait
