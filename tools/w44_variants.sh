#!/bin/bash
# Builds one copy of the HIP library per variant of csrc/winograd44.hip (compile-time knobs) into build/w44v/lib_<name>.so;
# `W44_LIB=build/w44v/lib_<name>.so python tools/w44_check.py time` times it.   tools/w44_variants.sh "base:" "notr:-DW44_ABL=16" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p build/w44v
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Ideqsci_amd/csrc -Wall -Wno-unused-function -Wno-inline-asm"
for f in sci_ops anderson epilogue ffdnet_edges winograd; do
  if [ ! -f build/w44v/$f.o ] || [ deqsci_amd/csrc/$f.hip -nt build/w44v/$f.o ]; then
    /opt/rocm/bin/hipcc $FLAGS -c -o build/w44v/$f.o deqsci_amd/csrc/$f.hip &
  fi
done
wait
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  ( mkdir -p build/w44v/$name &&      # (own directory: the -save-temps files of parallel builds would otherwise overwrite each other)
    /opt/rocm/bin/hipcc $FLAGS $defs -c -save-temps=obj -o build/w44v/$name/w44.o ${W44_SRC:-deqsci_amd/csrc/winograd44.hip} 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/w44v/lib_$name.so build/w44v/{sci_ops,anderson,epilogue,ffdnet_edges,winograd}.o build/w44v/$name/w44.o &&
    echo "built $name ($defs) $(grep -h -E 'vgpr_spill_count|NumVgprs' build/w44v/$name/*gfx950.s 2>/dev/null | tr '\n' ' ')" &&
    # a variant that spills is not a measurement of the design (tests/test_cabi_exports.py::test_winograd44_kernel_has_no_spills holds the product to it)
    if grep -q -E 'scratch_(load|store)|v_writelane' build/w44v/$name/*gfx950.s; then echo "SPILLS in variant $name" >&2; touch build/w44v/$name/SPILLS; fi ) &
done
wait
if ls build/w44v/*/SPILLS >/dev/null 2>&1; then rm -f build/w44v/*/SPILLS; exit 3; fi
