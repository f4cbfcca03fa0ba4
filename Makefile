# Builds the C-ABI HIP library (gfx950 only) and the plain-C oracle helper.
HIPCC      ?= /opt/rocm/bin/hipcc
ARCH       ?= gfx950
HIPFLAGS   := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Ideqsci_amd/csrc -Wall -Wno-unused-function
LIB        := deqsci_amd/lib/libdeqsci_hip.so
SRCS       := deqsci_amd/csrc/conv_w16.hip deqsci_amd/csrc/sci_ops.hip deqsci_amd/csrc/anderson.hip deqsci_amd/csrc/epilogue.hip deqsci_amd/csrc/ffdnet_edges.hip deqsci_amd/csrc/winograd.hip deqsci_amd/csrc/winograd44.hip deqsci_amd/csrc/conv_s16.hip
HDRS       := include/deqsci_hip.h deqsci_amd/csrc/common.hpp
ORACLE_LIB := oracle/libdeqsci_oracle.so

all: $(LIB) $(ORACLE_LIB)

$(LIB): $(SRCS) $(HDRS)
	@mkdir -p deqsci_amd/lib
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(SRCS)

$(ORACLE_LIB): oracle/deqsci_oracle.c include/deqsci_hip.h
	gcc -O2 -std=c11 -fPIC -shared -ffp-contract=off -Iinclude -o $@ oracle/deqsci_oracle.c -lm

# diagnostic variant for tools/ (env knobs DEQSCI_GRAM_NOISE, DEQSCI_K4_BLOCKS, DEQSCI_FORCE_POLICY, DEQSCI_HEAD_VALU); never loaded by the
# package unless DEQSCI_HIP_LIB points at it
DIAG_LIB   := build/diag/libdeqsci_hip_diag.so
diag: $(DIAG_LIB)
$(DIAG_LIB): $(SRCS) $(HDRS)
	@mkdir -p build/diag
	$(HIPCC) $(HIPFLAGS) -DDEQSCI_DIAG -shared -o $@ $(SRCS)

clean:
	rm -f $(LIB) $(ORACLE_LIB) $(DIAG_LIB)

.PHONY: all clean diag
