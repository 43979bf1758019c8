# Top-level build for C/C++ users (the Python entry point __graft_entry__.build() does the same).
#   make            libgbp_mi355x.so (HIP kernels + C-ABI + host helpers, gfx950) and bin/ba, bin/slam, bin/bal_convert
#   make oracle     the CPU oracle (test infrastructure)
#   make test       CPU test suite
HIPCC   ?= hipcc
CXX     ?= g++
ARCH    ?= gfx950
PKG     := gbp_poplar_amd
CSRC    := $(PKG)/csrc
LIB     := $(PKG)/libgbp_mi355x.so
HIPFLAGS := -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden --offload-arch=$(ARCH) -Wall -Wno-unused-function

all: $(LIB) $(PKG)/bin/ba $(PKG)/bin/slam $(PKG)/bin/bal_convert

$(LIB): $(CSRC)/gbp_kernels.hip $(CSRC)/gbp_capi.cpp $(CSRC)/gbp_layout.cpp $(CSRC)/gbp_layout.hpp $(CSRC)/gbp_comm.cpp $(CSRC)/gbp_comm.hpp $(CSRC)/gbp_host.cpp $(CSRC)/gbp_kernels.h $(CSRC)/gbp_device_math.hpp include/gbp_mi355x.h
	$(HIPCC) -shared -o $@ $(HIPFLAGS) -x hip $(CSRC)/gbp_kernels.hip $(CSRC)/gbp_capi.cpp $(CSRC)/gbp_layout.cpp $(CSRC)/gbp_comm.cpp $(CSRC)/gbp_host.cpp -ldl -Wl,--version-script=$(CSRC)/gbp_exports.map

$(PKG)/bin/%: $(CSRC)/%_main.cpp $(CSRC)/cli_common.hpp $(LIB)
	@mkdir -p $(PKG)/bin
	$(CXX) -o $@ -O2 -std=c++17 -ffp-contract=off -pthread $< -L$(PKG) -lgbp_mi355x '-Wl,-rpath,$$ORIGIN/..'

oracle:
	$(MAKE) -C oracle
	@if [ -d /root/reference/ba ]; then $(MAKE) -C oracle ref; fi

test: all oracle
	python -m pytest tests -x -q -m "not gpu"

clean:
	rm -f $(LIB) $(PKG)/bin/ba $(PKG)/bin/slam $(PKG)/bin/bal_convert
	$(MAKE) -C oracle clean

.PHONY: all oracle test clean
