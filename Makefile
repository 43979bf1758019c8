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
HIPFLAGS := -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function
OBJDIR  := $(PKG)/_obj/make
# device code: host + gfx950 pass; the C-ABI (gbp_api_*.cpp, see gbp_ctx.hpp), the device order, the transports, the host helpers: plain C++
HOST_SRCS := gbp_api_ctx.cpp gbp_api_launch.cpp gbp_api_persist.cpp gbp_api_eval.cpp gbp_api_comm.cpp gbp_api_debug.cpp gbp_layout.cpp gbp_comm.cpp gbp_host.cpp
OBJS    := $(OBJDIR)/gbp_kernels.o $(HOST_SRCS:%.cpp=$(OBJDIR)/%.o)
HDRS    := $(wildcard $(CSRC)/*.h $(CSRC)/*.hpp $(CSRC)/hooks/* $(CSRC)/experiments/* include/*.h)

all: $(LIB) $(PKG)/bin/ba $(PKG)/bin/slam $(PKG)/bin/bal_convert

$(OBJDIR)/gbp_kernels.o: $(CSRC)/gbp_kernels.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) -c -o $@ $(HIPFLAGS) --offload-arch=$(ARCH) -x hip $<

$(OBJDIR)/%.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) -c -o $@ $(HIPFLAGS) -x c++ -D__HIP_PLATFORM_AMD__ -I$(dir $(shell readlink -f $$(which $(HIPCC))))../include $<

$(LIB): $(OBJS)
	$(HIPCC) -shared -o $@ --offload-arch=$(ARCH) $(OBJS) -ldl -pthread -Wl,--version-script=$(CSRC)/gbp_exports.map

$(PKG)/bin/%: $(CSRC)/%_main.cpp $(CSRC)/cli_common.hpp include/gbp_mi355x.h include/gbp_mi355x_multi.h include/gbp_mi355x_compat.h $(LIB)
	@mkdir -p $(PKG)/bin
	$(CXX) -o $@ -O2 -std=c++17 -ffp-contract=off -pthread $< -L$(PKG) -lgbp_mi355x '-Wl,-rpath,$$ORIGIN/..'

oracle:
	$(MAKE) -C oracle
	@if [ -d /root/reference/ba ]; then $(MAKE) -C oracle ref; fi

test: all oracle
	python -m pytest tests -x -q -m "not gpu"

clean:
	rm -rf $(OBJDIR) $(LIB) $(PKG)/bin/ba $(PKG)/bin/slam $(PKG)/bin/bal_convert
	$(MAKE) -C oracle clean

.PHONY: all oracle test clean
