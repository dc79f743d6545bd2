# Builds everything in-tree (the .so files travel to the GPU box with gpurun; they are git-ignored):
#   montecarlooptionspricer_amd/lib/libmcgpu.so  -- the product: HIP kernels + C ABI + C++ drop-in classes
#   oracle/libmcgoracle.so, oracle/_ref/libmcref.so -- the parity oracle (test infrastructure)
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
PKG     := montecarlooptionspricer_amd
OBJDIR  := build/obj
LIB     := $(PKG)/lib/libmcgpu.so

# HIPFLAGS_EXTRA: experiment switches for timing studies (-DRB_NO_STORE: generator without its stores, -DRB_NBUF=1,
# -DRB_WAVES=3, -DRB_LT_DEFAULT=3), empty for the product build
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Iinclude -Wall -Wno-unused-function $(HIPFLAGS_EXTRA)
# host-only TUs: no FMA contraction so the estimators match the reference bit for bit
HOSTFLAGS := -O2 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Wall -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include

HIP_SRCS  := $(wildcard $(PKG)/csrc/*.hip)
HOST_SRCS := $(wildcard $(PKG)/host/*.cpp) $(wildcard $(PKG)/csrc/*.cpp)
OBJS := $(patsubst %,$(OBJDIR)/%.o,$(notdir $(HIP_SRCS) $(HOST_SRCS)))
HDRS := $(wildcard $(PKG)/csrc/*.hpp) $(wildcard include/*.h) $(wildcard include/mcgpu/*.hpp)

all: lib oracle

lib: $(LIB)

$(LIB): $(OBJS)
	@mkdir -p $(dir $@)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(OBJS) -ldl

$(OBJDIR)/%.hip.o: $(PKG)/csrc/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJDIR)/%.cpp.o: $(PKG)/host/%.cpp $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HOSTFLAGS) -x c++ -c $< -o $@

$(OBJDIR)/%.cpp.o: $(PKG)/csrc/%.cpp $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HOSTFLAGS) -x c++ -c $< -o $@

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf build $(PKG)/lib
	$(MAKE) -C oracle clean

.PHONY: all lib oracle clean

# C++ drop-in driver (tests/cpp): compiled against the reference-shaped headers with plain g++,
# linked to libmcgpu.so; run by tests/test_gpu_cpp_dropin.py on the GPU box.
build/dropin_driver: tests/cpp/dropin_driver.cpp $(LIB) $(HDRS)
	@mkdir -p build
	g++ -O2 -std=c++17 -fopenmp -Iinclude tests/cpp/dropin_driver.cpp -o $@ \
	    -L$(PKG)/lib -lmcgpu -Wl,-rpath,'$$ORIGIN/../$(PKG)/lib'

build/batch_driver: tests/cpp/batch_driver.cpp $(LIB) $(HDRS)
	@mkdir -p build
	g++ -O2 -std=c++17 -fopenmp -Iinclude tests/cpp/batch_driver.cpp -o $@ \
	    -L$(PKG)/lib -lmcgpu -Wl,-rpath,'$$ORIGIN/../$(PKG)/lib' -Wl,-rpath-link,/opt/rocm/lib

build/thread_ranks_driver: tests/cpp/thread_ranks_driver.cpp $(LIB) $(HDRS)
	@mkdir -p build
	g++ -O2 -std=c++17 -fopenmp -Iinclude tests/cpp/thread_ranks_driver.cpp -o $@ \
	    -L$(PKG)/lib -lmcgpu -Wl,-rpath,'$$ORIGIN/../$(PKG)/lib' -Wl,-rpath-link,/opt/rocm/lib

# the reference driver's row loop, unchanged in shape, as a measured workload (bench.py's "unchanged driver" row; tests/test_gpu_round6.py)
build/unchanged_driver: tests/cpp/unchanged_driver.cpp $(LIB) $(HDRS)
	@mkdir -p build
	g++ -O2 -std=c++17 -fopenmp -Iinclude tests/cpp/unchanged_driver.cpp -o $@ \
	    -L$(PKG)/lib -lmcgpu -Wl,-rpath,'$$ORIGIN/../$(PKG)/lib' -Wl,-rpath-link,/opt/rocm/lib

cpp: build/dropin_driver build/batch_driver build/thread_ranks_driver build/unchanged_driver
.PHONY: cpp

# Static check of the device assembly for the gfx940+ hazards hipcc does not cover inside inline-asm statements
# (tools/check_asm_hazards.py: VALU-written SGPRs read too soon by an asm statement; an asm store's data registers
# overwritten too soon).  Run by tests/test_asm_hazards.py.
ASM_SRCS := $(HIP_SRCS)
ASM_OUT  := $(patsubst %,build/asm/%.s,$(notdir $(ASM_SRCS)))
build/asm/%.hip.s: $(PKG)/csrc/%.hip $(HDRS)
	@mkdir -p build/asm
	$(HIPCC) $(HIPFLAGS) -S --cuda-device-only -o $@ $< 2>/dev/null
asmcheck: $(ASM_OUT)
	python3 tools/check_asm_hazards.py $(ASM_OUT)
.PHONY: asmcheck
