# Builds the MI355X engine (libstn.so, gfx950 only) and the CPU oracle.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := supertonic_amd/csrc
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -Iinclude
KERNELS := $(CSRC)/kernels_gemm.hip $(CSRC)/kernels_misc.hip $(CSRC)/kernels_attn.hip
HOSTSRC := $(CSRC)/engine.cpp $(CSRC)/api.cpp $(wildcard $(CSRC)/host/*.cpp)
OBJS    := $(patsubst %.hip,build/%.o,$(KERNELS)) $(patsubst %.cpp,build/%.o,$(HOSTSRC))
HDRS    := $(wildcard $(CSRC)/*.hpp $(CSRC)/host/*.hpp include/*.h)

all: supertonic_amd/libstn.so supertonic_amd/example_native oracle

supertonic_amd/libstn.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

supertonic_amd/example_native: $(CSRC)/cli/example_native.cpp supertonic_amd/libstn.so $(HDRS)
	$(HIPCC) -O2 -std=c++17 -Iinclude -o $@ $< -Lsupertonic_amd -lstn -Wl,-rpath,'$$ORIGIN'

build/%.o: %.hip $(HDRS)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

build/%.o: %.cpp $(HDRS)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

oracle:
	$(MAKE) -C oracle -s

clean:
	rm -rf build supertonic_amd/libstn.so supertonic_amd/example_native
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
