# Builds the MI355X engine (libstn.so, gfx950 only) and the CPU oracle.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := supertonic_amd/csrc
# NOPKF32: no packed-fp32 VALU ops in device code.  On MI355X a v_pk_{mul,fma,add}_f32 whose op_sel crosses register halves
# returns wrong results in lanes 48-63 while another wave's MFMA is executing (tools/probe/pk_probe.hip reproduces it with
# five instructions).  The compiler forms such ops freely, so the feature is switched off for the
# device pass (the host pass prints "not a recognized feature", which is expected).
NOPKF32 := -Xclang -target-feature -Xclang -packed-fp32-ops
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -Iinclude $(NOPKF32) $(EXTRA)
KERNELS := $(CSRC)/kernels_gemm.hip $(CSRC)/kernels_misc.hip $(CSRC)/kernels_attn.hip $(CSRC)/kernels_xattn_hs.hip $(CSRC)/kernels_ffn.hip
HOSTSRC := $(CSRC)/engine.cpp $(CSRC)/engine_batch.cpp $(CSRC)/engine_ops.cpp $(CSRC)/api.cpp $(CSRC)/group.cpp $(wildcard $(CSRC)/host/*.cpp)
OBJS    := $(patsubst %.hip,build/%.o,$(KERNELS)) $(patsubst %.cpp,build/%.o,$(HOSTSRC))
HDRS    := $(wildcard $(CSRC)/*.hpp $(CSRC)/*.inc $(CSRC)/host/*.hpp include/*.h)

all: supertonic_amd/libstn.so supertonic_amd/example_native oracle

supertonic_amd/libstn.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl

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

# hardware probe for the packed-fp32 erratum (not part of the product; run on a GPU box: build/pk_probe 2000)
probe: build/pk_probe
build/pk_probe: tools/probe/pk_probe.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -o $@ $<
# L2 -> CU ingest rate with every CU streaming (the wall DESIGN.md section 9.1 prices the fused FFN kernels against; run on a GPU box: build/l2_ingest_probe)
probe-ingest: build/l2_ingest_probe
build/l2_ingest_probe: tools/probe/l2_ingest_probe.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -o $@ $<
# the fixed cost of one dependent launch in a replayed graph (empty, with the fused kernels' resources, producer / consumer): build/launch_floor_probe
probe-launch: build/launch_floor_probe
build/launch_floor_probe: tools/probe/launch_floor_probe.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -o $@ $<

# Sanitizer build of the host-side parsers (CPU only; no GPU sanitizer exists on this pool): everything under csrc/host/ that reads
# caller-supplied files and strings — the protobuf reader, the graph binder, the JSON reader, the text frontend, the voice-style loader —
# as a host-only shared library (tests/test_host_asan_cpu.py runs the host test files against it: STN_LIB, STN_HOST_ONLY=1, libasan
# preloaded) and a corpus driver (tools/host_fuzz.cpp).  tts_host.cpp (TextToSpeech: calls the engine ABI) is not part of it.
ASAN_FLAGS := -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -fPIC -Iinclude -Wall
HOST_ONLY  := $(filter-out $(CSRC)/host/tts_host.cpp,$(wildcard $(CSRC)/host/*.cpp))
host-asan: build_asan/libstn_host_asan.so build_asan/host_fuzz
build_asan/libstn_host_asan.so: $(HOST_ONLY) $(HDRS)
	@mkdir -p build_asan
	g++ $(ASAN_FLAGS) -shared -o $@ $(HOST_ONLY)
build_asan/host_fuzz: tools/host_fuzz.cpp $(HOST_ONLY) $(HDRS)
	@mkdir -p build_asan
	g++ $(ASAN_FLAGS) -o $@ tools/host_fuzz.cpp $(HOST_ONLY)

clean:
	rm -rf build build_asan supertonic_amd/libstn.so supertonic_amd/example_native
	$(MAKE) -C oracle clean

.PHONY: all oracle clean probe probe-ingest probe-launch host-asan
