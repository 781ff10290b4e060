# Builds the three in-tree shared libraries:
#   clraytracer_amd/csrc/libcrt_hip.so   HIP kernels + C-ABI (include/crt_api.h), gfx950 only
#   clraytracer_amd/host/libcrt_host.so  C++ host mirror of Renderer/ResourceManager/AssetManager (+ include/crt_host.h)
#   oracle/libcrt_oracle.so              CPU oracle (test infrastructure only)
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
ARCH ?= gfx950
# -fno-slp-vectorize: the SLP vectoriser turns the slab tests into v_pk_* with splat operands that cost the trace kernel 20+ VGPRs
# (96 -> 73) and spilled the 6-waves/SIMD flavour to scratch (290 MB per frame); without it the kernel fits 7 waves/SIMD.
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function
CXXFLAGS = -O2 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wall -Wextra -pthread

HIP_SO = clraytracer_amd/csrc/libcrt_hip.so
HOST_SO = clraytracer_amd/host/libcrt_host.so
HOST_SRC = $(addprefix clraytracer_amd/host/,AssetManager.cpp MeshCache.cpp JpegDecode.cpp BVH.cpp CPURayTrace.cpp Renderer.cpp ResourceManager.cpp crt_host_c.cpp ShmBarrier.cpp)
HOST_HDR = $(wildcard clraytracer_amd/host/*.hpp) $(wildcard include/*.h)

EXAMPLE = examples/crt_headless

UBENCH = tools/ubench/gather tools/ubench/chain tools/ubench/cumask

all: $(HIP_SO) $(HOST_SO) $(EXAMPLE) $(UBENCH) oracle

# vector-L1 gather microbenchmark (profiles/r01_ubench_gather.txt)
tools/ubench/gather: tools/ubench/gather.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ $<
# dependent-chain 64-B gather microbenchmark: the ceiling bench.py's roofline.chain is taken against (profiles/r03_ubench_chain.*)
tools/ubench/chain: tools/ubench/chain.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -ffp-contract=off -fno-slp-vectorize -Wno-unused-value -Wno-uninitialized -Wno-sometimes-uninitialized -o $@ $<

# CU-mask probe: hipExtStreamCreateWithCUMask on this part, and which CU each mask bit names (round 4's reserved-CU experiment)
tools/ubench/cumask: tools/ubench/cumask.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-value -o $@ $<

# the reference's EngineMain loop over the mirrored C++ API
$(EXAMPLE): examples/headless_main.cpp $(HOST_SO)
	$(CXX) $(CXXFLAGS) -o $@ examples/headless_main.cpp -Lclraytracer_amd/host -lcrt_host -Lclraytracer_amd/csrc -lcrt_hip -Wl,-rpath,'$$ORIGIN/../clraytracer_amd/host' -Wl,-rpath,'$$ORIGIN/../clraytracer_amd/csrc'

$(HIP_SO): $(wildcard clraytracer_amd/csrc/*.h) clraytracer_amd/csrc/crt_shim.hip include/crt_api.h include/crt_debug.h include/crt_types.h
	$(HIPCC) $(HIPFLAGS) -shared -o $@ clraytracer_amd/csrc/crt_shim.hip

$(HOST_SO): $(HOST_SRC) $(HOST_HDR) $(HIP_SO)
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRC) -Lclraytracer_amd/csrc -lcrt_hip -lrt -Wl,-rpath,'$$ORIGIN/../csrc'

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(HIP_SO) $(HOST_SO) $(EXAMPLE) $(UBENCH)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
