# Builds vq_amd/libvqhip.so (HIP, gfx950 only) and the CPU oracle.
# hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     := gfx950
CSRC     := vq_amd/csrc
# atomic optimizer off: it turns a one-lane returning atomic into atomic + immediate readfirstlane, i.e. a
# wait for the atomic's round trip right where it is issued (k_resident_pipe claims its batches ahead of time)
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -mllvm -amdgpu-atomic-optimizer-strategy=None -Wall -Wextra -Wno-unused-parameter -Iinclude
OBJS     := $(CSRC)/api.o $(CSRC)/kernels_scalar.o $(CSRC)/kernels_engine.o $(CSRC)/kernels_resident.o $(CSRC)/kernels_omega.o $(CSRC)/kernels_long.o $(CSRC)/kernels_direct.o \
            $(CSRC)/kernels_project.o $(CSRC)/engine_host.o $(CSRC)/synth.o
HDRS     := $(CSRC)/vq_common.h $(CSRC)/engine.h $(CSRC)/mfma_device.h $(CSRC)/mfma_host.h $(CSRC)/mfma_images.h include/vqhip.h
LIB      := vq_amd/libvqhip.so

MODELS_TEST := tests/cpp/test_models
ROWS_BENCH  := tools/host_rows_bench

all: $(LIB) oracle $(MODELS_TEST) $(ROWS_BENCH)

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ $(OBJS)

$(CSRC)/%.o: $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/%.o: $(CSRC)/%.cpp $(HDRS)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

# the C++ host mirror (include/vq_models.hpp) with its tests; plain g++ against the C ABI
$(MODELS_TEST): tests/cpp/test_models.cpp include/vq_models.hpp include/vqhip.h $(LIB)
	g++ -O2 -std=c++17 -Wall -Wextra -Iinclude $< -o $@ -Lvq_amd -lvqhip -Wl,-rpath,'$$ORIGIN/../../vq_amd'

# native timing of the host-pointer entry points (PCIe-inclusive)
$(ROWS_BENCH): tools/host_rows_bench.cpp include/vqhip.h $(LIB)
	g++ -O2 -std=c++17 -Wall -Wextra -Iinclude $< -o $@ -Lvq_amd -lvqhip -Wl,-rpath,'$$ORIGIN/../vq_amd'

oracle:
	$(MAKE) -C oracle -s libvqoracle.so

clean:
	rm -f $(OBJS) $(LIB) $(MODELS_TEST) $(ROWS_BENCH)
	$(MAKE) -C oracle -s clean

.PHONY: all oracle clean
