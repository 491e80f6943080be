# Builds vq_amd/libvqhip.so (HIP, gfx950 only) and the CPU oracle.
# hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     := gfx950
CSRC     := vq_amd/csrc
# atomic optimizer off: it turns a one-lane returning atomic into atomic + immediate readfirstlane, i.e. a
# wait for the atomic's round trip right where it is issued (k_resident_pipe claims its batches ahead of time)
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -mllvm -amdgpu-atomic-optimizer-strategy=None -Wall -Wextra -Wno-unused-parameter -Iinclude
OBJS     := $(CSRC)/api.o $(CSRC)/kernels_scalar.o $(CSRC)/kernels_engine.o $(CSRC)/kernels_resident.o $(CSRC)/kernels_omega.o $(CSRC)/kernels_local.o $(CSRC)/kernels_long.o $(CSRC)/kernels_direct.o $(CSRC)/kernels_glvq.o \
            $(CSRC)/kernels_project.o $(CSRC)/engine_host.o $(CSRC)/synth.o
HDRS     := $(CSRC)/vq_common.h $(CSRC)/engine.h $(CSRC)/mfma_device.h $(CSRC)/mfma_host.h $(CSRC)/mfma_images.h include/vqhip.h
LIB      := vq_amd/libvqhip.so
# the same objects as a static archive: what a build.rs links with `cargo:rustc-link-lib=static=vqhip`
LIBA     := vq_amd/libvqhip.a
ROCM_LIB ?= /opt/rocm/lib

MODELS_TEST := tests/cpp/test_models
ROWS_BENCH  := tools/host_rows_bench

MODELS_TEST_STATIC := tests/cpp/test_models_static

all: $(LIB) $(LIBA) oracle $(MODELS_TEST) $(MODELS_TEST_STATIC) $(ROWS_BENCH) tools/hbm_stream tools/bf16x3_probe tools/bf16x6_probe tools/issue_probe tools/local_probe asan

$(LIBA): $(OBJS)
	rm -f $@ && ar rcs $@ $(OBJS)

# the C++ mirror's tests linked against the ARCHIVE with plain g++, the way a Rust build would link it:
# the archive + the HIP runtime + the C++ runtime, nothing else (RCCL is bound by dlopen at first use)
$(MODELS_TEST_STATIC): tests/cpp/test_models.cpp include/vq_models.hpp include/vqhip.h $(LIBA)
	g++ -O2 -std=c++17 -Wall -Wextra -Iinclude $< -o $@ $(LIBA) -L$(ROCM_LIB) -lamdhip64 -ldl -lpthread -Wl,-rpath,$(ROCM_LIB)

tools/hbm_stream: tools/hbm_stream.hip
	$(HIPCC) -O3 --offload-arch=$(ARCH) $< -o $@
# measurement tools behind DESIGN.md: accuracy of the split-bf16 products, issue cost of the search's vector instructions
tools/bf16x3_probe: tools/bf16x3_probe.hip
	$(HIPCC) -O3 --offload-arch=$(ARCH) $< -o $@
tools/bf16x6_probe: tools/bf16x6_probe.hip
	$(HIPCC) -O3 --offload-arch=$(ARCH) $< -o $@
tools/local_probe: tools/local_probe.hip $(CSRC)/mfma_device.h
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -Iinclude -I$(CSRC) $< -o $@
tools/issue_probe: tools/issue_probe.hip
	$(HIPCC) -O3 --offload-arch=$(ARCH) $< -o $@

# host-side code of the library (the image builders, duplicate detection, centring) under AddressSanitizer and UBSan,
# CPU only: tests/cpp/test_host_asan.cpp drives engine_build's host half without a device
asan: tests/cpp/test_host_asan
tests/cpp/test_host_asan: tests/cpp/test_host_asan.cpp $(CSRC)/engine_host.hip $(HDRS)
	$(HIPCC) -O1 -g -std=c++17 --offload-arch=$(ARCH) -fno-gpu-sanitize -fsanitize=address,undefined -fno-omit-frame-pointer \
	    -Iinclude -I$(CSRC) -DVQHIP_HOST_TEST $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ $(OBJS) -ldl

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
	rm -f $(OBJS) $(LIB) $(LIBA) $(MODELS_TEST) $(MODELS_TEST_STATIC) $(ROWS_BENCH) tools/hbm_stream tests/cpp/test_host_asan
	$(MAKE) -C oracle -s clean

.PHONY: all oracle clean asan
