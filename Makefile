# Builds the HIP library (product), the CPU lane emulator (tests only) and the
# C oracle (tests / cpu_baseline only).
HIPCC ?= hipcc
ARCH  ?= gfx950
CSRC  := cameracalibration_amd/csrc
# the GCN register-pressure trackers keep 9 more accumulator words of the single-pass
# kernel in architectural VGPRs (481 instead of 499 instructions per observation, -2.5 % time)
HIPFLAGS_EXTRA ?= -mllvm -amdgpu-use-amdgpu-trackers=1

all: lib hostemu oracle examples

lib: $(CSRC)/libcamcal.so
CSRC_HDRS := $(wildcard $(CSRC)/*.h) include/camcal.h
$(CSRC)/libcamcal.so: $(CSRC)/camcal_kernels.hip $(CSRC)/camcal_layout.hip $(CSRC_HDRS)
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared $(HIPFLAGS_EXTRA) $(CSRC)/camcal_kernels.hip $(CSRC)/camcal_layout.hip -o $@

hostemu: tests/hostemu/libcamcal_hostemu.so
tests/hostemu/libcamcal_hostemu.so: tests/hostemu/hostemu.cpp $(CSRC_HDRS)
	g++ -O2 -std=c++17 -fPIC -shared -pthread $< -o $@

oracle:
	$(MAKE) -C oracle

# pure C-ABI host program (no Python, no torch); run on the GPU box by tests/test_gpu_cabi.py
examples: examples/cabi_fisheye examples/cabi_rig
examples/cabi_%: examples/cabi_%.cpp include/camcal.h $(CSRC)/libcamcal.so
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 $< -o $@ -L$(CSRC) -lcamcal -Wl,-rpath,'$$ORIGIN/../$(CSRC)'

# issue-rate microbenchmarks behind DESIGN.md section 4 (run on the GPU box: tools/ubench/ubench)
ubench: tools/ubench/ubench
tools/ubench/ubench: tools/ubench/ubench.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 $< -o $@

clean:
	rm -f tools/ubench/ubench $(CSRC)/libcamcal.so tests/hostemu/libcamcal_hostemu.so examples/cabi_fisheye examples/cabi_rig
	$(MAKE) -C oracle clean

.PHONY: all lib hostemu oracle examples ubench clean
