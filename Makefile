# Builds the HIP library (product), the CPU lane emulator (tests only) and the
# C oracle (tests / cpu_baseline only).
HIPCC ?= hipcc
ARCH  ?= gfx950
CSRC  := cameracalibration_amd/csrc

all: lib hostemu oracle

lib: $(CSRC)/libcamcal.so
$(CSRC)/libcamcal.so: $(CSRC)/camcal_kernels.hip $(CSRC)/camcal_lane.h $(CSRC)/camcal_math.h include/camcal.h
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared $< -o $@

hostemu: tests/hostemu/libcamcal_hostemu.so
tests/hostemu/libcamcal_hostemu.so: tests/hostemu/hostemu.cpp $(CSRC)/camcal_lane.h $(CSRC)/camcal_math.h include/camcal.h
	g++ -O2 -std=c++17 -fPIC -shared $< -o $@

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(CSRC)/libcamcal.so tests/hostemu/libcamcal_hostemu.so
	$(MAKE) -C oracle clean

.PHONY: all lib hostemu oracle clean
