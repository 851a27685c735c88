# Builds the MI355X engine: hand-written HIP kernels + C-ABI -> icp_amd/libicp_amd.so (gfx950 only).
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
# -ffp-contract=off: the canonical arithmetic has no fused multiply-add
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result -mllvm -amdgpu-kernarg-preload-count=14
SRC      := icp_amd/csrc/icp_kernels.hip icp_amd/csrc/icp_capi.hip icp_amd/csrc/icp_reduce_scan.hip icp_amd/csrc/icp_synth.cpp
HDR      := icp_amd/csrc/icp_device.h icp_amd/csrc/icp_kernels.h include/icp_amd.h
LIB      := icp_amd/libicp_amd.so

all: $(LIB) oracle

$(LIB): $(SRC) $(HDR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(SRC)

oracle:
	$(MAKE) -C oracle

facade_test: $(LIB) tests/cpp/facade_test.cpp include/ICP/algorithms.hpp
	g++ -O2 -std=c++17 -Iinclude -o tests/cpp/facade_test tests/cpp/facade_test.cpp -Licp_amd -licp_amd -Wl,-rpath,'$$ORIGIN/../../icp_amd'

icpreg_test: $(LIB) tests/cpp/icpreg_test.cpp include/ocl_icp_reg.hpp include/ocl_icp_sbs.hpp include/ICP/algorithms.hpp
	g++ -O2 -std=c++17 -Iinclude -o tests/cpp/icpreg_test tests/cpp/icpreg_test.cpp -Licp_amd -licp_amd -Wl,-rpath,'$$ORIGIN/../../icp_amd'

capi_example: $(LIB) tests/cpp/capi_example.c include/icp_amd.h
	gcc -O2 -std=c99 -Wall -Iinclude -o tests/cpp/capi_example tests/cpp/capi_example.c -Licp_amd -licp_amd -Wl,-rpath,'$$ORIGIN/../../icp_amd'

clean:
	rm -f $(LIB) tests/cpp/facade_test tests/cpp/capi_example tests/cpp/icpreg_test
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
