# Builds the MI355X engine: hand-written HIP kernels + C-ABI -> icp_amd/libicp_amd.so (gfx950 only).
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
# -ffp-contract=off: the canonical arithmetic has no fused multiply-add
# EXTRA: -D switches of an A/B build; BUILD: its object directory — e.g. `make BUILD=build/ol32 LIB=build/libicp_ol32.so EXTRA=-DICP_OL_BOXED_MIN=32 build/libicp_ol32.so`
EXTRA    ?=
BUILD    ?= build
HIPFLAGS ?= $(EXTRA) -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=$(ARCH) -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result -pthread -mllvm -amdgpu-kernarg-preload-count=14
SRC      := icp_amd/csrc/icp_kernels.hip icp_amd/csrc/icp_search_dense.hip icp_amd/csrc/icp_build.hip icp_amd/csrc/icp_capi.hip icp_amd/csrc/icp_run.hip icp_amd/csrc/icp_track.hip icp_amd/csrc/icp_reduce_scan.hip icp_amd/csrc/icp_standalone.hip icp_amd/csrc/icp_synth.cpp icp_amd/csrc/icp_batch.cpp
HDR      := icp_amd/csrc/icp_device.h icp_amd/csrc/icp_kernels.h icp_amd/csrc/icp_search.h icp_amd/csrc/icp_host.h icp_amd/csrc/icp_cguard.h include/icp_amd.h
LIB      ?= icp_amd/libicp_amd.so
OBJ      := $(patsubst icp_amd/csrc/%,$(BUILD)/%.o,$(SRC))

all: $(LIB) oracle

# one object per source (the two k_search translation units are the long ones: `make` builds them side by side)
$(BUILD)/%.o: icp_amd/csrc/% $(HDR)
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(LIB): $(OBJ)
	$(HIPCC) -fPIC --offload-arch=$(ARCH) -pthread -shared -o $@ $(OBJ)

oracle:
	$(MAKE) -C oracle

facade_test: $(LIB) tests/cpp/facade_test.cpp include/ICP/algorithms.hpp
	g++ -O2 -std=c++17 -Iinclude -o tests/cpp/facade_test tests/cpp/facade_test.cpp -Licp_amd -licp_amd -Wl,-rpath,'$$ORIGIN/../../icp_amd'

icpreg_test: $(LIB) tests/cpp/icpreg_test.cpp include/ocl_icp_reg.hpp include/ocl_icp_sbs.hpp include/ICP/algorithms.hpp
	g++ -O2 -std=c++17 -Iinclude -o tests/cpp/icpreg_test tests/cpp/icpreg_test.cpp -Licp_amd -licp_amd -Wl,-rpath,'$$ORIGIN/../../icp_amd'

# the reference's two example programs as command-line programs (examples/registration.cpp, examples/step_by_step.cpp)
examples: $(LIB) examples/registration.cpp examples/step_by_step.cpp include/ocl_icp_reg.hpp include/ocl_icp_sbs.hpp include/ICP/algorithms.hpp
	g++ -O2 -std=c++17 -Wall -Iinclude -o examples/registration examples/registration.cpp -Licp_amd -licp_amd -Wl,-rpath,'$$ORIGIN/../icp_amd'
	g++ -O2 -std=c++17 -Wall -Iinclude -o examples/step_by_step examples/step_by_step.cpp -Licp_amd -licp_amd -Wl,-rpath,'$$ORIGIN/../icp_amd'

# measurement: SIMD cycles per wave64 vector instruction by class (bench.py prices SQ_INSTS_VALU with the result: profiles/valu_issue.json)
valu_issue_probe: tests/cpp/valu_issue_probe.hip
	$(HIPCC) --offload-arch=$(ARCH) -O2 -o tests/cpp/valu_issue_probe tests/cpp/valu_issue_probe.hip

capi_example: $(LIB) tests/cpp/capi_example.c include/icp_amd.h
	gcc -O2 -std=c99 -Wall -Iinclude -o tests/cpp/capi_example tests/cpp/capi_example.c -Licp_amd -licp_amd -Wl,-rpath,'$$ORIGIN/../../icp_amd'

# Host-side sanitizer build (GPU ASan is not available on this pool): the CPU oracle, the synthetic
# generator and a driver over every oracle entry point at small / ragged / degenerate sizes, AddressSanitizer + UBSan.
# No OpenMP (its runtime is not instrumented).  `make asan` builds and runs it; tests/test_oracle_golden.py does too.
ASANFLAGS := -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -ffp-contract=off -fno-fast-math -mfma -Wall -Wno-unknown-pragmas
tests/cpp/asan_host: tests/cpp/asan_host.cpp oracle/icp_oracle.c oracle/icp_oracle.h icp_amd/csrc/icp_synth.cpp include/icp_amd.h
	gcc $(ASANFLAGS) -std=gnu11 -c oracle/icp_oracle.c -o tests/cpp/asan_oracle.o
	g++ $(ASANFLAGS) -std=c++17 -o $@ tests/cpp/asan_host.cpp icp_amd/csrc/icp_synth.cpp tests/cpp/asan_oracle.o -lm

asan: tests/cpp/asan_host
	ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 tests/cpp/asan_host

clean:
	rm -rf build; rm -f $(LIB) examples/registration examples/step_by_step tests/cpp/facade_test tests/cpp/capi_example tests/cpp/icpreg_test tests/cpp/asan_host tests/cpp/asan_oracle.o
	$(MAKE) -C oracle clean

.PHONY: all oracle clean asan examples
