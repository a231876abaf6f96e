# Root build: libfamseq_hip.so (HIP, gfx950 only) + the test oracle.
#   make            -> famseq_amd/lib/libfamseq_hip.so
#   make oracle     -> oracle/liboracle_bn.so (+ oracle/_ref when /root/reference exists)
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := famseq_amd/csrc
LIB := famseq_amd/lib/libfamseq_hip.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -I$(CSRC) -Wall -Wno-unused-result

SRCS := $(CSRC)/bn_kernel.hip $(CSRC)/io_kernels.hip $(CSRC)/capi.cpp $(CSRC)/plan.cpp $(CSRC)/model.cpp $(CSRC)/jit.cpp $(CSRC)/elim_codegen.cpp $(CSRC)/enum_codegen.cpp
OBJS := $(patsubst $(CSRC)/%,build/%.o,$(SRCS))

CLI := bin/FamSeq

all: $(LIB) $(CLI)

build/%.o: $(CSRC)/% $(wildcard $(CSRC)/*.h) include/famseq_hip.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

# Link step.  The library must bind to whichever HIP runtime the process already uses:
# PyTorch-ROCm wheels bundle their own runtime under the SONAME "libamdhip64.so", the system
# ROCm one is "libamdhip64.so.7", and two HIP/HSA runtimes in one process cannot both own the
# GPU.  So we record the unversioned name as DT_NEEDED (via an empty link-time stub): inside a
# torch process (torch imported first) it resolves to torch's runtime, elsewhere to
# /opt/rocm/lib/libamdhip64.so through the RUNPATH.
build/stub/libamdhip64.so:
	@mkdir -p build/stub
	echo "" | gcc -shared -fPIC -x c - -Wl,-soname,libamdhip64.so -o $@

$(LIB): $(OBJS) build/stub/libamdhip64.so
	@mkdir -p famseq_amd/lib
	g++ -shared -fPIC -o $@ $(OBJS) -Wl,--no-as-needed -Lbuild/stub -lamdhip64 -Wl,--as-needed \
	    -Wl,-rpath,/opt/rocm/lib -Wl,--enable-new-dtags -ldl -lpthread

# FamSeq-compatible command line (host C++ only; talks to the GPU through the C ABI)
$(CLI): $(CSRC)/host/famseq_cli.cpp $(CSRC)/host/fmt_g6.h include/famseq_hip.h $(LIB)
	@mkdir -p bin
	g++ -O2 -std=c++17 -Wall -Iinclude -o $@ $(CSRC)/host/famseq_cli.cpp -Lfamseq_amd/lib -lfamseq_hip -lpthread \
	    -Wl,-rpath,'$$ORIGIN/../famseq_amd/lib' -Wl,-rpath,/opt/rocm/lib -Wl,--enable-new-dtags

# measuring aids: FETCH_SIZE / WRITE_SIZE calibration, the traffic shape's
# ceiling, sustained fp64 FMA rate, timing of hand-edited generated kernels
TOOLS := tools/calib_fetch tools/io_ceiling tools/fp64_latency tools/kernel_bench tools/coexec tools/write_sweep tools/fma_issue
tools/%: tools/%.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -Wno-unused-value -o $@ $<
tools: $(TOOLS)

# Sanitizer recipe: the library's HOST code — model, plan, both kernel generators, the
# JIT and the C ABI — rebuilt by g++ under AddressSanitizer + UBSan and driven on plan-only contexts by
# tests/asan_host_check.cpp.  CPU only (GPU AddressSanitizer is not available on the pool); the two
# kernel translation units are linked as they are (their host side is launch glue).  `make asan` builds
# and runs it; tests/test_abi.py runs it as part of the CPU suite.
ASAN_SRCS := capi.cpp plan.cpp model.cpp jit.cpp elim_codegen.cpp enum_codegen.cpp
ASAN_OBJS := $(patsubst %,build/asan/%.o,$(ASAN_SRCS))
ASAN_FLAGS := -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
              -ffp-contract=off -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -I$(CSRC) -Wall -Wno-unused-result
build/asan/%.o: $(CSRC)/% $(wildcard $(CSRC)/*.h) include/famseq_hip.h
	@mkdir -p build/asan
	g++ $(ASAN_FLAGS) -c $< -o $@
build/asan/host_check: tests/asan_host_check.cpp $(ASAN_OBJS) build/bn_kernel.hip.o build/io_kernels.hip.o
	g++ $(ASAN_FLAGS) -o $@ $^ -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -ldl -lpthread
# ... and the command line's own host code (line cutter, block parser, packed-file writer), the same way, through
# `FamSeq pack` (the mode that needs no GPU): TestData's full VCF six times over — blocks parsed on several threads —
# and the probe file (every echo / drop / GL rule).
build/asan/FamSeq: $(CSRC)/host/famseq_cli.cpp $(CSRC)/host/fmt_g6.h include/famseq_hip.h $(ASAN_OBJS) build/bn_kernel.hip.o build/io_kernels.hip.o
	g++ $(ASAN_FLAGS) -I$(CSRC)/host -o $@ $(CSRC)/host/famseq_cli.cpp $(ASAN_OBJS) build/bn_kernel.hip.o build/io_kernels.hip.o \
	    -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -ldl -lpthread
asan: build/asan/host_check build/asan/FamSeq
	ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 ./build/asan/host_check
	@mkdir -p build/asan/tmp && zcat tests/golden/testdata/test_full.vcf.gz > build/asan/tmp/one.vcf && \
	  (grep '^#' build/asan/tmp/one.vcf; for i in 1 2 3 4 5 6; do grep -v '^#' build/asan/tmp/one.vcf; done) > build/asan/tmp/six.vcf
	ASAN_OPTIONS=detect_leaks=1 FAMSEQ_THREADS=5 ./build/asan/FamSeq pack -vcfFile build/asan/tmp/six.vcf -pedFile tests/golden/testdata/fam01.ped -output build/asan/tmp/six.fspl
	ASAN_OPTIONS=detect_leaks=1 FAMSEQ_BATCH=7 ./build/asan/FamSeq pack -vcfFile tests/golden/testdata/probe.vcf -pedFile tests/golden/testdata/probe.ped -output build/asan/tmp/probe.fspl

oracle:
	$(MAKE) -C oracle all $(if $(wildcard /root/reference/src/family.cpp),ref,)

clean:
	rm -rf build famseq_amd/lib bin

.PHONY: all oracle clean tools asan
