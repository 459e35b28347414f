# Builds libmi355zk.so (hand-written HIP for gfx950) and the CPU oracle used by the tests.
#   make            -> phase2-bn254_amd/libmi355zk.so + oracle/_build/liboracle.so
#   make -j8        translation units compile in parallel (msm_g2 is the long pole)
HIPCC ?= hipcc
ARCH ?= gfx950
PKG := phase2-bn254_amd
SRC := $(PKG)/csrc
HIPFLAGS ?= --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-result
OBJS := build/ntt.o build/msm_g1.o build/msm_g2.o build/api.o build/host_entry.o build/scalar_mul.o build/field_ops.o build/point_fft.o build/point_fft_g2.o build/codec.o
HDRS := $(SRC)/field.hpp $(SRC)/mont_mul_gfx950.inc $(SRC)/curve.hpp $(SRC)/fieldu.hpp $(SRC)/curveu.hpp $(SRC)/device_util.hpp $(SRC)/msm_impl.hpp $(SRC)/api_internal.hpp $(SRC)/glv.hpp include/mi355zk.h

all: $(PKG)/libmi355zk.so oracle tools/bin/ubench_valu tools/bin/ubench_gather tools/bin/ubench_fieldmul tools/bin/ubench_wave_bucket tools/bin/ubench_gather_footprint

# standalone microbenchmarks (instruction issue rates; FETCH_SIZE calibration) used by tools/refresh_profiles.sh
tools/bin/%: tools/%.hip $(HDRS)
	@mkdir -p tools/bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -Iinclude -Wno-unused-value -Wno-unused-result $< -o $@

$(PKG)/libmi355zk.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

build/%.o: $(SRC)/%.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf build $(PKG)/libmi355zk.so
	$(MAKE) -C oracle clean

.PHONY: all oracle clean

# ---- sanitizer build: the same sources with the HOST side instrumented by AddressSanitizer + UndefinedBehaviorSanitizer
# (-fno-gpu-sanitize: the gfx950 code objects are the product's).  `make asan` -> tools/bin/libmi355zk_asan.so; run a python process over it with
#   LD_PRELOAD=$(ASAN_RT) ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 MI355ZK_SO=tools/bin/libmi355zk_asan.so
# tests/test_asan_host.py (CPU: the host arithmetic / self-test hooks / argument paths) and tests/test_gpu_ubsan.py (GPU, UBSan build below) do that.
ASAN_RT := $(shell /opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so 2>/dev/null)
ASAN_OBJS := $(patsubst build/%.o,build_asan/%.o,$(OBJS))
asan: tools/bin/libmi355zk_asan.so
tools/bin/libmi355zk_asan.so: $(ASAN_OBJS)
	@mkdir -p tools/bin
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -fsanitize=address,undefined -shared-libsan -o $@ $(ASAN_OBJS)
build_asan/%.o: $(SRC)/%.hip $(HDRS)
	@mkdir -p build_asan
	$(HIPCC) $(HIPFLAGS) -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -shared-libsan -c $< -o $@
# The GPU box cannot run the ASan build: ROCm's ASan runtime intercepts hsa_amd_memory_pool_allocate and needs an xnack / ASan-enabled ROCr ("AddressSanitizer:
# out of memory ... hsa_amd_memory_pool_allocate" at the first device allocation: profiles/r06_asan_on_gpu.txt).  What runs WITH device work is the
# UBSan-only build (no interceptors): `make ubsan` -> tools/bin/libmi355zk_ubsan.so, -fno-sanitize-recover: the first report aborts the process.
UBSAN_OBJS := $(patsubst build/%.o,build_ubsan/%.o,$(OBJS))
ubsan: tools/bin/libmi355zk_ubsan.so
tools/bin/libmi355zk_ubsan.so: $(UBSAN_OBJS)
	@mkdir -p tools/bin
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -fsanitize=undefined -shared-libsan -Wl,-rpath,$(dir $(ASAN_RT)) -o $@ $(UBSAN_OBJS)
build_ubsan/%.o: $(SRC)/%.hip $(HDRS)
	@mkdir -p build_ubsan
	$(HIPCC) $(HIPFLAGS) -fsanitize=undefined -fno-gpu-sanitize -fno-sanitize-recover=undefined -fno-omit-frame-pointer -shared-libsan -c $< -o $@
# ThreadSanitizer on the host side (`make tsan` -> tools/bin/libmi355zk_tsan.so; tools/run_tsan.sh <command>): runs with device work, ~10 x slower; the HIP runtime is not
# instrumented, so its internal races are suppressed (tools/tsan.supp) and only reports that name this library count (tests/test_gpu_tsan.py; profiles/r06_tsan.txt).
TSAN_OBJS := $(patsubst build/%.o,build_tsan/%.o,$(OBJS))
tsan: tools/bin/libmi355zk_tsan.so
tools/bin/libmi355zk_tsan.so: $(TSAN_OBJS)
	@mkdir -p tools/bin
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -fsanitize=thread -shared-libsan -o $@ $(TSAN_OBJS)
build_tsan/%.o: $(SRC)/%.hip $(HDRS)
	@mkdir -p build_tsan
	$(HIPCC) $(HIPFLAGS) -fsanitize=thread -fno-gpu-sanitize -fno-omit-frame-pointer -shared-libsan -c $< -o $@
.PHONY: asan ubsan tsan
