# Builds libmi355zk.so (hand-written HIP for gfx950) and the CPU oracle used by the tests.
#   make            -> phase2-bn254_amd/libmi355zk.so + oracle/_build/liboracle.so
#   make -j8        translation units compile in parallel (msm_g2 is the long pole)
HIPCC ?= hipcc
ARCH ?= gfx950
PKG := phase2-bn254_amd
SRC := $(PKG)/csrc
HIPFLAGS ?= --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-result
OBJS := build/ntt.o build/msm_g1.o build/msm_g2.o build/api.o build/field_ops.o build/point_fft.o build/point_fft_g2.o build/codec.o
HDRS := $(SRC)/field.hpp $(SRC)/mont_mul_gfx950.inc $(SRC)/curve.hpp $(SRC)/fieldu.hpp $(SRC)/curveu.hpp $(SRC)/device_util.hpp $(SRC)/msm_impl.hpp include/mi355zk.h

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
