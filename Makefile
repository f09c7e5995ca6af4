# Builds the product library (HIP, gfx950) and the test oracle (plain C++).
#   make            both
#   make lib        real-time-video-mosaic_amd/lib/librtvm.so
#   make oracle     oracle/liboracle_rtvm.so
PKG      := real-time-video-mosaic_amd
CSRC     := $(PKG)/csrc
LIBDIR   := $(PKG)/lib
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
# -ffp-contract=off: the fp32/fp64 stages are specified operation by operation
# (no fused multiply-add), so that device results equal the CPU restatement's.
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -ffp-contract=off -fno-gpu-rdc -Wall -Wno-unused-function
HIPSRC   := $(wildcard $(CSRC)/*.hip)
HIPOBJ   := $(patsubst $(CSRC)/%.hip,build/%.o,$(HIPSRC))

all: lib oracle

lib: $(LIBDIR)/librtvm.so

# the rBRIEF pattern as a C initialiser list, generated from the data file
$(CSRC)/orb_pattern.inc: $(PKG)/data/orb_pattern_31.txt
	grep -v '^#' $< | awk '{printf "%d, %d, %d, %d,\n", $$1, $$2, $$3, $$4}' > $@

build/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/rtvm.h $(CSRC)/orb_pattern.inc
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/librtvm.so: $(HIPOBJ)
	@mkdir -p $(LIBDIR)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -fno-gpu-rdc -o $@ $(HIPOBJ)

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf build $(LIBDIR)/librtvm.so
	$(MAKE) -C oracle clean

.PHONY: all lib oracle clean
