# Top-level build: product libraries (handbrake_amd/) + checkers (oracle/).
#   make            everything
#   make product    libhbrt.so, libhbhip.so (HIP kernels + C-ABI), libhbhip_filters.so
#   make dev        the product libraries with -DHBHIP_DEV (tuning knobs from the environment; never shipped / tested)
#   make devlib     the same kernels library as build/dev/libhbhip.so, beside the product one (tools/exp_knobs.sh,
#                   tools/dev_run.sh); DEVDIR=build/devX DEVFLAGS=-D... for variants (-DHBHIP_DEV_STATS: search counters)
#   make oracle     liboracle.so and, when /root/reference exists, oracle/_ref/libhbref.so
HIPCC   ?= /opt/rocm/bin/hipcc
CC      ?= gcc
ARCH    ?= gfx950
PKG     := handbrake_amd
CFLAGS  := -std=gnu99 -O2 -g -fPIC -Wall -Wno-unused-function -Iinclude -I$(PKG)/libhb
HIPFLAGS:= --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -I$(PKG)/csrc -Wall -Wno-unused-function

HIP_SRC := $(wildcard $(PKG)/csrc/*.hip)
HIP_OBJ := $(HIP_SRC:%.hip=%.o)
HIP_HDR := $(wildcard $(PKG)/csrc/*.h) $(wildcard include/*.h)
FLT_SRC := $(wildcard $(PKG)/libhb/*_hip.c) $(PKG)/libhb/hbhip_registry.c $(PKG)/libhb/hip_common.c $(PKG)/libhb/vfr_standin.c

all: product oracle
product: $(PKG)/libhbrt.so $(PKG)/libhbhip.so $(PKG)/libhbhip_filters.so tools/vote_avg_check

# GPU-side exhaustive check of a device function shared with the kernels (tests/test_eedi2_gpu.py runs it)
tools/vote_avg_check: tools/vote_avg_check.hip $(PKG)/csrc/eedi2_vote.h
	$(HIPCC) --offload-arch=$(ARCH) -O3 -ffp-contract=off -fno-fast-math -I$(PKG)/csrc -Wno-unused-result $< -o $@

$(PKG)/libhbrt.so: $(PKG)/libhb/hb_runtime.c $(PKG)/libhb/hb_harness.c $(PKG)/libhb/hb_harness.h include/hbhip_libhb.h
	$(CC) $(CFLAGS) -shared -o $@ $(PKG)/libhb/hb_runtime.c $(PKG)/libhb/hb_harness.c -lm -lpthread

$(PKG)/csrc/%.o: $(PKG)/csrc/%.hip $(HIP_HDR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(PKG)/libhbhip.so: $(HIP_OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(HIP_OBJ)

$(PKG)/libhbhip_filters.so: $(FLT_SRC) $(PKG)/libhbrt.so $(PKG)/libhbhip.so include/hbhip.h include/hbhip_libhb.h $(wildcard $(PKG)/libhb/*.h)
	$(CC) $(CFLAGS) -shared -o $@ $(FLT_SRC) -L$(PKG) -lhbhip -lhbrt -lm -Wl,-rpath,'$$ORIGIN' -Wl,--no-undefined

oracle: $(PKG)/libhbrt.so
	$(MAKE) -C oracle all

# development build of the kernels library: environment tuning knobs and HBHIP_SKIP_KERNELS compiled in
dev:
	rm -f $(HIP_OBJ)
	$(MAKE) product HIPFLAGS="$(HIPFLAGS) -DHBHIP_DEV"
	rm -f $(HIP_OBJ)

# the development kernels library BESIDE the product one (build/dev/libhbhip.so, own objects): what tools/exp_knobs.sh swaps
# in on the GPU box for knob experiments and swaps out again; nothing of the product build is touched
DEVDIR   ?= build/dev
DEVFLAGS ?=
DEV_OBJ := $(HIP_SRC:$(PKG)/csrc/%.hip=$(DEVDIR)/%.o)
$(DEVDIR)/%.o: $(PKG)/csrc/%.hip $(HIP_HDR)
	@mkdir -p $(DEVDIR)
	$(HIPCC) $(HIPFLAGS) -DHBHIP_DEV $(DEVFLAGS) -c $< -o $@
$(DEVDIR)/libhbhip.so: $(DEV_OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(DEV_OBJ)
devlib: $(DEVDIR)/libhbhip.so

clean:
	rm -rf build/dev
	rm -f $(PKG)/*.so $(PKG)/csrc/*.o
	$(MAKE) -C oracle clean

.PHONY: all product oracle clean dev devlib
