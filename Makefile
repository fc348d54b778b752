# Build of the MI355X (gfx950) product and of the test-side helpers.  `python -c "import __graft_entry__ as g; g.build()"`
# drives this file.  hipcc cross-compiles gfx950 without a GPU.
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
ARCH    ?= gfx950
# -ffp-contract=off: the decode path must round exactly like the CPU oracle (no FMA contraction)
# (the cycle counters of the trellis wavefronts, AUGX_PROF=1, are a developer build: `make prof` -> augustus_amd/libaugx_prof.so,
#  loaded instead of the product library when AUGX_LIB names it; the product is built without them)
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result
CXXFLAGS = -O2 -std=c++17 -fPIC -ffp-contract=off
SRC     = augustus_amd/csrc
HOSTSRC = $(SRC)/model.cc $(SRC)/capi_model.cc $(SRC)/genes.cc $(SRC)/driver.cc $(SRC)/sharded.cc
DEVHDR  = $(SRC)/device/assmemo.h $(SRC)/device/dp.h $(SRC)/device/kernels.h $(SRC)/device/dense.h $(SRC)/device/layout.h $(SRC)/device/sampler.h $(SRC)/device/snipmemo.h $(SRC)/device/launch.h
HOSTHDR = $(SRC)/model.h $(SRC)/genes.h $(SRC)/capi_internal.h include/augx.h

all: product oracle emu
product: augustus_amd/libaugx.so augustus_amd/bin/augustus

# The heavy kernel families are translation units of their own, one object per block size (device/launch.h), so that `make -j`
# compiles them side by side (the whole library: ~1.5 min on 8 cores instead of ~4.5) and a kernel change rebuilds one family.
OBJ     = build/obj
FAMILIES = trellis cand forward dense
KOBJS   = $(foreach f,$(FAMILIES),$(foreach b,8 4 2,$(OBJ)/k_$(f)_$(b).o))
HOSTOBJS = $(patsubst $(SRC)/%.cc,$(OBJ)/%.o,$(HOSTSRC))
PRODFLAGS =

define KRULE
$(OBJ)/k_$(1)_$(2).o: $(SRC)/device/k_$(1).hip $(DEVHDR) include/augx.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) $$(PRODFLAGS) -DAUGX_TU_BLK=$(2) -c -o $$@ $$<
endef
$(foreach f,$(FAMILIES),$(foreach b,8 4 2,$(eval $(call KRULE,$(f),$(b)))))

$(OBJ)/decoder.o: $(SRC)/device/decoder.hip $(DEVHDR) $(HOSTHDR)
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) $(PRODFLAGS) -c -o $@ $<
$(OBJ)/%.o: $(SRC)/%.cc $(HOSTHDR) $(SRC)/device/layout.h $(SRC)/device/dp.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) $(PRODFLAGS) -c -o $@ $<

augustus_amd/libaugx.so: $(HOSTOBJS) $(OBJ)/decoder.o $(KOBJS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $^

# developer builds, objects of their own, loaded instead of the product library when AUGX_LIB names them:
#   make prof                                   cycle counters of the trellis wavefronts (AUGX_PROF=1) -> augustus_amd/libaugx_prof.so
#   make variant NAME=x FLAGS="-DAUGX_..."      any other set of build-time switches             -> augustus_amd/libaugx_x.so
prof:
	$(MAKE) variant NAME=prof FLAGS=-DAUGX_PROFILE
variant:
	$(MAKE) OBJ=build/obj_$(NAME) PRODFLAGS="$(FLAGS)" build/obj_$(NAME)/libaugx_variant.so
	cp build/obj_$(NAME)/libaugx_variant.so augustus_amd/libaugx_$(NAME).so
$(OBJ)/libaugx_variant.so: $(HOSTOBJS) $(OBJ)/decoder.o $(KOBJS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $^

augustus_amd/bin/augustus: $(SRC)/augustus_main.cc augustus_amd/libaugx.so
	@mkdir -p augustus_amd/bin
	$(HIPCC) -O2 -o $@ $(SRC)/augustus_main.cc -Laugustus_amd -laugx -Wl,-rpath,'$$ORIGIN/..'

# ---- test infrastructure (never linked into the product) ----
oracle: oracle/libghmm_twin.so
oracle/libghmm_twin.so: oracle/ghmm_twin.cc include/augx.h
	$(CXX) $(CXXFLAGS) -shared -o $@ oracle/ghmm_twin.cc

emu: build/libaugx_emu.so build/libaugx_emu_pl1.so
build/libaugx_emu.so: tests/emu/emu.cc $(DEVHDR) include/augx.h
	@mkdir -p build
	$(CXX) $(CXXFLAGS) -shared -o $@ tests/emu/emu.cc
# the same emulator with one plane of transition terms in (emulated) LDS: pieces with two GC classes then take the path
# that pieces with more than eight take in the product
build/libaugx_emu_pl1.so: tests/emu/emu.cc $(DEVHDR) include/augx.h
	@mkdir -p build
	$(CXX) $(CXXFLAGS) -DAUGX_MAXPL_LDS=1 -shared -o $@ tests/emu/emu.cc

ref:
	$(MAKE) -C oracle -j8

clean:
	rm -rf build augustus_amd/libaugx.so augustus_amd/bin oracle/libghmm_twin.so
.PHONY: all product prof variant oracle emu ref clean
