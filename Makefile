# Top-level build: the product (librt_hip.so: HIP kernels + C ABI; librt_host.so: C++ host side) and the
# test oracle (oracle/liboracle.so).  Everything is built IN-TREE so that the .so files travel with gpurun.
#
# The kernels are built for gfx950 only, with contraction OFF: fused multiply-adds appear only where the source
# writes them (arithmetic contract v1); sqrt and division stay correctly rounded (hipcc default).

HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
LIBDIR   := rt_amd/lib
HIPFLAGS := --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wextra -Wno-unused-parameter
HOSTFLAGS:= -std=c++20 -O2 -fPIC -Wall -Wextra

HIPFLAGS += -fvisibility=hidden
# No SLP vectorisation: it pairs the kernels' float arithmetic into v_pk_fma/mul/add_f32, which issue at half rate on this chip
# (two of them cost what four plain instructions do, profiles/r04/valu_issue_costs.txt) and need their operands in even-aligned
# register pairs — moves, and a higher register count.  Measured on the contract-v4 kernels: headline 2.28 -> 2.14 ms,
# basic + plane 2.74 -> 2.62, tilted camera 2.80 -> 2.62 (profiles/r05/codegen_ab.txt).
HIPFLAGS += -fno-slp-vectorize
API_UNITS := context scene frame render multi group
HIP_HDR  := rt_amd/csrc/kernels.hpp rt_amd/csrc/contract.hpp rt_amd/csrc/scan.hpp rt_amd/csrc/frame_group.hpp rt_amd/csrc/delivery.hpp rt_amd/csrc/internal.hpp include/rt_hip.h
HOST_SRC := rt_amd/host/host_capi.cpp rt_amd/host/scene.cpp rt_amd/host/toml_subset.cpp
HOST_HDR := $(wildcard rt_amd/host/*.hpp) rt_amd/host/host_capi.h rt_amd/host/named_colours.inc include/rt_hip.h

all: $(LIBDIR)/librt_hip.so $(LIBDIR)/librt_hip_kat.so $(LIBDIR)/librt_host.so rt_amd/bin/rt_headless oracle

# kernels.hip is compiled twice: the parity contract (contraction off), and RT_HIP_FLAG_FAST's arithmetic
# (-DRT_HIP_FAST_BUILD -ffp-contract=fast: only launch_render_fast comes out of that one)
FASTFLAGS := $(filter-out -ffp-contract=off,$(HIPFLAGS)) -ffp-contract=fast -DRT_HIP_FAST_BUILD
OBJDIR   := build/obj$(NAME)
HIP_OBJS := $(OBJDIR)/kernels.o $(OBJDIR)/kernels_fast.o $(API_UNITS:%=$(OBJDIR)/%.o) $(OBJDIR)/delivery.o

$(OBJDIR)/kernels.o: rt_amd/csrc/kernels.hip $(HIP_HDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(DEFS) -c $< -o $@
$(OBJDIR)/kernels_fast.o: rt_amd/csrc/kernels.hip $(HIP_HDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(FASTFLAGS) $(DEFS) -c $< -o $@
$(OBJDIR)/%.o: rt_amd/csrc/%.hip $(HIP_HDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(DEFS) -c $< -o $@
# the pixel carrier is plain C++17 (no HIP): the same source is built into tests/native/pixel_carrier_test on the CPU
$(OBJDIR)/delivery.o: rt_amd/csrc/delivery.cpp rt_amd/csrc/delivery.hpp
	@mkdir -p $(OBJDIR)
	$(CXX) -std=c++17 -O2 -fPIC -fvisibility=hidden -Wall -Wextra -c $< -o $@

# the product: exports the C entry points of include/rt_hip.h and nothing else
$(LIBDIR)/librt_hip.so: $(HIP_OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=gfx950 -fPIC -shared -o $@ $(HIP_OBJS) -L/opt/rocm/lib -lrccl -lpthread
	python3 tools/kernel_sources_hash.py > $(LIBDIR)/librt_hip.kernels.sha16   # what THIS binary's kernels were compiled from (bench.py, profiles)

# test-only: the known-answer entry points of include/rt_hip_kat.h (never shipped; loads next to librt_hip.so)
$(LIBDIR)/librt_hip_kat.so: $(OBJDIR)/kat.o $(LIBDIR)/librt_hip.so
	$(HIPCC) --offload-arch=gfx950 -fPIC -shared -o $@ $(OBJDIR)/kat.o -L$(LIBDIR) -lrt_hip -Wl,-rpath,'$$ORIGIN' -L/opt/rocm/lib -lrccl
$(OBJDIR)/kat.o: include/rt_hip_kat.h

$(LIBDIR)/librt_host.so: $(HOST_SRC) $(HOST_HDR)
	@mkdir -p $(LIBDIR)
	$(CXX) $(HOSTFLAGS) -shared -o $@ $(HOST_SRC)

# experiment builds for tools/gpu_ab.py: make variant NAME=x DEFS="-DRT_HIP_SOMETHING=1" -> rt_amd/lib/librt_hip_x.so
# (objects under build/obj<NAME>; an experiment library carries the known-answer and debug entry points itself)
variant: $(HIP_OBJS) $(OBJDIR)/kat.o
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=gfx950 -fPIC -shared -o $(LIBDIR)/librt_hip_$(NAME).so $(HIP_OBJS) $(OBJDIR)/kat.o -L/opt/rocm/lib -lrccl -lpthread

# windowless driver: the registry, the hip_ray_tracer plug-in and the scene loader, linked against the C ABI only
HEADLESS_SRC := rt_amd/host/main.cpp rt_amd/host/hip_ray_tracer.cpp rt_amd/host/scene.cpp rt_amd/host/toml_subset.cpp
rt_amd/bin/rt_headless: $(HEADLESS_SRC) $(HOST_HDR) $(LIBDIR)/librt_hip.so
	@mkdir -p rt_amd/bin
	$(CXX) $(HOSTFLAGS) -o $@ $(HEADLESS_SRC) -L$(LIBDIR) -lrt_hip -Wl,-rpath,'$$ORIGIN/../lib'

oracle:
	$(MAKE) -C oracle

# tests/native/soagen_columns.cpp against the reference's own vendored container runtime, compiled from where it lies under
# /root/reference (nothing is copied; the reference tree exists in the build container only).  The binary goes to oracle/_ref/:
# git-ignored, not gpurun-ignored — the GPU box runs it with --gpu (tests/test_soagen_columns.py).
SOAGEN_DIR := /root/reference/vendor
ifneq ($(wildcard $(SOAGEN_DIR)/soagen.hpp),)
all: oracle/_ref/soagen_columns
oracle/_ref/soagen_columns: tests/native/soagen_columns.cpp $(LIBDIR)/librt_hip.so oracle include/rt_hip.h oracle/cpu_ref.h
	@mkdir -p oracle/_ref
	$(CXX) -std=c++20 -O1 -Wall -Wextra -I$(SOAGEN_DIR) -Iinclude -Ioracle $< -o $@ -L$(LIBDIR) -lrt_hip -Loracle -loracle -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)' -Wl,-rpath,'$$ORIGIN/..'
endif

clean:
	rm -f $(LIBDIR)/*.so rt_amd/bin/rt_headless
	$(MAKE) -C oracle clean

.PHONY: all oracle clean variant
