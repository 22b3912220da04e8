# rrt-mi355x build.  Everything is built IN-TREE (the .so / binaries are git-ignored but travel to
# the GPU box with the gpurun snapshot).
#
#   make            -> rrt_amd/librrtx.so (C ABI + HIP kernels for gfx950), rrt, rrtd (CLI)
#   make oracle     -> oracle/librrt_oracle.so (+ oracle/_ref when /root/reference is present)
#   make all        -> both
HIPCC     ?= /opt/rocm/bin/hipcc
CXX       ?= g++
ARCH      ?= gfx950
CSRC      := rrt_amd/csrc
# -ffp-contract=off: the reference image depends on unfused fp32 rounding (DESIGN.md "Numerics").
# -fno-slp-vectorize: keeps the primitive scan on plain v_mul/v_add (see DESIGN.md "Kernel").
DEVFLAGS  ?= -O3 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form
KFLAGS    := --offload-arch=$(ARCH) $(DEVFLAGS) -fPIC -std=c++17
HOSTFLAGS := -O2 -ffp-contract=off -fPIC -std=c++17 -Wall

LIB       := rrt_amd/librrtx.so
OBJS      := $(CSRC)/rrtx_kernels.o $(CSRC)/rrtx_api.o $(CSRC)/rrtx_group.o $(CSRC)/host_scene.o $(CSRC)/host_image.o

default: $(LIB) rrt rrtd

all: default oracle

$(CSRC)/rrtx_kernels.o: $(CSRC)/rrtx_kernels.hip $(CSRC)/rrtx_device.h $(CSRC)/rrtx_path.h $(CSRC)/rrtx_wave.h $(CSRC)/rrtx_launch.h
	$(HIPCC) $(KFLAGS) -c $< -o $@

$(CSRC)/rrtx_api.o: $(CSRC)/rrtx_api.cpp $(CSRC)/rrtx_device.h $(CSRC)/rrtx_grid.h $(CSRC)/rrtx_pack.h $(CSRC)/rrtx_launch.h include/rrtx.h
	$(HIPCC) $(HOSTFLAGS) -c $< -o $@

# (RCCL: the header only - librccl.so is dlopen()ed by the first rrtx_group_create)
$(CSRC)/rrtx_group.o: $(CSRC)/rrtx_group.cpp $(CSRC)/rrtx_device.h $(CSRC)/rrtx_launch.h include/rrtx.h
	$(HIPCC) $(HOSTFLAGS) -c $< -o $@

$(CSRC)/host_scene.o: $(CSRC)/host_scene.cpp include/rrtx.h
	$(CXX) $(HOSTFLAGS) -c $< -o $@

$(CSRC)/host_image.o: $(CSRC)/host_image.cpp include/rrtx.h
	$(CXX) $(HOSTFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -lz -ldl

# drop-in binaries: `rrt` (float) and `rrtd` (double) — same source, precision chosen by name
rrt: $(CSRC)/rrt_main.cpp $(LIB) include/rrtx.h
	$(CXX) $(HOSTFLAGS) -fPIE -pthread $< -o $@ -Lrrt_amd -lrrtx -Wl,-rpath,'$$ORIGIN/rrt_amd'

rrtd: $(CSRC)/rrt_main.cpp $(LIB) include/rrtx.h
	$(CXX) $(HOSTFLAGS) -fPIE -pthread -DRRTX_DOUBLE $< -o $@ -Lrrt_amd -lrrtx -Wl,-rpath,'$$ORIGIN/rrt_amd'

oracle:
	$(MAKE) -C oracle all ref

clean:
	rm -f $(OBJS) $(LIB) rrt rrtd
	$(MAKE) -C oracle clean

.PHONY: default all oracle clean
