# Builds the C-ABI kernel library for MI355X (gfx950) and the oracle's C parts (none yet).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
SRC   := $(wildcard vqa_amd/csrc/*.hip)
OBJ   := $(SRC:.hip=.o)
HDR   := $(wildcard vqa_amd/csrc/*.h) include/vqa_hip.h
FLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Iinclude -Ivqa_amd/csrc -Wall -Wno-unused-function

all: vqa_amd/libvqa_hip.so

%.o: %.hip $(HDR)
	$(HIPCC) $(FLAGS) -c $< -o $@

vqa_amd/libvqa_hip.so: $(OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ)

clean:
	rm -f $(OBJ) vqa_amd/libvqa_hip.so

.PHONY: all clean

# Development build for A/B and probe runs on the GPU box: environment switches (VQA_GEMM_NT, VQA_GEMM_PP) and,
# with PP_PROBE=1..4, the timing probes of the eight-phase GEMM (wrong results by construction).  Never loaded by
# the package unless VQA_HIP_LIB points at it.
DEVDIR ?= build/dev
DEVOBJ := $(patsubst vqa_amd/csrc/%.hip,$(DEVDIR)/%.o,$(SRC))
DEVFLAGS := -DVQA_DEV $(if $(PP_PROBE),-DPP_PROBE=$(PP_PROBE)) $(EXTRA)
DEVLIB ?= vqa_amd/libvqa_hip_dev.so

$(DEVDIR)/%.o: vqa_amd/csrc/%.hip $(HDR)
	@mkdir -p $(DEVDIR)
	$(HIPCC) $(FLAGS) $(DEVFLAGS) -c $< -o $@

dev: $(DEVOBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $(DEVLIB) $(DEVOBJ)

.PHONY: dev
