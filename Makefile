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
