# Build the C-ABI HIP library (cross-compiles for gfx950 without a GPU) and the oracle's C pieces.
HIPCC ?= hipcc
ARCH  ?= gfx950
CSRC  := $(wildcard srl_amd/csrc/*.hip)
HDRS  := $(wildcard srl_amd/csrc/*.h) include/srl_hip.h
LIB   := srl_amd/csrc/libsrlhip.so

all: $(LIB)

$(LIB): $(CSRC) $(HDRS)
	$(HIPCC) -O3 --offload-arch=$(ARCH) -std=c++17 -fPIC -shared -o $@ $(CSRC)

clean:
	rm -f $(LIB)

.PHONY: all clean
