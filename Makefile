# Build the C-ABI HIP library (cross-compiles for gfx950 without a GPU).  One object per source so that a
# kernel edit recompiles one file; `make -j` compiles them in parallel.
HIPCC ?= hipcc
ARCH  ?= gfx950
CSRC  := $(wildcard srl_amd/csrc/*.hip)
HDRS  := $(wildcard srl_amd/csrc/*.h) include/srl_hip.h
OBJD  := srl_amd/csrc/build
OBJS  := $(patsubst srl_amd/csrc/%.hip,$(OBJD)/%.o,$(CSRC))
LIB   := srl_amd/csrc/libsrlhip.so
FLAGS := -O3 --offload-arch=$(ARCH) -std=c++17 -fPIC

all: $(LIB)

$(OBJD)/%.o: srl_amd/csrc/%.hip $(HDRS)
	@mkdir -p $(OBJD)
	$(HIPCC) $(FLAGS) -c -o $@ $<

$(LIB): $(OBJS)
	$(HIPCC) $(FLAGS) -shared -o $@ $(OBJS)

clean:
	rm -rf $(LIB) $(OBJD)

.PHONY: all clean
