# Builds libtecount.so (HIP, gfx950 only) in-tree and the C oracle (test infrastructure).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := $(wildcard te_counter_amd/csrc/*.hip)
HDRS  := $(wildcard te_counter_amd/csrc/*.h) include/tecount.h Makefile
OBJS  := $(patsubst te_counter_amd/csrc/%.hip,build/%.o,$(CSRC))
LIB   := te_counter_amd/libtecount.so
# packed FP32 instructions are switched off for the device pass (hipcc's host pass does not know the feature and says so)
FLAGS := $(EXTRA) -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Wall -Wno-unused-result -Iinclude \
         -Xclang -target-feature -Xclang -packed-fp32-ops

all: $(LIB) oracle

build/%.o: te_counter_amd/csrc/%.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(FLAGS) -c -o $@ $<

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -lz -lpthread

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf build $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
