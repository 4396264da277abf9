# Builds libtecount.so (HIP, gfx950 only) in-tree and the C oracle (test infrastructure).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := $(wildcard te_counter_amd/csrc/*.hip)
HDRS  := $(wildcard te_counter_amd/csrc/*.h) include/tecount.h
LIB   := te_counter_amd/libtecount.so

all: $(LIB) oracle

$(LIB): $(CSRC) $(HDRS)
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -Wall -Wno-unused-result \
	    -Iinclude -o $@ $(CSRC) -lz -lpthread

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
