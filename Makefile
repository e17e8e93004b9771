# Convenience targets; __graft_entry__.build() does the same from Python.
HIPCC ?= /opt/rocm/bin/hipcc
HIPFLAGS := --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -mllvm -disable-vector-combine
CSRC := putslam_amd/csrc
LIB := putslam_amd/libputslam_hip.so
DROPIN := putslam_amd/libputslam_dropin.so
SHARD := putslam_amd/libputslam_shard.so

all: $(LIB) $(DROPIN) $(SHARD) oracle

# (the recipe -- one device translation unit + host-only ones -- lives in putslam_amd/_build.py)
$(LIB): $(wildcard $(CSRC)/*.hip $(CSRC)/*.h $(CSRC)/*.cpp) include/putslam_hip.h
	python -c "from putslam_amd import _build; _build.build_hip()"

$(DROPIN): $(CSRC)/dropin/putslam_dropin.cpp $(CSRC)/dropin/putslam_dropin.h $(CSRC)/dropin/putslam_compat_types.h $(LIB)
	g++ -O2 -std=c++17 -fPIC -shared -Wall -Iinclude -I$(CSRC)/dropin $< -o $@ -Lputslam_amd -lputslam_hip '-Wl,-rpath,$$ORIGIN'

$(SHARD): $(CSRC)/ps_shard.hip include/putslam_shard.h include/putslam_hip.h $(LIB)
	$(HIPCC) --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -shared -Iinclude $< -o $@ -Lputslam_amd -lputslam_hip -L/opt/rocm/lib -lrccl '-Wl,-rpath,$$ORIGIN' -Wl,-rpath,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle

test-cpu: all
	python -m pytest tests -q -m "not gpu"

test-gpu: all
	python -m pytest tests -q -m gpu

bench: all
	python bench.py

clean:
	rm -f $(LIB) $(DROPIN) $(SHARD) tests/cpp/test_dropin tests/cpp/test_reference_shaped demos/cpp/demo_matching demos/cpp/demo_sequences_multi_gpu demos/cpp/demo_latency
	$(MAKE) -C oracle clean

.PHONY: all oracle test-cpu test-gpu bench clean
