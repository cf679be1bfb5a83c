# Builds libjsg.so (hand-written HIP for gfx950 + C-ABI) without Python; the same commands as
# jadespectrogram_amd/_build.py.  `make test-cpp` builds the C++ drop-in test driver against it.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
SRC   := jadespectrogram_amd/csrc
OUT   := jadespectrogram_amd/libjsg.so
OBJ   := jadespectrogram_amd/build

.PHONY: lib oracle test-cpp clean
lib: $(OUT)

HIPFLAGS := -std=c++17 -O3 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden -Iinclude --offload-arch=$(ARCH) -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=16
KDEPS    := $(SRC)/jsg_stft_kernel.h $(SRC)/jsg_internal.h $(SRC)/jsg_exact_math.h include/jsg.h

$(OBJ)/jsg_kernels.o: $(SRC)/jsg_kernels.hip $(KDEPS)
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
# the 512 / 1024 / 2048 / 8192-point kernels: ILP-first machine scheduler (see the unit's header comment)
$(OBJ)/jsg_stft_a.o: $(SRC)/jsg_stft_a.hip $(KDEPS)
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -mllvm -amdgpu-sched-strategy=max-ilp -c $< -o $@
$(OBJ)/jsg_stft_b.o: $(SRC)/jsg_stft_b.hip $(KDEPS)
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJ)/%.o: $(SRC)/%.cpp $(SRC)/jsg_internal.h $(SRC)/jsg_block_queue.h $(SRC)/jsg_exact_math.h $(SRC)/jsg_colormap_tables.inc include/jsg.h
	@mkdir -p $(OBJ)
	$(HIPCC) -std=c++17 -O3 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden -Iinclude -ffp-contract=off -D__HIP_PLATFORM_AMD__ -c $< -o $@

$(OUT): $(OBJ)/jsg_kernels.o $(OBJ)/jsg_stft_a.o $(OBJ)/jsg_stft_b.o $(OBJ)/jsg_engine.o $(OBJ)/jsg_host_math.o
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $^

oracle:
	$(MAKE) -C oracle port ref

test-cpp: lib
	g++ -std=c++17 -O1 -Wall -Wextra -Iinclude tests/cpp/host_dropin_test.cpp -o $(OBJ)/host_dropin_test \
	    -Ljadespectrogram_amd -ljsg -Wl,-rpath,$(CURDIR)/jadespectrogram_amd

clean:
	rm -rf $(OBJ) $(OUT)
