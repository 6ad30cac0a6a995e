# Builds libsgf.so (the gfx950 C-ABI library).  hipcc cross-compiles without a GPU.  (The oracle is Python / numpy.)
HIPCC ?= hipcc
ARCH ?= gfx950
HIPFLAGS ?= --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function \
            -fhip-fp32-correctly-rounded-divide-sqrt
# make PROBES=1 BUILD=build_probes LIB=sgformer_amd/lib/libsgf_probes.so: also compile the timing-ablation variants (SGF_*_DEBUG masks; results are then wrong) the probe scripts use
ifdef PROBES
HIPFLAGS += -DSGF_PROBES
endif
CSRC := sgformer_amd/csrc
SRCS := $(CSRC)/capi.hip $(CSRC)/csr.hip $(CSRC)/spmm.hip $(CSRC)/attn.hip $(CSRC)/fused.hip $(CSRC)/subgraph.hip $(CSRC)/reorder.hip $(CSRC)/spmm_plan.hip $(CSRC)/prologue.hip $(CSRC)/head.hip $(CSRC)/rowgemm.hip $(CSRC)/spmm_tile.hip $(CSRC)/spmm_pack.hip $(CSRC)/sampler.hip $(CSRC)/linear_f32.hip $(CSRC)/gemm.hip $(CSRC)/attn_small.hip $(CSRC)/comm.hip $(CSRC)/subgraph_csr.hip $(CSRC)/gramx.hip
BUILD ?= build
OBJS := $(patsubst $(CSRC)/%.hip,$(BUILD)/%.o,$(SRCS))
LIB  ?= sgformer_amd/lib/libsgf.so

all: $(LIB)

$(LIB): $(OBJS)
	@mkdir -p $(dir $@)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl

$(BUILD)/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/spmm_shared.h $(CSRC)/reduce_shared.h include/sgf.h
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

clean:
	rm -rf build build_probes $(LIB) sgformer_amd/lib/libsgf_probes.so oracle/_build oracle/_ref

.PHONY: all clean
