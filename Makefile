# Build of the MI355X render path (libc2rt.so) and of the CPU oracle
# (oracle/libc2rt_oracle.so, test infrastructure).  `make -j8`.
#
# Arithmetic flags are part of the contract: no contraction (the reference's
# x86-64 code has separate mul/add), IEEE fp32 divide/sqrt, no fast-math.
HIPCC      ?= hipcc
CC         ?= gcc
ARCH       ?= gfx950
FPFLAGS    := -ffp-contract=off -fno-fast-math
HIPFLAGS   := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC $(FPFLAGS) -fhip-fp32-correctly-rounded-divide-sqrt \
              -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical $(EXTRA_HIPFLAGS)
# Kernel units only.  Machine LICM hoists loop-invariant scalar and vector values out of the node / CSG
# stepping loops of a kernel that is already at its register budget: more values live across the loops,
# more SGPRs spilled to VGPR lanes.  Measured with it off (profiles/r02_variants.md): lecture5 4K 1 tap
# 0.389 -> 0.375 ms, zaphod DOF 5.23 -> 5.04 ms, depth-4 VGPR spills 90 -> 60.
# -phi-node-folding-threshold=4 (round 4; LLVM's default is 2): SimplifyCFG turns slightly larger two-way branches into
# selects.  The only one of ~25 code-generation options screened that helps (profiles/r04_variants.md step 8): the
# headline instance drops from 122 to 118 VGPRs and from 151 to 135 SGPRs spilled to VGPR lanes; lecture5 4K x5
# 1.016 -> 0.997 ms, 4K x1 0.255 -> 0.249, 8K x4 3.20 -> 3.14, 1080p 0.078 -> 0.077; planes-only and depth-4
# instances unchanged.  (6 / 8 / 16: 133 spilled, a little slower than 4; 3: no change.)  Speculating a few more
# side-effect-free fp64 operations does not change a bit of any frame (the suite is bit-identical either way).
KERNELFLAGS := -mllvm -disable-machine-licm -mllvm -phi-node-folding-threshold=4
# Per unit (= CSG nesting depth of the frame-kernel instances in it), on top of KERNELFLAGS: GVN's partial-redundancy
# elimination off and SimplifyCFG's bonus-instruction threshold at 4 (profiles/r04_variants.md step 10, same-call
# A/B): planes-only / depth-0 instances -1.1 ... -2.5 % (zaphod DOF 3.947 -> 3.905 ms, zaphod x4 0.483 -> 0.472,
# lecture4 1080p 134.4 -> 138.0 Gray/s), csg_stress cut to depth 2 / 3 / 4: 1.715 -> 1.664, 3.735 -> 3.50, 8.29 ->
# 8.14 ms.  The depth-1 instances (the headline) were 0.6 % slower with them at the time and went without; measured
# again after steps 12 - 17 had changed what those instances keep in registers they gain 1.9 % (0.925 -> 0.908 ms,
# 1080p +2.5 %; the threshold alone: 1.6 %) and the other depths still want theirs (step 18).  Kept per unit: the
# answer has changed once already.  Depth 4 also takes the scheduler's max-memory-clause strategy: those instances are
# latency-bound (VALU busy 0.64, 43 % of the wave-cycles waiting for memory) and gain 2.3 % from it (csg_stress 7.80 ->
# 7.62 ms; the other strategies 1.2 %); depth 3 LOSES 2.2 % with it, depth 2 and the issue-bound depths 0 / 1 do not
# care for it (step 20); depth 1 takes iterative-ilp, worth 0.6 % on the headline frame; depth 2 iterative-ilp as well (-3 % on csg_stress cut
# to depth 2); depth 0 and depth 3 keep the default (depth 0: max-ilp gains 1.8 % on lecture4 1080p but puts 32 B of
# scratch into the depth-of-field planes instance; depth 3: every other strategy is slower).
KERNELFLAGS_u0 := -mllvm -enable-pre=false -mllvm -bonus-inst-threshold=4
KERNELFLAGS_u1 := -mllvm -enable-pre=false -mllvm -bonus-inst-threshold=4 -mllvm -amdgpu-sched-strategy=iterative-ilp
KERNELFLAGS_u2 := -mllvm -enable-pre=false -mllvm -bonus-inst-threshold=4 -mllvm -amdgpu-sched-strategy=iterative-ilp
KERNELFLAGS_u3 := -mllvm -enable-pre=false -mllvm -bonus-inst-threshold=4
KERNELFLAGS_u4 := -mllvm -enable-pre=false -mllvm -bonus-inst-threshold=4 -mllvm -amdgpu-sched-strategy=max-memory-clause
KERNELFLAGS_u5 :=
CXXFLAGS   := -O2 -std=c++17 -fPIC $(FPFLAGS) -Wall -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include
CSRC       := chess2rt_amd/csrc
# development knob: `make VARIANT=name EXTRA_HIPFLAGS=... EXTRA_KERNEL_FLAGS=...` builds chess2rt_amd/libc2rt_name.so
# (EXTRA_KERNEL_FLAGS reach only the hipcc kernel units, e.g. -mllvm options)
VARIANT    ?=
BUILD      := build$(if $(VARIANT),_$(VARIANT))
LIBNAME    := chess2rt_amd/libc2rt$(if $(VARIANT),_$(VARIANT)).so

UNITS      := 0 1 2 3 4 5
KOBJS      := $(foreach u,$(UNITS),$(BUILD)/c2rt_kernels_u$(u).o)
HOBJS      := $(BUILD)/c2rt_api.o $(BUILD)/dsc.o $(BUILD)/scene.o $(BUILD)/host_api.o

# diagnostics build of the same library: c2rt_api.cpp with the environment hooks compiled in (-DC2RT_DIAG=1), linked
# over the SAME kernel objects; tests and scripts that need a hook load it with C2RT_LIB_VARIANT=diag
DIAGNAME   := chess2rt_amd/libc2rt_diag.so

all: $(LIBNAME) $(if $(VARIANT),,$(DIAGNAME)) oracle/libc2rt_oracle.so oracle/libc2rt_oracle_count.so tests/fp64_lean_check

$(BUILD):
	mkdir -p $(BUILD)

# (the Makefile is a prerequisite: the arithmetic and code-generation flags are part of what a kernel object is)
$(BUILD)/c2rt_kernels_u%.o: $(CSRC)/c2rt_kernels.hip $(CSRC)/c2rt_trace.inc $(CSRC)/c2rt_device.h $(CSRC)/x87.h $(CSRC)/fp64_lean.h include/c2rt.h Makefile | $(BUILD)
	$(HIPCC) $(HIPFLAGS) $(KERNELFLAGS) $(KERNELFLAGS_u$*) $(EXTRA_KERNEL_FLAGS) -DC2RT_UNIT=$* -c $< -o $@

$(BUILD)/c2rt_api.o: $(CSRC)/c2rt_api.cpp $(CSRC)/c2rt_device.h include/c2rt.h | $(BUILD)
	g++ $(CXXFLAGS) $(EXTRA_HIPFLAGS) -c $< -o $@

$(BUILD)/%.o: $(CSRC)/host/%.cpp $(CSRC)/host/scene.hpp $(CSRC)/host/dsc.hpp include/c2rt.h include/c2rt_host.h | $(BUILD)
	g++ $(CXXFLAGS) -c $< -o $@

$(LIBNAME): $(KOBJS) $(HOBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -lpthread

$(BUILD)/c2rt_api_diag.o: $(CSRC)/c2rt_api.cpp $(CSRC)/c2rt_device.h include/c2rt.h | $(BUILD)
	g++ $(CXXFLAGS) $(EXTRA_HIPFLAGS) -DC2RT_DIAG=1 -c $< -o $@

$(DIAGNAME): $(KOBJS) $(BUILD)/c2rt_api_diag.o $(filter-out $(BUILD)/c2rt_api.o,$(HOBJS))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -lpthread

# device check of fp64_lean.h against the compiler's own divide / sqrt expansions (tests/test_gpu_parity.py runs it)
tests/fp64_lean_check: tests/fp64_lean_check.hip $(CSRC)/fp64_lean.h
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 $(FPFLAGS) -fhip-fp32-correctly-rounded-divide-sqrt $< -o $@

# CPU oracle: plain C restatement of the reference algorithm (tests only)
oracle/libc2rt_oracle.so: oracle/c2rt_oracle.c oracle/c2rt_oracle.h include/c2rt.h
	$(CC) -O2 -std=gnu11 -fPIC -shared $(FPFLAGS) -Wall -o $@ oracle/c2rt_oracle.c -lm -lpthread

# the same restatement with its arithmetic statements tallied (algorithmic fp op count for bench.py's roofline.flops)
oracle/libc2rt_oracle_count.so: oracle/c2rt_oracle.c oracle/c2rt_oracle.h include/c2rt.h
	$(CC) -O2 -std=gnu11 -fPIC -shared $(FPFLAGS) -Wall -DORC_COUNT_OPS -o $@ oracle/c2rt_oracle.c -lm -lpthread

# the compiler's own per-kernel register / scratch / occupancy figures (committed as profiles/rNN_resource_usage.txt:
# rocprofv3's VGPR_Count column is in allocation granules and does not show them)
resource-usage: | $(BUILD)
	@$(foreach u,$(UNITS),$(HIPCC) $(HIPFLAGS) $(KERNELFLAGS) $(KERNELFLAGS_u$(u)) $(EXTRA_KERNEL_FLAGS) -DC2RT_UNIT=$(u) -Rpass-analysis=kernel-resource-usage \
	    -c $(CSRC)/c2rt_kernels.hip -o $(BUILD)/ru_u$(u).o 2>&1 | grep -E "remark:" | sed -e 's/.*remark: [^ ]* *//' -e 's/ \[-Rpass-analysis=kernel-resource-usage\]//' ;)

clean:
	rm -rf build build_* chess2rt_amd/libc2rt*.so oracle/libc2rt_oracle.so oracle/libc2rt_oracle_count.so tests/fp64_lean_check

.PHONY: all clean resource-usage
