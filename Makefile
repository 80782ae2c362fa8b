# Top-level build: the product library (HIP, gfx950 only) and the CPU checker.
#   make            -> plaac_amd/libplaac_native.so + oracle/libplaac_oracle.so (+ bin/plaac when its source exists)
# -ffp-contract=off is part of the numerical contract: never remove it.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result
CSRC      = plaac_amd/csrc
LIB       = plaac_amd/libplaac_native.so

# make DIAG=1 (or `make diag`): the DIAGNOSTIC build, plaac_amd/libplaac_native_diag.so, loaded through PLAAC_NATIVE_LIB - the
# forms that were measured slower and are no longer selected (k_fwd_direct, k_bwd_fwd_post, the lane-store forms of the
# forward pass with the posteriors, the launch-order and segment switches: EXPERIMENTS.md) and the result-breaking ablation
# switches (DEBUG_SKIP / _SKIP_FROM: named kernels are not launched, rows stale; VIT_STOP; DEBUG_COUNTER), all read from the
# environment as PLAAC_<KEY>. The release library contains neither the kernels nor the names (include/plaac_native.h,
# "environment and test hooks").
DIAGLIB   = plaac_amd/libplaac_native_diag.so
ifeq ($(DIAG),1)
all: $(DIAGLIB)
endif

all: $(LIB) oracle $(if $(wildcard $(CSRC)/plaac_cli.cpp),cli) $(if $(JNI_H),jni)

# One object per source, so that `make -j` compiles the two device units (plaac_kernels.hip: 60 s, plaac_kernels_lat.hip: 20 s)
# and the host sources side by side; build() runs `make -j4 all`.
DEPS      = $(wildcard $(CSRC)/*.hip.inc) $(wildcard include/*.h)
HOSTSRC   = plaac_host plaac_node $(if $(wildcard $(CSRC)/plaac_io.cpp),plaac_io)
OBJS      = build/plaac_kernels.o build/plaac_kernels_lat.o $(patsubst %,build/%.o,$(HOSTSRC))
DIAGOBJS  = build/diag/plaac_kernels.o build/diag/plaac_kernels_lat.o $(patsubst %,build/%.o,$(HOSTSRC))
# The latency-form chain kernels a second time, as a unit of their own under the compiler's max-ilp instruction scheduling
# (plaac_kernels_lat.hip: chain-bound calls 6 % faster with it, the throughput-bound headline 3.7 % slower - so only there)
LATFLAGS  = -Wno-unused-function -mllvm -amdgpu-sched-strategy=max-ilp

build/plaac_kernels.o: $(CSRC)/plaac_kernels.hip $(DEPS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -Iinclude -c -o $@ $<
build/plaac_kernels_lat.o: $(CSRC)/plaac_kernels_lat.hip $(DEPS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) $(LATFLAGS) -Iinclude -c -o $@ $<
build/diag/plaac_kernels.o: $(CSRC)/plaac_kernels.hip $(DEPS)
	@mkdir -p build/diag
	$(HIPCC) $(HIPFLAGS) -DPLAAC_DIAG=1 -Iinclude -c -o $@ $<
build/diag/plaac_kernels_lat.o: $(CSRC)/plaac_kernels_lat.hip $(DEPS)
	@mkdir -p build/diag
	$(HIPCC) $(HIPFLAGS) $(LATFLAGS) -DPLAAC_DIAG=1 -Iinclude -c -o $@ $<
build/%.o: $(CSRC)/%.cpp $(wildcard include/*.h)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -Iinclude -c -o $@ $<

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -fPIC -shared -o $@ $(OBJS) -Wl,-rpath,/opt/rocm/lib

$(DIAGLIB): $(DIAGOBJS)
	$(HIPCC) --offload-arch=$(ARCH) -fPIC -shared -o $@ $(DIAGOBJS) -Wl,-rpath,/opt/rocm/lib
diag: $(DIAGLIB)

cli: bin/plaac
bin/plaac: $(CSRC)/plaac_cli.cpp $(LIB)
	mkdir -p bin
	$(HIPCC) -O2 -std=c++17 -Iinclude -o $@ $(CSRC)/plaac_cli.cpp -Lplaac_amd -lplaac_native \
		-Wl,-rpath,'$$ORIGIN/../plaac_amd' -Wl,-rpath,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle

# JNI shim for the reference's Java host (jni/PlaacNative.java + jni/plaac_jni.cpp -> jni/libplaac_jni.so,
# jni/PlaacNative.class). Needs a JDK: built only where $(JAVA_HOME)/include/jni.h exists (not in this image).
JAVA_HOME ?= $(shell dirname $$(dirname $$(readlink -f $$(command -v javac 2>/dev/null) 2>/dev/null) 2>/dev/null) 2>/dev/null)
JNI_H      = $(wildcard $(JAVA_HOME)/include/jni.h)
jni: $(if $(JNI_H),jni/libplaac_jni.so,jni-skipped)
jni/libplaac_jni.so: jni/plaac_jni.cpp jni/PlaacNative.java $(LIB) include/plaac_native.h
	g++ -O2 -std=c++17 -fPIC -shared -Iinclude -I$(JAVA_HOME)/include -I$(JAVA_HOME)/include/linux -o $@ jni/plaac_jni.cpp \
		-Lplaac_amd -lplaac_native -Wl,-rpath,'$$ORIGIN/../plaac_amd' -Wl,-rpath,/opt/rocm/lib
	$(JAVA_HOME)/bin/javac -d jni jni/PlaacNative.java
jni-skipped:
	@echo "jni: no JDK found (JAVA_HOME/include/jni.h missing) - shim not built"

asm: $(CSRC)/plaac_kernels.hip $(wildcard $(CSRC)/*.hip.inc)
	mkdir -p build
	$(HIPCC) $(HIPFLAGS) -Iinclude --cuda-device-only -S -o build/plaac_kernels.s $(CSRC)/plaac_kernels.hip \
		-Rpass-analysis=kernel-resource-usage 2> build/resource_usage.txt || true

# the stand-alone measurement probes of tools/ (issue costs, hardware queues, counter calibration, store shapes): build/<name>
PROBES = $(patsubst tools/%.hip,build/%,$(wildcard tools/*.hip))
probes: $(PROBES)
build/%: tools/%.hip
	mkdir -p build
	$(HIPCC) -O3 --offload-arch=$(ARCH) -o $@ $<

clean:
	rm -f $(LIB) $(DIAGLIB) bin/plaac
	$(MAKE) -C oracle clean

.PHONY: all oracle cli asm clean jni jni-skipped probes diag
