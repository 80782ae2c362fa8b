# Top-level build: the product library (HIP, gfx950 only) and the CPU checker.
#   make            -> plaac_amd/libplaac_native.so + oracle/libplaac_oracle.so (+ bin/plaac when its source exists)
# -ffp-contract=off is part of the numerical contract: never remove it.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result
CSRC      = plaac_amd/csrc
LIB       = plaac_amd/libplaac_native.so

# make DIAG=1: the same library with the result-breaking ablation switches compiled in (PLAAC_DEBUG_SKIP / _SKIP_FROM: named
# kernels are not launched, rows stale; PLAAC_VIT_STOP; PLAAC_DEBUG_COUNTER) as plaac_amd/libplaac_native_diag.so, which
# tools/r04_ablate*.sh load through PLAAC_NATIVE_LIB. The release library does not read them.
DIAGLIB   = plaac_amd/libplaac_native_diag.so
ifeq ($(DIAG),1)
all: $(DIAGLIB)
endif

all: $(LIB) oracle $(if $(wildcard $(CSRC)/plaac_cli.cpp),cli) $(if $(JNI_H),jni)

LIBSRC    = $(CSRC)/plaac_kernels.hip $(CSRC)/plaac_host.cpp $(CSRC)/plaac_node.cpp $(wildcard $(CSRC)/plaac_io.cpp)
# The latency-form chain kernels a second time, as a unit of their own under the compiler's max-ilp instruction scheduling
# (plaac_kernels_lat.hip: chain-bound calls 6 % faster with it, the throughput-bound headline 3.7 % slower - so only there)
LATOBJ    = build/plaac_kernels_lat.o
$(LATOBJ): $(CSRC)/plaac_kernels_lat.hip $(wildcard $(CSRC)/*.hip.inc) $(wildcard include/*.h)
	mkdir -p build
	$(HIPCC) $(HIPFLAGS) -Wno-unused-function -mllvm -amdgpu-sched-strategy=max-ilp -Iinclude -c -o $@ $(CSRC)/plaac_kernels_lat.hip

$(LIB): $(LIBSRC) $(LATOBJ) $(wildcard $(CSRC)/*.hip.inc) $(wildcard include/*.h)
	$(HIPCC) $(HIPFLAGS) -Iinclude -shared -o $@ $(LIBSRC) -Wl,$(LATOBJ) -Wl,-rpath,/opt/rocm/lib

$(DIAGLIB): $(LIBSRC) $(LATOBJ) $(wildcard $(CSRC)/*.hip.inc) $(wildcard include/*.h)
	$(HIPCC) $(HIPFLAGS) -DPLAAC_DIAG=1 -Iinclude -shared -o $@ $(LIBSRC) -Wl,$(LATOBJ) -Wl,-rpath,/opt/rocm/lib
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
