# Host-only sanitizer builds of the CLI twin (no GPU needed; GPU AddressSanitizer / XNACK are not available on this
# pool). make -C rocoder_amd/csrc/host -f sanitize.mk   (tools/run_sanitizers.sh builds, runs and logs them)
#   ../../bin/rocoder_asan : host/rocoder_cli.cpp with -fsanitize=address,undefined over the real engine library - the
#                            WAV parser, duration grammar, flag parsing and (on a GPU box) the whole CLI
#   ../../bin/rocoder_tsan : the same source with -fsanitize=thread over tests/c/stub_engine.c (a stand-in that
#                            computes nothing): StretcherProcessor thread, WindowQueue, AudioBus drain, hot-swap watcher
ROOT := ../../..
CXX  ?= g++
COMMON := -g -O1 -std=c++17 -Wall -fno-omit-frame-pointer rocoder_cli.cpp -ldl -lpthread
all: ../../bin/rocoder_asan ../../bin/rocoder_tsan ../../bin/engine_asan ../../bin/engine_tsan
../../bin/rocoder_asan: rocoder_cli.cpp $(ROOT)/include/rocoder_hip.h ../../librocoder_hip.so
	mkdir -p ../../bin
	$(CXX) -fsanitize=address,undefined -fno-sanitize-recover=undefined $(COMMON) -o $@ -L../.. -lrocoder_hip \
	    -Wl,-rpath,'$$ORIGIN/..' -Wl,-rpath-link,/opt/rocm/lib
../../bin/libstub_engine.so: $(ROOT)/tests/c/stub_engine.c $(ROOT)/include/rocoder_hip.h
	mkdir -p ../../bin
	gcc -g -O1 -fPIC -shared -fsanitize=thread -I$(ROOT)/include -o $@ $(ROOT)/tests/c/stub_engine.c
../../bin/rocoder_tsan: rocoder_cli.cpp $(ROOT)/include/rocoder_hip.h ../../bin/libstub_engine.so
	$(CXX) -fsanitize=thread $(COMMON) -o $@ -L../../bin -lstub_engine -Wl,-rpath,'$$ORIGIN'
# The ENGINE's host code (rc_engine.cpp: worker pools, the pinned three-set pipeline, rc_multi's persistent workers, the
# streaming seam) host-only over tests/c/hip_stub.cpp - a HIP runtime whose device memory is host memory and whose
# kernels compute nothing - driven by tests/c/engine_host_driver.cpp:
#   ../../bin/engine_asan : -fsanitize=address,undefined      ../../bin/engine_tsan : -fsanitize=thread
ENGINE_SRC := ../rc_engine.cpp $(ROOT)/tests/c/hip_stub.cpp $(ROOT)/tests/c/engine_host_driver.cpp
ENGINE_FLAGS := -g -O1 -std=c++17 -Wall -Wno-unused-function -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -DRC_PMAX=32 \
    -I/opt/rocm/include -x c++
../../bin/engine_asan: $(ENGINE_SRC) ../rc_kernels.h $(ROOT)/include/rocoder_hip.h
	mkdir -p ../../bin
	$(CXX) -fsanitize=address,undefined -fno-sanitize-recover=undefined $(ENGINE_FLAGS) $(ENGINE_SRC) -o $@ -lpthread
../../bin/engine_tsan: $(ENGINE_SRC) ../rc_kernels.h $(ROOT)/include/rocoder_hip.h
	mkdir -p ../../bin
	$(CXX) -fsanitize=thread $(ENGINE_FLAGS) $(ENGINE_SRC) -o $@ -lpthread
.PHONY: all
