# Build of the eMI355X backend (no cmake/ninja needed; hipcc + g++ + make).
#
#   make            -> etol_amd/lib/libemi355x.so      (HIP kernels + C ABI, gfx950)
#                      etol_amd/lib/libetol_mi355x.so  (C++ host: TrajectoryOptimizer + eMI355X)
#                      etol_amd/lib/etol_mi355x_example1, tests/harness/libetol_harness.so
#   make oracle     -> oracle/liboracle.so, oracle/libepsopt_style.so  (test infrastructure)
#
# Built artefacts are git-ignored but travel to the GPU box with the snapshot.
ROCM      ?= /opt/rocm
HIPCC     ?= $(ROCM)/bin/hipcc
CXX       ?= g++
CC        ?= gcc
ARCH      ?= gfx950
LIBDIR    := etol_amd/lib
CSRC      := etol_amd/csrc
HOST      := etol_amd/host

HIPFLAGS  := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -I$(CSRC) -Wall -Wno-unused-function -Wno-inline-asm $(EXTRA_HIPFLAGS)
CXXFLAGS  := -O2 -std=c++17 -fPIC -Iinclude -Wall
XML2_INC  := -I/usr/include/libxml2
XML2_LIB  := -lxml2

.PHONY: all lib host oracle clean
ifneq ($(wildcard etol_amd/host/eMI355X.cpp),)
all: lib host oracle
else
all: lib oracle
endif

lib: $(LIBDIR)/libemi355x.so
host: $(LIBDIR)/libetol_mi355x.so $(LIBDIR)/etol_mi355x_example1 $(LIBDIR)/etol_mi355x_montecarlo tests/harness/libetol_harness.so

$(LIBDIR):
	mkdir -p $(LIBDIR)

CSRC_HDR  := $(wildcard $(CSRC)/*.hpp) include/emi355x.h
# headers a model program compiled at run time (hiprtc, emi_rtc.hip) includes: embedded as text
RTC_HDR   := $(CSRC)/emi_models.hpp $(CSRC)/emi_args.hpp $(CSRC)/emi_node_kernels.hpp $(CSRC)/emi_symdefect_kernels.hpp

$(LIBDIR)/emi_kernels.o: $(CSRC)/emi_kernels.hip $(CSRC_HDR) | $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(LIBDIR)/emi_symdefect.o: $(CSRC)/emi_symdefect.hip $(CSRC_HDR) | $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(LIBDIR)/emi_defect_f32.o: $(CSRC)/emi_defect_f32.hip $(CSRC_HDR) | $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(LIBDIR)/emi_api.o: $(CSRC)/emi_api.hip $(CSRC_HDR) | $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(LIBDIR)/emi_kkt.o: $(CSRC)/emi_kkt.hip $(CSRC_HDR) | $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(LIBDIR)/emi_rtc_sources.inc: $(RTC_HDR) tools/embed_src.py | $(LIBDIR)
	python3 tools/embed_src.py $@ $(RTC_HDR)
$(LIBDIR)/emi_rtc.o: $(CSRC)/emi_rtc.hip $(LIBDIR)/emi_rtc_sources.inc $(CSRC_HDR) | $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -I$(LIBDIR) -c $< -o $@
$(LIBDIR)/emi_host.o: $(CSRC)/emi_host.cpp include/emi355x.h | $(LIBDIR)
	$(CXX) $(CXXFLAGS) -c $< -o $@
# RCCL gather (librccl is dlopen'ed at first use, not linked)
$(LIBDIR)/emi_comm.o: $(CSRC)/emi_comm.cpp include/emi355x.h | $(LIBDIR)
	$(CXX) $(CXXFLAGS) -D__HIP_PLATFORM_AMD__ -I$(ROCM)/include -c $< -o $@

$(LIBDIR)/libemi355x.so: $(LIBDIR)/emi_kernels.o $(LIBDIR)/emi_symdefect.o $(LIBDIR)/emi_defect_f32.o $(LIBDIR)/emi_api.o $(LIBDIR)/emi_rtc.o $(LIBDIR)/emi_kkt.o $(LIBDIR)/emi_host.o $(LIBDIR)/emi_comm.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -L$(ROCM)/lib -lhiprtc -lrocsolver -lrocblas -ldl

HOST_SRC := $(HOST)/TrajectoryOptimizer.cpp $(HOST)/eMI355X.cpp $(HOST)/emi_nlp.cpp $(HOST)/emi_trace.cpp
HOST_HDR := $(wildcard include/ETOL/*.hpp) $(wildcard $(HOST)/*.hpp) include/emi355x.h

# the host side of the Newton step (block eigen-decompositions, r x r Cholesky of the low-rank correction) wants AVX2
$(LIBDIR)/libetol_mi355x.so: CXXFLAGS := -O3 -march=x86-64-v3 -fopenmp-simd -std=c++17 -fPIC -Iinclude -Wall
$(LIBDIR)/libetol_mi355x.so: $(HOST_SRC) $(HOST_HDR) $(LIBDIR)/libemi355x.so
	$(CXX) $(CXXFLAGS) $(XML2_INC) -I$(HOST) -shared -o $@ $(HOST_SRC) -L$(LIBDIR) -lemi355x $(XML2_LIB) \
		-Wl,-rpath,'$$ORIGIN'

$(LIBDIR)/etol_mi355x_example1: etol_amd/examples/etol_mi355x_example1.cpp $(LIBDIR)/libetol_mi355x.so
	$(CXX) $(CXXFLAGS) -I$(HOST) -o $@ $< -L$(LIBDIR) -letol_mi355x -lemi355x -Wl,-rpath,'$$ORIGIN'

$(LIBDIR)/etol_mi355x_montecarlo: etol_amd/examples/etol_mi355x_montecarlo.cpp $(LIBDIR)/libetol_mi355x.so
	$(CXX) $(CXXFLAGS) -I$(HOST) -pthread -o $@ $< -L$(LIBDIR) -letol_mi355x -lemi355x -Wl,-rpath,'$$ORIGIN'

tests/harness/libetol_harness.so: tests/harness/etol_harness.cpp $(LIBDIR)/libetol_mi355x.so $(HOST_HDR)
	$(CXX) $(CXXFLAGS) -I$(HOST) -shared -o $@ $< -L$(LIBDIR) -letol_mi355x -lemi355x -ldl \
		-Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)'

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(LIBDIR) tests/harness/*.so
	$(MAKE) -C oracle clean
