# Builds the gfx950 solver library (C ABI), the C++ `Solver` shim, the headless host
# programs and the CPU oracle.  hipcc cross-compiles without a GPU.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
# -ffp-contract=off: the kernels must round exactly like an uncontracted host build.
HIPFLAGS ?= -O3 --offload-arch=$(ARCH) -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 -Wall -Wno-unused-function
CXX      ?= g++
CXXFLAGS ?= -O2 -std=c++17 -fPIC -Wall

CSRC   := fluidsolver_amd/csrc
LIBDIR := fluidsolver_amd/lib
BINDIR := fluidsolver_amd/bin
LIB    := $(LIBDIR)/libfluidsolver.so
SHIM   := $(LIBDIR)/libCuSolver.so

all: $(LIB) $(SHIM) host oracle

$(LIBDIR) $(BINDIR):
	mkdir -p $@

KOBJS := fs_kernels fs_jacobi fs_advect fs_vort fs_mgpcg
SOBJS := fs_solver fs_exchange fs_project_jacobi fs_project_mgpcg fs_stages fs_step
OBJS  := $(addprefix $(LIBDIR)/,$(addsuffix .o,$(KOBJS) $(SOBJS)))

# kernel translation units
$(addprefix $(LIBDIR)/,$(addsuffix .o,$(KOBJS))): $(LIBDIR)/%.o: $(CSRC)/%.hip $(CSRC)/fs_kernels.h | $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# host-side translation units behind the C ABI
$(addprefix $(LIBDIR)/,$(addsuffix .o,$(SOBJS))): $(LIBDIR)/%.o: $(CSRC)/%.hip $(CSRC)/fs_solver_impl.h $(CSRC)/fs_kernels.h include/fluidsolver.h | $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -Wl,-rpath,/opt/rocm/lib -Wl,-soname,libfluidsolver.so

$(SHIM): $(CSRC)/Solver.cpp include/Solver.h include/fluidsolver.h $(LIB)
	$(CXX) $(CXXFLAGS) -shared -Iinclude -o $@ $(CSRC)/Solver.cpp -L$(LIBDIR) -lfluidsolver -Wl,-rpath,'$$ORIGIN' -Wl,-soname,libCuSolver.so

host: $(BINDIR)/fs_bench $(BINDIR)/fs_headless

$(BINDIR)/fs_bench: fluidsolver_amd/host/fs_bench.cpp include/fluidsolver.h $(LIB) | $(BINDIR)
	$(CXX) $(CXXFLAGS) -Iinclude -o $@ $< -L$(LIBDIR) -lfluidsolver -Wl,-rpath,'$$ORIGIN/../lib'

$(BINDIR)/fs_headless: fluidsolver_amd/host/fs_headless.cpp include/Solver.h $(SHIM) | $(BINDIR)
	$(CXX) $(CXXFLAGS) -Iinclude -o $@ $< -L$(LIBDIR) -lCuSolver -lfluidsolver -Wl,-rpath,'$$ORIGIN/../lib'

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(LIBDIR) $(BINDIR)
	$(MAKE) -C oracle clean

.PHONY: all host oracle clean
