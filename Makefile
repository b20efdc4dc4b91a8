# Build without Python: the same commands `python -c "import __graft_entry__ as g; g.build()"` runs.
#   make            libmi_blur.so + both hosts
#   make CIMG=/path/to/dir/holding/CImg.h [JPEG=1]   hosts with the reference's CImg I/O (CImg is NOT vendored here)
#   make oracle     the CPU checker (test infrastructure only; never linked into the product)
#   make test       CPU suite;   make gputest   the -m gpu suite (needs an MI355X)
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
PKG     := heterogeneous-opencl-image-processing-engine_amd
CSRC    := $(PKG)/csrc
APPS    := $(PKG)/apps
LIB     := $(PKG)/libmi_blur.so
LIBSRC  := $(CSRC)/blur_kernels.hip $(CSRC)/layout_kernels.hip $(CSRC)/mi_blur_api.cpp $(CSRC)/cpu_device.cpp
LIBDEPS := $(LIBSRC) $(CSRC)/blur_launch.h $(CSRC)/cpu_device.h include/mi_blur.h
APPFLAGS := -O2 -std=c++17 -Wall -Wextra -I include
ifdef CIMG
APPFLAGS += -DMI_BLUR_WITH_CIMG -I $(CIMG)
ifdef JPEG
APPFLAGS += -Dcimg_use_jpeg
APPLIBS  += -ljpeg
endif
endif

all: $(LIB) $(APPS)/heterogeneous_blur $(APPS)/split_image_blur

$(LIB): $(LIBDEPS)
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -Wall -Wextra -o $@ $(LIBSRC) -ldl -lpthread

$(APPS)/%: $(APPS)/%.cpp $(APPS)/host_common.h include/mi_blur.h $(LIB)
	$(HIPCC) $(APPFLAGS) -o $@ $< -L $(PKG) -lmi_blur '-Wl,-rpath,$$ORIGIN/..' -lpthread $(APPLIBS)

oracle:
	$(MAKE) -C oracle

test: all
	python -m pytest tests -q -m "not gpu"

gputest: all
	python -m pytest tests -q -m gpu

clean:
	rm -f $(LIB) $(APPS)/heterogeneous_blur $(APPS)/split_image_blur

.PHONY: all oracle test gputest clean
