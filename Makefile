# Top-level convenience targets.  The product is built by `python -c "import __graft_entry__ as g; g.build()"` (hipcc, no make needed);
# this file holds the maintainer-side PIN KIT (tools/pin_kit.md): compare the CPU oracle with the reference itself on a machine that has
# OpenCV 3.4.8, Eigen, Sophus, g2o and glog.  None of it runs in this repository's build box (no network, libraries absent).
#     make pin REF=/path/to/A-Simple-Stereo-SLAM-System-with-Deep-Loop-Closing
PYTHON ?= python3
CXX ?= g++
REF ?= /root/reference
PIN_DIR = tests/golden/reference
PIN_BIN = tools/build/dump_reference_goldens
OPENCV_PC ?= $(shell pkg-config --exists opencv && echo opencv || echo opencv4)
G2O_LIBS ?= -lg2o_core -lg2o_stuff -lg2o_solver_csparse -lg2o_csparse_extension -lg2o_solver_dense -lg2o_solver_eigen -lg2o_types_slam3d -lcxsparse
PIN_INC = -I$(REF)/include -I/usr/include/eigen3 -I/usr/include/suitesparse -I/usr/local/include $(shell pkg-config --cflags $(OPENCV_PC) 2>/dev/null)
PIN_LIBS = $(shell pkg-config --libs $(OPENCV_PC) 2>/dev/null) -L/usr/local/lib $(G2O_LIBS) -lglog -lgflags -lpthread
# the reference's one translation unit the dump program needs, compiled WHERE IT LIES (never copied); PIN_LIBMYSLAM=1 links the maintainer's libmyslam.so instead
ifeq ($(PIN_LIBMYSLAM),1)
PIN_REFSRC = -L$(REF)/lib -lmyslam -Wl,-rpath,$(REF)/lib
else
PIN_REFSRC = $(REF)/src/ORBextractor.cpp
endif

.PHONY: pin pin-inputs pin-build pin-dump pin-opencv pin-test pin-docker build test

pin: pin-inputs pin-build pin-dump pin-opencv pin-test

pin-inputs:
	$(PYTHON) tools/make_reference_inputs.py

pin-build:
	@test -f $(REF)/src/ORBextractor.cpp || { echo "REF=$(REF) is not a checkout of the reference"; exit 1; }
	mkdir -p tools/build
	$(CXX) -O2 -std=c++14 -w tools/dump_reference_goldens.cpp $(PIN_REFSRC) -o $(PIN_BIN) $(PIN_INC) $(PIN_LIBS)

pin-dump:
	$(PIN_BIN) $(PIN_DIR)

pin-opencv:
	@$(PYTHON) -c "import cv2" 2>/dev/null && $(PYTHON) tools/dump_opencv_goldens.py || echo "pin-opencv: no cv2 for $(PYTHON): tests/test_opencv_pin.py stays XFAIL"

pin-test:
	$(PYTHON) -m pytest tests/test_reference_pin.py tests/test_opencv_pin.py -q -rx

pin-docker:
	docker build -f tools/pin_kit.Dockerfile -t myslam-pin .
	docker run --rm -v $(CURDIR):/repo -v $(abspath $(REF)):/ref:ro -w /repo myslam-pin make pin REF=/ref PYTHON=python3

build:
	$(PYTHON) -c "import __graft_entry__ as g; g.build()"

test:
	$(PYTHON) -m pytest tests -x -q -m "not gpu"
