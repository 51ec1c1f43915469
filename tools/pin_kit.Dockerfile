# Environment of the pin kit (tools/pin_kit.md): the libraries the reference links, at the versions its README names, on its platform.
#   docker build -f tools/pin_kit.Dockerfile -t myslam-pin .        (needs a network connection; nothing here runs in this repository's build box)
#   docker run --rm -v $PWD:/repo -v $REF:/ref:ro -w /repo myslam-pin make pin REF=/ref
FROM ubuntu:18.04
ENV DEBIAN_FRONTEND=noninteractive
RUN apt-get update && apt-get install -y --no-install-recommends build-essential cmake git pkg-config ca-certificates wget \
        python3 python3-dev python3-pip python3-setuptools python3-numpy \
        libeigen3-dev libgoogle-glog-dev libgflags-dev libsuitesparse-dev libboost-filesystem-dev && rm -rf /var/lib/apt/lists/*
RUN pip3 install --no-cache-dir pytest scipy
WORKDIR /deps
# OpenCV 3.4.8 (reference README.md:28).  IPP off: the oracle restates the portable fixed-point paths of cv::resize / cv::GaussianBlur.
RUN git clone --depth 1 -b 3.4.8 https://github.com/opencv/opencv && \
    cmake -S opencv -B opencv/build -DCMAKE_BUILD_TYPE=Release -DBUILD_LIST=core,imgproc,imgcodecs,highgui,features2d,calib3d,video,flann,python3 \
          -DWITH_IPP=OFF -DBUILD_TESTS=OFF -DBUILD_PERF_TESTS=OFF -DBUILD_EXAMPLES=OFF -DOPENCV_GENERATE_PKGCONFIG=ON && \
    make -C opencv/build -j"$(nproc)" && make -C opencv/build install && rm -rf opencv
# Sophus (README.md:36-37): the templated se3.hpp the reference includes (common_include.h:104-105)
RUN git clone https://github.com/strasdat/Sophus && git -C Sophus checkout 13fb3288 && \
    cmake -S Sophus -B Sophus/build -DBUILD_TESTS=OFF -DBUILD_EXAMPLES=OFF && make -C Sophus/build install && rm -rf Sophus
# g2o (README.md:39-40): a snapshot with g2o::make_unique (src/backend.cpp:128-133) and the CSparse solver (CMakeLists.txt:86)
RUN git clone https://github.com/RainerKuemmerle/g2o && git -C g2o checkout 20200410_git && \
    cmake -S g2o -B g2o/build -DCMAKE_BUILD_TYPE=Release -DG2O_BUILD_EXAMPLES=OFF -DG2O_BUILD_APPS=OFF && \
    make -C g2o/build -j"$(nproc)" && make -C g2o/build install && ldconfig && rm -rf g2o
