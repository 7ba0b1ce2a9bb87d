#!/bin/bash
# builds tools/devcount/libmnk_devcount.so (rocprofiler-sdk tool library; no GPU needed to build)
cd "$(dirname "$0")"
g++ -O2 -std=c++17 -fPIC -shared -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ mnk_devcount.cpp -o libmnk_devcount.so -L/opt/rocm/lib -lrocprofiler-sdk -Wl,-rpath,/opt/rocm/lib
