#!/bin/bash
# The CPU test suite against sanitizer builds of the host code: the oracle (oracle/crt_ref.c) and the host stepper
# (tests/hostsim: crt_api.cpp's host paths, crt_bvh.cpp, the traversal headers).  Host code only: GPU sanitizers are not
# available on this pool.  Usage (repo root): bash tools/sanitize_cpu_suite.sh        -> rebuilds the normal libraries at the end
set -e
HS="-O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mfma -I/opt/rocm/include -D__HIP_PLATFORM_AMD__"
OR="-O1 -g -std=gnu11 -ffp-contract=off -fno-fast-math -mfma -fPIC -pthread -shared"
echo "== UndefinedBehaviorSanitizer"
make -s -C tests/hostsim -B CXXFLAGS="$HS -fsanitize=undefined -fno-sanitize-recover=undefined"
(cd oracle && cc $OR -fsanitize=undefined -fno-sanitize-recover=undefined -o liboracle_crt.so crt_ref.c -lm -lpthread)
python -m pytest tests -x -q -m "not gpu" | tail -2
echo "== AddressSanitizer"
make -s -C tests/hostsim -B CXXFLAGS="$HS -fsanitize=address"
(cd oracle && cc $OR -fsanitize=address -o liboracle_crt.so crt_ref.c -lm -lpthread)
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so) python -m pytest tests -x -q -m "not gpu" | tail -2
echo "== back to the normal builds"
make -s -C tests/hostsim -B
make -s -C oracle -B
# The library's own host code (crt_api.cpp, crt_bvh.cpp, the host halves of the .hip files) under UBSan on the GPU box -- run once in
# round 3, all 190 GPU tests passed:
#   g++ ... -fsanitize=undefined -fno-sanitize=vptr -fno-sanitize-recover=undefined -c crt_api.cpp crt_bvh.cpp
#   hipcc ... -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize=function,vptr -Xarch_host -fno-sanitize-recover=undefined -c *.hip
#   hipcc -shared ... *.o $(gcc -print-file-name=libubsan.so) -o ablib/lib_ubsan.so
#   CRT_LIB=$PWD/ablib/lib_ubsan.so python -m pytest tests -m gpu     (via gpurun)
