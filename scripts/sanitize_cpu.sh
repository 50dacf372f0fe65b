#!/bin/bash
# ASan + UBSan over the CPU-side code (oracle C, host loader / ABI C++), run with the
# CPU test suite.  GPU sanitizers are not available on the pool; device code is
# covered by the parity + fuzz tests instead.
set -e
cd "$(dirname "$0")/.."
make -j8 all >/dev/null
T=$(mktemp -d)
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -O1 -g -ffp-contract=off"
gcc $SAN -std=gnu11 -fPIC -shared -o $T/oracle.so oracle/c2rt_oracle.c -lm -lpthread
for f in chess2rt_amd/csrc/c2rt_api.cpp chess2rt_amd/csrc/host/dsc.cpp chess2rt_amd/csrc/host/scene.cpp chess2rt_amd/csrc/host/host_api.cpp; do
  g++ $SAN -std=c++17 -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c $f -o $T/$(basename $f .cpp).o
done
hipcc --offload-arch=gfx950 -shared -fPIC -o chess2rt_amd/libc2rt_san.so build/c2rt_kernels_u*.o $T/*.o -lpthread
cp oracle/libc2rt_oracle.so $T/orig.so
trap 'cp $T/orig.so oracle/libc2rt_oracle.so; rm -f chess2rt_amd/libc2rt_san.so' EXIT
cp $T/oracle.so oracle/libc2rt_oracle.so
export ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
C2RT_LIB_VARIANT=san python -m pytest tests/test_oracle_golden.py tests/test_host_loader.py tests/test_abi_exports.py -x -q -p no:cacheprovider
