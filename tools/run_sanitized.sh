#!/bin/bash
# The CPU side under AddressSanitizer + UndefinedBehaviorSanitizer (the reference's analogue is the D3D12 debug layer,
# src/nri/Device.cpp:43-50).  BUILD CONTAINER ONLY -- never on the GPU box (GPU sanitizers are refused there, and nothing here needs a GPU):
#   1. oracle/libneb_oracle_san.so (make -C oracle SAN=1) and tools/lit_proto.cpp -- which compiles the very header the device compiles,
#      nebulae_amd/csrc/lit_predicate.h -- built with -fsanitize=address,undefined; the whole CPU suite (pytest -m "not gpu") runs on them,
#      libasan preloaded into the interpreter for the ctypes loads;
#   2. the HOST half of libnebulae_hip.so (hipcc -fsanitize=address,undefined -fno-gpu-sanitize: device code as shipped) under tests/test_abi.py, whose calls never
#      reach a device: argument validation, option parsing, the error paths of neb_create without a GPU.
# usage: bash tools/run_sanitized.sh [extra pytest arguments]      output: profiles/r05_sanitizers.txt is a copy of a run's tail
set -o pipefail
cd "$(dirname "$0")/.."
asan=$(gcc -print-file-name=libasan.so)
ubsan=$(gcc -print-file-name=libubsan.so)
[ -f "$asan" ] || { echo "libasan.so not found (gcc -print-file-name)"; exit 2; }
make -C oracle -s SAN=1 || exit 1
export NEB_ORACLE_SAN=1
# (leak detection off: the interpreter itself never frees most of what it allocates; halt on the first real finding)
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
echo "== 1. CPU suite on the sanitized oracle / certificate =="
LD_PRELOAD="$asan:$ubsan" python -m pytest tests -q -m "not gpu" -x -p no:cacheprovider "$@" || exit 1
echo "== 2. host half of the HIP library under ASan + UBSan (tests/test_abi.py) =="
mkdir -p build_variants
src="nebulae_amd/csrc/api.hip nebulae_amd/csrc/svgf.hip nebulae_amd/csrc/gi.hip nebulae_amd/csrc/gi_build.hip nebulae_amd/csrc/gi_sun_table.hip nebulae_amd/csrc/raysort.hip nebulae_amd/csrc/strips.hip"
if /opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -fPIC -shared -fno-gpu-rdc -fno-slp-vectorize -w -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer \
     -shared-libsan $src -ldl -o build_variants/lib_host_san.so 2> build_variants/host_san_build.log; then
  clang_rt=$(dirname "$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)")
  NEB_LIB_PATH=$PWD/build_variants/lib_host_san.so LD_LIBRARY_PATH="$clang_rt:$LD_LIBRARY_PATH" \
    LD_PRELOAD="$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)" NEB_ORACLE_SAN= \
    python -m pytest tests/test_abi.py -q -x -p no:cacheprovider || exit 1
else
  echo "host-side sanitizer build of the HIP library failed (see build_variants/host_san_build.log): step 2 skipped"; tail -5 build_variants/host_san_build.log
fi
echo "== sanitizers: done =="
