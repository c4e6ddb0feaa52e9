#!/bin/bash
# tuning only: build the HIP library with extra -D flags into build_variants/lib_<name>.so (select it with NEB_LIB_PATH)
# usage: tools/build_variant.sh <name> [-DNEB_...=...]...
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_variants
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -fno-gpu-rdc -fno-slp-vectorize -w "$@" \
    nebulae_amd/csrc/api.hip nebulae_amd/csrc/svgf.hip nebulae_amd/csrc/gi.hip nebulae_amd/csrc/gi_build.hip nebulae_amd/csrc/gi_sun_table.hip nebulae_amd/csrc/raysort.hip nebulae_amd/csrc/strips.hip -ldl -o build_variants/lib_$name.so
