#!/usr/bin/env bash
# tools/build_variant.sh <name> <extra hipcc -D flags...> : a tuning build of libsnk with different compile-time
# parameters of the count kernel -> supernova_amd/variants/libsnk_<name>.so (select with SNK_LIB_PATH=...)
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p supernova_amd/variants supernova_amd/csrc/_obj
o=supernova_amd/csrc/_obj/snk_count_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off "$@" -c supernova_amd/csrc/snk_count.hip -o $o
g++ -shared -fPIC -o supernova_amd/variants/libsnk_$name.so $o $(ls supernova_amd/csrc/_obj/*.o | grep -v snk_count)
echo supernova_amd/variants/libsnk_$name.so
