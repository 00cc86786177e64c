#!/usr/bin/env bash
# tools/build_variant.sh <name> <extra hipcc -D flags...> : a tuning build of libsnk with different compile-time
# parameters of the count kernel -> supernova_amd/variants/libsnk_<name>.so (select with SNK_LIB_PATH=...).
# The files named in $FILES (default: snk_count snk_stages; -DSNK_COUNT_PROF needs both) are recompiled with the flags, the rest is reused.
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p supernova_amd/variants supernova_amd/csrc/_obj/var
objs=()
for f in ${FILES:-snk_count snk_stages}; do
  o=supernova_amd/csrc/_obj/var/${f}_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off "$@" -c supernova_amd/csrc/$f.hip -o $o
  objs+=($o)
done
g++ -shared -fPIC -o supernova_amd/variants/libsnk_$name.so "${objs[@]}" $(ls supernova_amd/csrc/_obj/*.o | grep -v $(for f in ${FILES:-snk_count snk_stages}; do printf -- "-e /%s.o " $f; done)) -lz -ldl -lpthread
echo supernova_amd/variants/libsnk_$name.so
