#!/bin/bash
# round 5: TIGHT count kernel, booking sent before the hash (whole-bucket passes) with the bounded wait in place
for lib in supernova_amd/libsnk.so supernova_amd/variants/libsnk_early.so; do
export SNK_LIB_PATH=$PWD/$lib
timeout 300 python tools/err_probe.py 1e8 e06,e15 2>&1 | grep -v amdgpu | grep "call 3" | sed "s|^|$lib |"
done
