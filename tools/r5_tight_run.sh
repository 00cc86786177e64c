#!/bin/bash
# round 5: thread-0 phase profile of the count kernel (SNK_COUNT_PROF build) on clean / 0.6 % / 1.5 % reads
export SNK_LIB_PATH=$PWD/supernova_amd/variants/libsnk_prof.so
timeout 300 python tools/err_probe.py 1e8 headline,e06,e15 2>&1 | grep -v amdgpu | grep "call 3\|snk prof" | tail -40
