#!/bin/bash
# round 5: ungrouped bit filter as committed (auto above 0.3 distinct k-mers per instance, 4000 instances per bucket): suite, forced parity, models
timeout 2400 python -m pytest tests -m gpu -q --timeout 400 > gpurun_out/r5_suite.log 2>&1; grep -E "passed|failed|error" gpurun_out/r5_suite.log | tail -3
SNK_COUNT_SCREEN_NG=2 timeout 1200 python -m pytest tests/test_gpu_parity.py -q --timeout 200 -k "golden or oracle or hot or passes or minbc or independence or k60 or booked or full_size" 2>&1 | grep -E "passed|failed|Error|error" | tail -3
SNK_COUNT_SCREEN_NG=2 timeout 600 python tests/tools/fuzz_parity.py 60 9191 2>&1 | tail -1
timeout 300 python tools/err_probe.py 1e8 e06,e15 2>&1 | grep -v amdgpu | grep "call"
