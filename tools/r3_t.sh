cd /root/repo
timeout 1800 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_asm_sn.py tests/test_gpu_bigparity.py -x -q > gpurun_out/r3_t_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r3_t_tests.log | tail -2
timeout 900 python tools/sim_scale.py 8 1.25e7 > gpurun_out/r03_sim8.log 2>&1; tail -12 gpurun_out/r03_sim8.log | cut -c1-400
timeout 900 python tools/sim_scale.py 2 5e7 > gpurun_out/r03_sim2.log 2>&1; tail -6 gpurun_out/r03_sim2.log | cut -c1-400
