cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bigparity.py -x -q -k "path or dup or barcode or unitig_bc or finger or 200k" > gpurun_out/r3_path_tests.log 2>&1; tail -3 gpurun_out/r3_path_tests.log
timeout 600 python tools/path_probe.py 1e8 2 2>&1 | tail -1 | cut -c1-250
