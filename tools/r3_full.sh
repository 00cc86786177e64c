set -x
cd /root/repo
export PYTHONUNBUFFERED=1
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r3_full_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_full_tests.log
tail -8 gpurun_out/r3_full_tests.log
