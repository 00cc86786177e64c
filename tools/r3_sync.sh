cd /root/repo
SNK_SYNC_TRACE=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-next-rows --no-ingest --sharded --no-verify > gpurun_out/r3_sync_trace_sharded.log 2>&1
SNK_SYNC_TRACE=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-next-rows --no-ingest > gpurun_out/r3_sync_trace_single.log 2>&1
grep -c "snk sync" gpurun_out/r3_sync_trace_sharded.log gpurun_out/r3_sync_trace_single.log
