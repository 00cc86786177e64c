export TMPDIR=/tmp; cd /root/repo
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null; taskset -p $$; lscpu | grep -E "Model name|Thread|Core|Socket|MHz|NUMA" | head -12; free -g | head -2
for t in 8 16 32 64 128; do python tools/fasth_decode_probe.py 128 $t 2>&1 | tail -1; done
SNK_FASTH_LIBDEFLATE=0 python tools/fasth_decode_probe.py 128 64 2>&1 | tail -1
