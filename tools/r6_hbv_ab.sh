for sq in 1 0 1 0; do echo "hbv_short_queue=$sq"; SNK_TUNING="hbv_short_queue=$sq" timeout 600 python tools/hbv_scale_probe.py 1e7 2>&1 | grep -v amdgpu | grep -E "flood rep|equal" ; done
