cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
timeout 300 python -m pytest $R/tests/test_formats_df.py $R/tests/test_martian.py $R/tests/test_graphio.py -q -x 2>&1 | tail -3
timeout 300 python $R/tools/host_seam_probe.py 1e7 pageable 2 2>&1 | grep -E "rep|unitig file"
timeout 300 python $R/tools/host_seam_probe.py 1e7 pinned 2 2>&1 | grep -E "rep|unitig file"
timeout 600 python $R/tools/host_seam_probe.py 1e8 pageable 3 2>&1 | grep -E "rep|unitig file"
timeout 600 python $R/tools/host_seam_probe.py 1e8 pinned 3 2>&1 | grep -E "rep|unitig file"
