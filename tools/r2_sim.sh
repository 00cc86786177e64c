cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
for w in $1; do echo "== W=$w join=${SNK_JOIN:-owner}"; timeout 600 python $R/tools/sim_scale.py $w ${2:-1.25e7} 2 serial 2>&1 | grep "^rep1" | grep -E "rank0 "; done
