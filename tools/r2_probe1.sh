cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
$R/tools/probe/_bin/valu_rate > $O/valu_rate.txt 2>&1
python -m pytest $R/tests -m gpu -x -q > $O/t.log 2>&1; grep -E "passed|failed|rror" $O/t.log | tail -3
python $R/tools/count_probe.py 1e8 0,1,2,4 > $O/cp_main.log 2>&1; cat $O/cp_main.log
for v in run2 run3; do SNK_LIB_PATH=$R/supernova_amd/variants/libsnk_$v.so python $R/tools/count_probe.py 1e8 0 > $O/cp_$v.log 2>&1; echo $v; cat $O/cp_$v.log; done
SNK_LIB_PATH=$R/supernova_amd/variants/libsnk_prof.so python $R/tools/count_probe.py 1e8 0 > $O/cp_prof.log 2>&1; tail -3 $O/cp_prof.log
python $R/bench.py --no-cpu-baseline > $O/bench1.log 2>&1; tail -1 $O/bench1.log | cut -c1-600
