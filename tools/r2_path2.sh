cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
cd $R && timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "unitig_barcode or read_paths or mark_dups or reference_binary" 2>&1 | tail -4
cd /tmp; timeout 300 python $R/tools/path_probe.py 1e8 2 2>&1 | grep "^rep" | cut -c1-330
