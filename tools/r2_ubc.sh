cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
cd $R && timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "unitig_barcode or read_paths or mark_dups" 2>&1 | tail -6
