#!/bin/bash
O=gpurun_out/r02e; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
REPEAT=3 WL="c3 c2" timeout 1500 tools/ab.sh tools/bin/variants/base.so tools/bin/variants/w0.so tools/bin/variants/w1.so tools/bin/variants/w2.so tools/bin/variants/w3.so tools/bin/variants/w4.so > $O/ab.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
tail -3 $O/pytest_gpu.txt; cat $O/ab.txt
