#!/bin/bash
O=gpurun_out/r02pf; mkdir -p $O; export TMPDIR=/tmp
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
V=tools/bin/variants
REPEAT=2 WL="c3 c2 c5" timeout 2400 tools/ab.sh $V/base.so $V/rb.so > $O/ab23.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
cat $O/ab23.txt
