#!/bin/bash
O=gpurun_out/r02pf; mkdir -p $O; export TMPDIR=/tmp
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so; cp pfac_amd/lib/libpfac.so /tmp/keep_host.so
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
V=tools/bin/variants
REPEAT=2 WL="c3 c2 c5" timeout 2400 tools/ab.sh $V/prev.so $V/lc.so > $O/ab28.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so; cp /tmp/keep_host.so pfac_amd/lib/libpfac.so
cat $O/ab28.txt
