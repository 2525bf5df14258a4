#!/bin/bash
O=gpurun_out/r02j; mkdir -p $O; export TMPDIR=/tmp
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so; cp pfac_amd/lib/libpfac.so /tmp/keep_host.so
V=tools/bin/variants
REPEAT=3 WL="c3 c2 c5" timeout 1800 tools/ab.sh $V/e24.so $V/l1.so > $O/ab.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so; cp /tmp/keep_host.so pfac_amd/lib/libpfac.so
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
timeout 300 tools/pmc_mini.sh $V/l1.so > $O/pmc_mini.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
tail -3 $O/pytest_gpu.txt; cat $O/ab.txt; cat $O/pmc_mini.txt
