#!/bin/bash
O=gpurun_out/r02i; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
V=tools/bin/variants
REPEAT=3 WL="c3 c2 c5 c5h" timeout 1800 tools/ab.sh $V/cur.so $V/e24.so $V/e24_w3.so > $O/ab.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
for i in 1 2; do timeout 120 python tools/placement_mix.py 2>&1 | grep -v amdgpu.ids; done > $O/placement_mix.txt
tail -3 $O/pytest_gpu.txt; cat $O/ab.txt; cat $O/placement_mix.txt
