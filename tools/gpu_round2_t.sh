#!/bin/bash
O=gpurun_out/r02t; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "reduce or Reduce or fuzz or thread or every_position" > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
V=tools/bin/variants
REPEAT=3 WL="c3 c2 c5" timeout 900 tools/ab.sh $V/cur.so $V/red16.so > $O/ab.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
tail -3 $O/pytest_gpu.txt; cat $O/ab.txt
