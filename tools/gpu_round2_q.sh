#!/bin/bash
O=gpurun_out/r02q; mkdir -p $O; export TMPDIR=/tmp
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
V=tools/bin/variants
REPEAT=5 WL="c3 c2" timeout 1800 tools/ab.sh $V/cur.so $V/b16.so $V/b4.so > $O/ab.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
cat $O/ab.txt
