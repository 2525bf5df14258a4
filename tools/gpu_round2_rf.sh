#!/bin/bash
O=gpurun_out/r02rf; mkdir -p $O; export TMPDIR=/tmp
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
V=tools/bin/variants
REPEAT=4 WL="c3 c5" timeout 1800 tools/ab.sh $V/cur.so $V/rf32.so $V/rf40.so $V/rf48.so $V/rf56.so > $O/ab2.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
cat $O/ab2.txt
