#!/bin/bash
# experiment: run a walker round only when it is nearly full
O=gpurun_out/r02rm; mkdir -p $O; export TMPDIR=/tmp
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
V=tools/bin/variants
REPEAT=4 WL="c3 c5" timeout 1700 tools/ab.sh $V/cur.so $V/rm64.so $V/rm96.so $V/rm112.so $V/rm96e.so $V/rm112e.so $V/rm128e.so > $O/ab.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
cat $O/ab.txt
