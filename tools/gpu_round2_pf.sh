#!/bin/bash
O=gpurun_out/r02pf; mkdir -p $O; export TMPDIR=/tmp
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
V=tools/bin/variants
REPEAT=2 WL="c3 c2" timeout 2400 tools/ab.sh $V/d16.so $V/d16ra8.so $V/d16ra32.so $V/d24.so $V/d12.so > $O/ab12.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
cat $O/ab12.txt
