#!/bin/bash
O=gpurun_out/r02pf; mkdir -p $O; export TMPDIR=/tmp
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
V=tools/bin/variants
REPEAT=2 WL="c2 c3" timeout 2400 tools/ab.sh $V/fs0.so $V/fs3.so $V/fs5.so $V/fs6.so $V/fs7.so > $O/ab30.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
cat $O/ab30.txt
