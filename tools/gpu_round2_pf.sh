#!/bin/bash
O=gpurun_out/r02pf; mkdir -p $O; export TMPDIR=/tmp
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
V=tools/bin/variants
REPEAT=2 WL="c3 c2" timeout 2400 tools/ab.sh $V/zb0.so $V/zb1.so $V/zb2.so $V/zb3.so $V/zb4.so > $O/ab25.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
cat $O/ab25.txt
