#!/bin/bash
O=gpurun_out/r02g; mkdir -p $O; export TMPDIR=/tmp
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
V=tools/bin/variants
REPEAT=3 WL="c2 c3" timeout 2000 tools/ab.sh $V/s2g0.so $V/s2g1.so $V/s2g2.so $V/s3g0.so $V/s3g1.so $V/s2g1p8.so $V/s2g1p16.so $V/s2g1w3.so $V/s2g1w1.so > $O/ab.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
cat $O/ab.txt
