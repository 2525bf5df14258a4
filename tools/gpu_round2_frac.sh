#!/bin/bash
# timing experiment: walk only 0 / 25 / 50 / 75 / 100 % of the candidates (results wrong below 100 %)
O=gpurun_out/r02frac; mkdir -p $O; export TMPDIR=/tmp
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
V=tools/bin/variants
REPEAT=3 WL="c3" timeout 1500 tools/ab.sh $V/abl2.so $V/abl3.so $V/abl4.so $V/abl5.so $V/cur.so > $O/ab.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
cat $O/ab.txt
