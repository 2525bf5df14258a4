#!/bin/bash
# timing experiment: a second (dummy) gathered load per table step -- is the walk phase bound by gathered loads?
O=gpurun_out/r02dg; mkdir -p $O; export TMPDIR=/tmp
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
V=tools/bin/variants
REPEAT=3 WL="c3 c5" timeout 1500 tools/ab.sh $V/cur.so $V/dg1.so $V/dg8.so $V/dg4k.so > $O/ab.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
cat $O/ab.txt
