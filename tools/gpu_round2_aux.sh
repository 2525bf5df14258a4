#!/bin/bash
# timing experiment: cache-policy bits on the input stream (does it keep the table's hot lines in the L1?)
O=gpurun_out/r02aux; mkdir -p $O; export TMPDIR=/tmp
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
V=tools/bin/variants
REPEAT=3 WL="c3" timeout 1500 tools/ab.sh $V/cur.so $V/aux0.so $V/aux1.so $V/aux16.so $V/aux17.so $V/aux2.so $V/aux18.so $V/aux19.so > $O/ab.txt 2>&1
REPEAT=2 WL="c2 c5" timeout 1500 tools/ab.sh $V/cur.so $V/aux17.so $V/aux2.so >> $O/ab.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
cat $O/ab.txt
