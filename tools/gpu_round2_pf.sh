#!/bin/bash
O=gpurun_out/r02pf; mkdir -p $O; export TMPDIR=/tmp
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
V=tools/bin/variants
REPEAT=2 WL="c3 c2" timeout 2400 tools/ab.sh $V/ip0.so $V/ip1.so $V/ip2.so $V/ip3.so $V/ip7.so > $O/ab22.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
cat $O/ab22.txt
