#!/bin/bash
O=gpurun_out/r02m; mkdir -p $O; export TMPDIR=/tmp
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
V=tools/bin/variants
REPEAT=3 WL="c3 c2" timeout 1800 tools/ab.sh $V/cur.so $V/nt.so > $O/ab.txt 2>&1
timeout 900 tools/pmc_ab.sh c3 $V/cur.so $V/cur_abl2.so > $O/pmc_ab.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
cat $O/ab.txt; cat $O/pmc_ab.txt
