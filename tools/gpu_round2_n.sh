#!/bin/bash
O=gpurun_out/r02n; mkdir -p $O; export TMPDIR=/tmp
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
V=tools/bin/variants
REPEAT=6 WL="c3" timeout 1800 tools/ab.sh $V/cur.so $V/touch.so > $O/ab.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
cat $O/ab.txt
for i in 1 2; do timeout 120 python tools/placement_mix.py 2>&1 | grep -v amdgpu.ids; done > $O/placement_mix.txt; cat $O/placement_mix.txt
