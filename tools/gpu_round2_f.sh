#!/bin/bash
O=gpurun_out/r02f; mkdir -p $O; export TMPDIR=/tmp
timeout 100 tools/bin/stream_probe > $O/stream_probe.txt 2>&1
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
V=tools/bin/variants
REPEAT=3 WL="c2 c3" timeout 1800 tools/ab.sh $V/base.so $V/s4.so $V/s2.so $V/s1.so $V/s0.so $V/s2f.so $V/s4f.so $V/s1w3.so > $O/ab.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
cat $O/stream_probe.txt; cat $O/ab.txt
