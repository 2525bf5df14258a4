#!/bin/bash
O=gpurun_out/r02b; mkdir -p $O; export TMPDIR=/tmp
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
REPEAT=4 WL="c3" timeout 1200 tools/ab.sh tools/bin/variants/base.so tools/bin/variants/new.so tools/bin/variants/front4.so tools/bin/variants/front6.so > $O/ab.txt 2>&1
timeout 600 tools/pmc_mini.sh tools/bin/variants/base.so tools/bin/variants/new.so > $O/pmc_mini.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc $?" >> $O/bench_default.err
cat $O/ab.txt; cat $O/pmc_mini.txt; tail -5 $O/bench_default.err
