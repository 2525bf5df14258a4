#!/bin/bash
O=gpurun_out/r02h; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
V=tools/bin/variants
REPEAT=3 WL="c3 c2" EXTRA="--no-verify" timeout 1800 tools/ab.sh $V/cur.so $V/cur_abl1.so $V/cur_abl2.so $V/cur_w3.so $V/cur_g3.so $V/cur_q64.so > $O/ab.txt 2>&1
timeout 600 tools/pmc_ab.sh c3 $V/cur.so $V/cur_abl2.so > $O/pmc_ab.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
tail -3 $O/pytest_gpu.txt; cat $O/ab.txt; cat $O/pmc_ab.txt
