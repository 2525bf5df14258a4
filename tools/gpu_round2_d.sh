#!/bin/bash
O=gpurun_out/r02d; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
REPEAT=2 WL="c3" EXTRA="--no-verify" timeout 1500 tools/ab.sh tools/bin/variants/new.so tools/bin/variants/abl1.so tools/bin/variants/abl2.so > $O/ab.txt 2>&1
timeout 600 tools/pmc_ab.sh c3 tools/bin/variants/new.so tools/bin/variants/abl1.so tools/bin/variants/abl2.so > $O/pmc_ab.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
tail -3 $O/pytest_gpu.txt; cat $O/ab.txt; cat $O/pmc_ab.txt
