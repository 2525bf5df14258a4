#!/bin/bash
# instruction counts of the final build and its ablation stages (stream + level 1; + list + level 2; everything)
O=gpurun_out/r02pmcfinal; mkdir -p $O; export TMPDIR=/tmp
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
V=tools/bin/variants
timeout 900 tools/pmc_mini.sh $V/final_abl1.so $V/final_abl2.so $V/final.so > $O/pmc.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
cat $O/pmc.txt
