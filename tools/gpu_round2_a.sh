#!/bin/bash
# first GPU call of round 2: VALU probe, parity tests on the new module, A/B of module variants, new bench paths
O=gpurun_out/r02a; mkdir -p $O; export TMPDIR=/tmp
timeout 120 tools/bin/valu_probe > $O/valu_probe.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
cp pfac_amd/lib/libpfac_gfx950.so /tmp/keep.so
REPEAT=3 WL="c3 c2" timeout 1200 tools/ab.sh tools/bin/variants/base.so tools/bin/variants/new.so tools/bin/variants/front4.so tools/bin/variants/front6.so tools/bin/variants/front8.so > $O/ab.txt 2>&1
cp /tmp/keep.so pfac_amd/lib/libpfac_gfx950.so
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc $?" >> $O/bench_default.err
timeout 300 python bench.py --gpus 2 --dist-backend gloo --no-other-configs --pmc off > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "rc $?" >> $O/bench_2rank_gloo.err
tail -3 $O/pytest_gpu.txt; cat $O/ab.txt; cat $O/valu_probe.txt
