#!/bin/bash
O=gpurun_out/r02stress; mkdir -p $O; export TMPDIR=/tmp
timeout 2400 python tools/stress_fuzz.py ${1:-2000} ${2:-300} > $O/stress2.txt 2>&1; echo "rc $?" >> $O/stress2.txt
tail -5 $O/stress2.txt
