for m in 0.0625 64 256; do for v in filter; do echo "== $m MiB $v"; bash tools/call_trace.sh $m $v; done; done
echo "== 16 MiB auto"; bash tools/call_trace.sh 16 auto
echo "== 0.0625 MiB auto"; bash tools/call_trace.sh 0.0625 auto
