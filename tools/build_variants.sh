#!/bin/bash
# tools/build_variants.sh name[:flags] ...  -- cross-compile kernel-module variants into tools/bin/variants/<name>.so
# (git-ignored, travels to the GPU box).  "base" = the module of the last commit (git show HEAD:...).
# Example: tools/build_variants.sh base new front4:-DPFAC_FRONT_LOG2=4
cd "$(dirname "$0")/.."; mkdir -p tools/bin/variants
for v in "$@"; do
  name=${v%%:*}; flags=""; [ "$v" != "$name" ] && flags=${v#*:}
  src=pfac_amd/csrc/scan_gfx950.hip
  if [ "$name" = base ]; then mkdir -p /tmp/pfac_base; git show HEAD:pfac_amd/csrc/scan_gfx950.hip > /tmp/pfac_base/scan_gfx950.hip; src=/tmp/pfac_base/scan_gfx950.hip; fi
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Ipfac_amd/csrc $flags -shared -o tools/bin/variants/$name.so $src 2>&1 | grep -E "error|Error" ; echo "built $name ($flags)" ) &
done
wait
