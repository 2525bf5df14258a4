#!/bin/bash
# tools/build_variant.sh <name> <rev|WORK> [extra hipcc flags ...]
#   cross-compiles the kernel module AND the host library of one variant (the extra flags, -D..., go to both) into tools/bin/variants/<name>/ (git-ignored,
#   travels to the GPU box).  rev = a git revision (sources taken from `git archive`), WORK = the working tree.
#   Example: tools/build_variant.sh base HEAD ; tools/build_variant.sh noladder WORK -DPFAC_ABLATE=2
set -e
cd "$(dirname "$0")/.."
name=$1; rev=$2; shift 2
out=tools/bin/variants/$name; mkdir -p $out
src=$PWD
if [ "$rev" != WORK ]; then
  src=/tmp/pfac_variant_$name; rm -rf $src; mkdir -p $src
  git archive $rev pfac_amd/csrc include | tar -x -C $src
fi
units="$src/pfac_amd/csrc/scan_filter.hip $src/pfac_amd/csrc/scan_tiled.hip $src/pfac_amd/csrc/scan_module.hip"
[ -f $src/pfac_amd/csrc/scan_gfx950.hip ] && units=$src/pfac_amd/csrc/scan_gfx950.hip          # revisions before the module was split into units
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -I$src/include -I$src/pfac_amd/csrc "$@" -shared -o $out/libpfac_gfx950.so $units 2>&1 | grep -E "error" || true
( cd $src/pfac_amd/csrc && g++ -O2 -std=c++17 -fPIC -fopenmp -D__HIP_PLATFORM_AMD__ "$@" -I/opt/rocm/include -I../../include -I. -shared \
    -o $OLDPWD/$out/libpfac.so $(ls pfac_api.cpp host_pipeline.cpp multi_gpu.cpp compiled_set.cpp pattern_compiler.cpp tables.cpp cpu_engine.cpp 2>/dev/null) -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lamdhip64 -ldl )
echo "built $name from $rev ($*)"; ls -la $out
