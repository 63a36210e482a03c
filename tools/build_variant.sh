#!/bin/bash
# Build a variant of libhppfcl_amd.so with one or more kernel units recompiled under extra flags (A/B runs on the GPU box select it
# with HFCL_LIB_PATH).  Usage: tools/build_variant.sh <name> <units: k_gjk|k_epa|k_bvh|k_bvhd|host, comma-separated> <flags...>
# Output: build/ab/lib_<name>.so (git-ignored; travels with gpurun).  The other objects are the in-tree ones (run make first).
set -e
name=$1; units=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/hpp-fcl_amd/csrc
mkdir -p $root/build/ab
objs=""
pids=""
for o in hfcl_host hfcl_k_gjk hfcl_k_epa hfcl_k_bvh hfcl_k_bvhd hfcl_k_util; do
  u=${o#hfcl_}
  if [[ ",$units," == *",$u,"* ]]; then
    unitflags=$(make -s -C $csrc -pn 2>/dev/null | sed -n "s/^FLAGS_$u = //p" | head -1)
    (cd $csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value $unitflags "$@" -Wno-pass-failed -c -o $root/build/ab/${o}_$name.o $o.hip) &
    pids="$pids $!"
    objs="$objs $root/build/ab/${o}_$name.o"
  else
    objs="$objs $csrc/$o.o"
  fi
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $root/build/ab/lib_$name.so $objs $csrc/hfcl_bvh_build.o $csrc/hfcl_broadphase.o -lpthread
rm -f $root/build/ab/hfcl_*_$name.o
echo built build/ab/lib_$name.so
