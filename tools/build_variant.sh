#!/bin/bash
# Build a variant of libhppfcl_amd.so with one or more kernel units recompiled under extra flags (A/B runs on the GPU box select it
# with HFCL_LIB_PATH).  Usage: tools/build_variant.sh <name> <units, comma-separated> <flags...>
#   units: host k_gjk32 k_gjk64 k_epa32 k_epa64 k_bvh k_bvhc k_bvhs k_bvhd k_util   (k_gjk / k_epa: both precisions of the unit; k_bvhc / k_bvhs:
#   the mesh x solid collide() / distance() parts of hfcl_k_bvh.hip)
# Output: build/ab/lib_<name>.so (git-ignored; travels with gpurun).  The other objects are the in-tree ones (run make first).
set -e
name=$1; units=$2; shift 2
units=$(echo ",$units," | sed 's/,k_gjk,/,k_gjk32,k_gjk64,/; s/,k_epa,/,k_epa32,k_epa64,/')
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/hpp-fcl_amd/csrc
mkdir -p $root/build/ab
objs=""
pids=""
for u in host multi k_gjk32 k_gjk64 k_epa32 k_epa64 k_bvh k_bvhc k_bvhs k_bvhd k_util; do
  o=hfcl_$u
  if [[ "$units" == *",$u,"* ]]; then
    unitflags=$(make -s -C $csrc -pn 2>/dev/null | sed -n "s/^FLAGS_$u = //p" | head -1)
    src=$o; def=""
    case $u in *32) src=${o%32}; def="-DHFCL_UNIT_PRECISION=32";; *64) src=${o%64}; def="-DHFCL_UNIT_PRECISION=64";; k_bvhs|k_bvhc) src=hfcl_k_bvh;; esac
    (cd $csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value $unitflags $def "$@" -Wno-pass-failed -c -o $root/build/ab/${o}_$name.o $src.hip) &
    pids="$pids $!"
    objs="$objs $root/build/ab/${o}_$name.o"
  else
    objs="$objs $csrc/$o.o"
  fi
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $root/build/ab/lib_$name.so $objs $csrc/hfcl_bvh_build.o $csrc/hfcl_broadphase.o -lpthread -ldl
rm -f $root/build/ab/hfcl_*_$name.o
echo built build/ab/lib_$name.so
