#!/bin/bash
# A/B build of the native library: tools/ab_build.sh <name> "<extra hipcc flags>" [units...]
#   units: the kernel translation units the flags apply to (gjk epa bvh host; default: gjk epa bvh).
# Objects of the other units are taken from the in-tree build.  Result: build/ab/libhppfcl_amd_<name>.so
# (use with HFCL_LIB_PATH=build/ab/libhppfcl_amd_<name>.so; build/ travels to the GPU box, it is git-ignored).
set -e
name=$1; flags=$2; shift 2 || true
units=${@:-gjk epa bvh}
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/hpp-fcl_amd/csrc
out=$root/build/ab/$name
mkdir -p "$out"
make -s -j8 -C "$src" >/dev/null
objs=""
pids=""
for u in host k_gjk k_epa k_bvh; do
  short=${u#k_}
  unit_flags=""  # the per-unit flags of the Makefile (FLAGS_k_gjk / FLAGS_k_epa)
  if [[ $short == gjk || $short == epa ]]; then unit_flags="-fno-slp-vectorize -fno-hip-fp32-correctly-rounded-divide-sqrt"; fi
  if [[ $short == epa ]]; then unit_flags="$unit_flags -ffp-contract=on"; fi
  if [[ " $units " == *" $short "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-pass-failed $unit_flags $flags -c -o "$out/hfcl_$u.o" "$src/hfcl_$u.hip" &
    pids="$pids $!"
    objs="$objs $out/hfcl_$u.o"
  else
    objs="$objs $src/hfcl_$u.o"
  fi
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$root/build/ab/libhppfcl_amd_$name.so" $objs "$src/hfcl_bvh_build.o" "$src/hfcl_broadphase.o" -lpthread
rm -rf "$out"
echo "build/ab/libhppfcl_amd_$name.so"
