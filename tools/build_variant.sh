#!/bin/bash
# An A/B build of libmasp_hip.so that differs from the tree's build in the compile flags of some units:
#   tools/build_variant.sh <name> "<extra flags>" <unit> [<unit> ...]      -> tools/_build/ab/libmasp_hip_<name>.so
# (the other units' objects are taken from masp_amd/csrc/_build: run `make -C masp_amd/csrc` first).  The result travels to the GPU box
# (tools/_build/ab/ is not in .gpurunignore) and is selected with MASP_HIP_LIBRARY=... (masp_amd/hip.py), e.g. by tools/abn.sh.
set -e
name=$1; flags=$2; shift 2
root=$(cd $(dirname $0)/.. && pwd)
csrc=$root/masp_amd/csrc
out=$root/tools/_build/ab
mkdir -p $out/$name
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-value -Wno-unused-result"
objs=""
for u in prover k_msm_sort k_msm_g1 k_msm_g1_lone k_msm_g1_acc k_msm_g1_tree k_msm_g2 k_msm_g2_lone k_msm_g2_acc k_msm_g2_tree k_ntt k_groth16 k_setup k_verify; do
  if [[ " $* " == *" $u "* ]]; then
    (cd $csrc && /opt/rocm/bin/hipcc $FLAGS $flags -c $u.hip -o $out/$name/$u.o) &
    objs="$objs $out/$name/$u.o"
  else
    objs="$objs $csrc/_build/$u.o"
  fi
done
wait
/opt/rocm/bin/hipcc $FLAGS -shared $objs -Wl,--version-script=$csrc/exports_hip.map -o $out/libmasp_hip_$name.so
ls -la $out/libmasp_hip_$name.so
