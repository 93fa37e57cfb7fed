#!/bin/bash
# Builds another copy of libbzk.so with extra compile-time switches for same-box A/B runs:
#   tools/build_variant.sh NAME -DBZK_G1_ACC_INLINE=0 ...   ->  bazuka_amd/libbzk.so.NAME   (objects in bazuka_amd/csrc/_obj_NAME)
# The pattern bazuka_amd/libbzk.so.* is git-ignored but travels to the GPU box; BZK_LIBBZK=<path> selects it (bazuka_amd/lib.py).
set -e
NAME=$1; shift
cd "$(dirname "$0")/../bazuka_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
O=_obj_$NAME; mkdir -p $O
pids=()
for f in *.hip; do
  $HIPCC -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-result -Wno-pass-failed "$@" -c $f -o $O/${f%.hip}.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC -shared -fPIC --offload-arch=gfx950 -o ../libbzk.so.$NAME $O/*.o
echo built bazuka_amd/libbzk.so.$NAME "$@"
