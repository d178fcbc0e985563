#!/bin/bash
# build_variant.sh NAME [extra hipcc flags...]: builds normalizing-flows_amd/variants/libnf_hip_NAME.so from
# the current sources with extra flags, for A/B runs through NF_HIP_LIB.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/normalizing-flows_amd/csrc
tmp=/tmp/nf_variant_$name
mkdir -p $tmp $root/normalizing-flows_amd/variants
base="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -I$src"
for f in nf_api nf_gemm nf_maf nf_resnet nf_hidden; do /opt/rocm/bin/hipcc $base "$@" -c $src/$f.hip -o $tmp/$f.o & done
for f in nf_spline nf_fused nf_inverse nf_coupling; do /opt/rocm/bin/hipcc $base -ffp-contract=off "$@" -c $src/$f.hip -o $tmp/$f.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/normalizing-flows_amd/variants/libnf_hip_$name.so $tmp/*.o
echo built $name
