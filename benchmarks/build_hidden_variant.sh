#!/bin/bash
# build_hidden_variant.sh NAME [extra hipcc flags...]: only nf_hidden.hip is recompiled with the extra flags; the other
# objects are the in-tree ones (run `make -C normalizing-flows_amd/csrc` first).
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/normalizing-flows_amd/csrc
mkdir -p /tmp/nf_hv_$name $root/normalizing-flows_amd/variants
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -Wno-pass-failed -I$src "$@" \
    -c $src/nf_hidden.hip -o /tmp/nf_hv_$name/nf_hidden.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/normalizing-flows_amd/variants/libnf_hip_$name.so \
    /tmp/nf_hv_$name/nf_hidden.o $src/nf_api.o $src/nf_spline.o $src/nf_gemm.o $src/nf_maf.o $src/nf_resnet.o $src/nf_inverse.o $src/nf_fused.o $src/nf_coupling.o
echo built $name
