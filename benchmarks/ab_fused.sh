#!/bin/bash
# ab_fused.sh NAME...: benchmarks/ab_fused.py for each variant library, one process each, in the order given ("base" =
# the in-tree library).  Put base first AND last: the spread between the two is the run-to-run noise of this box.
for n in "$@"; do
  lib=$PWD/normalizing-flows_amd/variants/libnf_hip_$n.so
  [ "$n" = base ] && lib=$PWD/normalizing-flows_amd/libnf_hip.so
  NF_HIP_LIB=$lib timeout -k 10 120 python -u benchmarks/ab_fused.py ${AB_SECONDS:-2.5} ${AB_MODE:-f16x2} 2>> gpurun_out/ab_fused.err | tee -a gpurun_out/ab_fused.jsonl || { echo "$n failed"; exit 1; }
done
