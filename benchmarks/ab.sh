#!/bin/bash
# ab.sh NAME...: bench.py fused-kernel time for each variant library (normalizing-flows_amd/variants/libnf_hip_NAME.so)
for n in "$@"; do
  lib=$PWD/normalizing-flows_amd/variants/libnf_hip_$n.so
  [ "$n" = base ] && lib=$PWD/normalizing-flows_amd/libnf_hip.so
  NF_HIP_LIB=$lib timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu --no-bf16 > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err || { echo "$n failed"; exit 1; }
  python - "$n" <<'PY'
import json, sys
d = json.load(open('gpurun_out/ab_%s.json' % sys.argv[1]))
print(sys.argv[1], round(d['value']), 'frac %.4f' % d['roofline']['frac'], d['kernel_ms'])
PY
done
