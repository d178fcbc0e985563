#!/bin/bash
# ab_x3.sh NAME...: like ab.sh, in bf16x3 mode
for n in "$@"; do
  lib=$PWD/normalizing-flows_amd/variants/libnf_hip_$n.so
  [ "$n" = base ] && lib=$PWD/normalizing-flows_amd/libnf_hip.so
  NF_HIP_LIB=$lib timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu --no-bf16 --dtype ${NF_MODE:-bf16x3} > gpurun_out/abx_$n.json 2> gpurun_out/abx_$n.err || { echo "$n failed"; tail -5 gpurun_out/abx_$n.err; exit 1; }
  python - "$n" <<'PY'
import json, sys
d = json.load(open('gpurun_out/abx_%s.json' % sys.argv[1]))
print(sys.argv[1], round(d['value']), d['kernel_ms'])
PY
done
