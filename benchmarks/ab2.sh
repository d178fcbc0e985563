#!/bin/bash
# ab2.sh NAME...: fused-kernel time of bench.py (headline mode) for each variant library, base = the in-tree library
for n in "$@"; do
  lib=$PWD/normalizing-flows_amd/variants/libnf_hip_$n.so
  [ "$n" = base ] && lib=$PWD/normalizing-flows_amd/libnf_hip.so
  NF_HIP_LIB=$lib timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu --no-bf16 > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err || { echo "$n failed"; tail -3 gpurun_out/ab_$n.err; exit 1; }
  python - "$n" <<'PY'
import json, sys
d = json.load(open('gpurun_out/ab_%s.json' % sys.argv[1]))
print(sys.argv[1], 'samples/s', round(d['value']), 'ms/step %.2f' % d['ms_per_step'], 'err %.2e' % d.get('parity_max_rel_err_vs_oracle', -1), {k: v for k, v in d['kernel_ms'].items() if 'made_rqs' in k})
PY
done
