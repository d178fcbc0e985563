#!/bin/bash
# profile_coupling.sh OUTDIR -- bench_coupling line + rocprofv3 kernel stats of the same command (cfg-2)
out=$1
mkdir -p $out
python3 $GRAFT_REPO_ROOT/benchmarks/bench_coupling.py --steps 10 > $out/coupling.json 2> $out/coupling.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/coupling_trace -- python3 $GRAFT_REPO_ROOT/benchmarks/bench_coupling.py --steps 5 --no-cpu > $out/coupling_trace.log 2>&1 || exit 1
f=$(ls $out/coupling_trace/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY' > $out/coupling_kernel_summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:14]:
    print('%6.2f%%  calls %5s  avg %9.1f us  %s' % (100 * float(r['TotalDurationNs']) / tot, r['Calls'], float(r['AverageNs']) / 1e3, r['Name'].replace('(anonymous namespace)::', '')[:110]))
PY
