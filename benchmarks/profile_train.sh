#!/bin/bash
# profile_train.sh OUTDIR DTYPE -- bench_train line + rocprofv3 kernel stats of the same command
out=$1; dt=${2:-f16x2}
mkdir -p $out
python3 $GRAFT_REPO_ROOT/benchmarks/bench_train.py --dtype $dt --steps 5 > $out/train_$dt.json 2> $out/train_$dt.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/train_trace_$dt -- python3 $GRAFT_REPO_ROOT/benchmarks/bench_train.py --dtype $dt --steps 3 --warmup 1 > $out/train_trace_$dt.log 2>&1 || exit 1
f=$(ls $out/train_trace_$dt/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY' > $out/train_kernel_summary_$dt.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:22]:
    print('%6.2f%%  calls %5s  avg %9.1f us  %s' % (100 * float(r['TotalDurationNs']) / tot, r['Calls'], float(r['AverageNs']) / 1e3, r['Name'].replace('(anonymous namespace)::', '')[:110]))
PY
