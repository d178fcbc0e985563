#!/bin/bash
# profile_round.sh TAG: the profiles the round's numbers are quoted from (run on the GPU box through gpurun).
#   kernel trace + stats of bench.py, then separate --pmc passes (HBM traffic, matrix-pipe utilisation).
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B --steps 3 --warmup 1 --no-cpu > $out/bench_profiled.json 2> $out/trace.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- $B --steps 1 --warmup 1 --no-cpu --no-bf16 > $out/fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- $B --steps 1 --warmup 1 --no-cpu --no-bf16 > $out/write.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/mfma -- $B --steps 1 --warmup 1 --no-cpu --no-bf16 > $out/mfma.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT && timeout -k 10 300 python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1

# summaries (text) beside the raw output
python3 profiles/summarize_trace.py $(ls $out/trace/*/*kernel_trace.csv | head -1) 0.3 > $out/kernel_summary.txt 2>&1
for p in fetch write mfma; do python3 benchmarks/pmc_summary.py $out/$p made_rqs > $out/pmc_$p.txt 2>&1; done
echo profiled
