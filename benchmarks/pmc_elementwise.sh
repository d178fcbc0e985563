#!/bin/bash
# pmc_elementwise.sh OUTDIR KERNEL [args] -- kernel trace + one rocprofv3 --pmc run per counter group on
# benchmarks/run_elementwise.py KERNEL
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/benchmarks/run_elementwise.py "$@" > $out/trace.log 2>&1 || { echo "trace failed"; exit 1; }
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/benchmarks/run_elementwise.py "$@" > $out/p$i.log 2>&1 || { echo "pass $i failed"; exit 1; }
done <<'PASSES'
SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_TRANS
TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
PASSES
echo done
