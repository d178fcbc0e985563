#!/bin/bash
# pmc_passes.sh OUTDIR -- one rocprofv3 --pmc run per counter group on benchmarks/run_fused.py
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/benchmarks/run_fused.py "$@" > $out/p$i.log 2>&1 || { echo "pass $i failed"; exit 1; }
done <<'PASSES'
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_READ_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum GRBM_GUI_ACTIVE
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY
PASSES
echo done
