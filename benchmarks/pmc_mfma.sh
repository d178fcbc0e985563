#!/bin/bash
# pmc_mfma.sh OUTDIR [run_fused.py args]: matrix-pipe utilisation + clock of the fused kernel
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/p -- python3 $GRAFT_REPO_ROOT/benchmarks/run_fused.py "$@" > $out/p.log 2>&1 || { echo failed; tail -5 $out/p.log; exit 1; }
echo done
