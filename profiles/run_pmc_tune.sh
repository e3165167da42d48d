#!/bin/bash
# SQ counters of the sweep kernel under a tune.py option string.  usage: bash profiles/run_pmc_tune.sh <outdir> "<opts>"
set -e
O=$1; CFG=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $O
run() { n=$1; shift; timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -- python3 profiles/tune.py "$CFG" > $O/$n.out 2> $O/$n.err; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM
run p2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
python3 profiles/pmc_summary.py $O > $O/summary.txt
cat $O/p1.out
cat $O/summary.txt
