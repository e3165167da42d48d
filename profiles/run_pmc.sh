#!/bin/bash
# Collect the sweep kernel's PMC counters (separate passes; counters only with --kernel-trace, as gpurun requires).
# usage (on the GPU box, from the repo root):  bash profiles/run_pmc.sh <outdir> [bench args...]
set -e
O=$1; shift
A="--steps 3 --warmup 1 --no-cpu-baseline --no-pcie --streams 1 $@"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $O
run() { n=$1; shift; timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -- python3 bench.py $A > /dev/null 2> $O/$n.err; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM
run p2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
run p3 FETCH_SIZE GRBM_GUI_ACTIVE
run p4 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 profiles/pmc_summary.py $O > $O/summary.txt
cat $O/summary.txt
