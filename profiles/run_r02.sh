#!/bin/bash
# Round-2 profile set (one gpurun call): kernel-trace stats of the default bench command (2 sweeps in flight) and of
# the one-sweep-in-flight configuration, PMC counters of the headline kernel (separate passes, counters only with
# --kernel-trace), and stats + PMC of the other sweep variants (profiles/bench_paths.py).
# usage (GPU box, repo root):  bash profiles/run_r02.sh gpurun_out/r02
set -e
O=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $O
B="--steps 60 --warmup 10 --no-cpu-baseline --no-pcie"
stats() { n=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -- "$@" > $O/$n.out 2> $O/$n.err; echo "$n rc=$?"; }
pmc() { n=$1; c=$2; shift 2; timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$n -- "$@" > /dev/null 2> $O/$n.err; echo "$n rc=$?"; }
stats bench_streams2 python3 bench.py $B --streams 2
stats bench_streams1 python3 bench.py $B --streams 1
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
A="--steps 3 --warmup 1 --no-cpu-baseline --no-pcie --streams 1"
pmc p1 "$P1" python3 bench.py $A
pmc p2 "$P2" python3 bench.py $A
pmc p3 "FETCH_SIZE GRBM_GUI_ACTIVE" python3 bench.py $A
pmc p4 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" python3 bench.py $A
python3 profiles/pmc_summary.py $O > $O/pmc_summary_headline.txt
stats paths python3 profiles/bench_paths.py cfg2 cfg4 car order3 order1
pmc q1 "$P1" python3 profiles/bench_paths.py cfg2 cfg4 car order3
pmc q2 "$P2" python3 profiles/bench_paths.py cfg2 cfg4 car order3
pmc q3 "FETCH_SIZE GRBM_GUI_ACTIVE" python3 profiles/bench_paths.py cfg2 cfg4 car order3
pmc q4 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" python3 profiles/bench_paths.py cfg2 cfg4 car order3
mkdir -p $O/qsum && cp -r $O/q1 $O/q2 $O/q3 $O/q4 $O/qsum/ && python3 profiles/pmc_summary.py $O/qsum > $O/pmc_summary_paths.txt
echo done
