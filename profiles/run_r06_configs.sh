#!/bin/bash
# Round 6: per-kernel roofline of EVERY sweep variant on the current build.  For each config of
# `bench.py --config`: (1) rocprofv3 --kernel-trace --stats, (2) four --pmc passes (counters only with --kernel-trace, as
# gpurun requires; separate passes, MI355X_MICROARCH.md), (3) the bench line itself, which then reads (1) and (2) back.
# usage (GPU box, repo root):  bash profiles/run_r06_configs.sh gpurun_out/r06cfg [configs...]
O=$1; shift
CFGS=${@:-"order1 order3 cfg3 cfg2 cfg4 car cfg5"}
R=$PWD
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() { local n=$1; shift; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/$n -- "$@" > $R/$O/$n.out 2> $R/$O/$n.err; echo "$n rc=$?"; }
pmc() { local n=$1; local ctr=$2; shift 2; mkdir -p $R/$O/$(dirname $n); timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/$O/$n -- "$@" > /dev/null 2> $R/$O/$n.err; echo "$n rc=$?"; }
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
for c in $CFGS; do
  S=20; [ $c = cfg5 ] && S=1
  if [ -z "$SKIP_STATS" ]; then
    stats stats_$c python3 $R/bench.py --config $c --steps $S --warmup 2 --no-cpu-baseline || exit 1
    f=$(ls $R/$O/stats_$c/*/*kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$f" ] && cp $f $R/profiles/r06_kernel_stats_$c.csv
  fi
  A="--config $c --steps 2 --warmup 1 --no-cpu-baseline"; [ $c = cfg5 ] && A="--config $c --steps 1 --warmup 0 --no-cpu-baseline"
  pmc pmc_$c/p1 "$P1" python3 $R/bench.py $A || exit 1
  pmc pmc_$c/p2 "$P2" python3 $R/bench.py $A || exit 1
  pmc pmc_$c/p3 "FETCH_SIZE GRBM_GUI_ACTIVE" python3 $R/bench.py $A || exit 1
  pmc pmc_$c/p4 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" python3 $R/bench.py $A || exit 1
  python3 $R/profiles/pmc_summary.py $R/$O/pmc_$c > $R/profiles/r06_pmc_$c.txt
  cp $R/profiles/r06_pmc_$c.txt $R/profiles/r06_kernel_stats_$c.csv $R/$O/ 2>/dev/null
  (cd $R && timeout -k 10 400 python3 bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; echo "bench $c rc=$?"; tail -c 1500 $O/bench_$c.json; echo)
done
echo done
