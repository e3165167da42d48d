#!/bin/bash
# Round-6 final profile set of the HEADLINE (one gpurun call): the GPU test-suite, the default bench line, kernel-trace
# stats of the default bench command with one and two sweeps in flight, PMC counters of the headline kernel (separate
# passes, counters only with --kernel-trace), and the VALU count of cfg2 with the incremental homography path off.
# The other variants: profiles/run_r06_configs.sh.
# usage (GPU box, repo root):  bash profiles/run_r06.sh gpurun_out/r06final
O=$1
R=$PWD
mkdir -p $O
if [ -z "$PROFILE_ONLY" ]; then
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
tail -c 400 $O/bench.json; echo
fi
cd /tmp && export TMPDIR=/tmp
B="--steps 60 --warmup 10 --no-cpu-baseline --no-pcie --no-all-finite --sustained-seconds 0"
stats() { local n=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/$n -- "$@" > $R/$O/$n.out 2> $R/$O/$n.err; echo "$n rc=$?"; }
pmc() { local n=$1; local ctr=$2; shift 2; mkdir -p $R/$O/$(dirname $n); timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/$O/$n -- "$@" > /dev/null 2> $R/$O/$n.err; echo "$n rc=$?"; }
stats bench_streams2 python3 $R/bench.py $B --streams 2 || exit 1
stats bench_streams1 python3 $R/bench.py $B --streams 1 || exit 1
for s in 1 2; do f=$(ls $R/$O/bench_streams$s/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $R/$O/r06_kernel_stats_streams$s.csv; done
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
A="--steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-all-finite --streams 1 --sustained-seconds 0"
pmc headline/p1 "$P1" python3 $R/bench.py $A || exit 1
pmc headline/p2 "$P2" python3 $R/bench.py $A || exit 1
pmc headline/p3 "FETCH_SIZE GRBM_GUI_ACTIVE" python3 $R/bench.py $A || exit 1
pmc headline/p4 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" python3 $R/bench.py $A || exit 1
python3 $R/profiles/pmc_summary.py $R/$O/headline > $R/$O/r06_pmc_summary.txt
cat $R/$O/r06_pmc_summary.txt
# cfg2 with the per-sample homography (h_incr = 0): VALU instructions per wave-sample before round 6's path
COREG_BENCH_OPTS="h_incr=0" pmc cfg2_h_incr0/p1 "$P1" python3 $R/bench.py --config cfg2 --steps 2 --warmup 1 --no-cpu-baseline || exit 1
python3 $R/profiles/pmc_summary.py $R/$O/cfg2_h_incr0 > $R/$O/r06_pmc_cfg2_h_incr0.txt
cat $R/$O/r06_pmc_cfg2_h_incr0.txt
# the bench line once more, now that this run's own summaries exist for it to read back
cp $R/$O/r06_kernel_stats_streams1.csv $R/$O/r06_pmc_summary.txt $R/profiles/ 2>/dev/null
cd $R
timeout -k 10 400 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench_final rc=$?"; tail -c 300 $O/bench_final.json; echo
echo done
