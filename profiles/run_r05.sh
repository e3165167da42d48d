#!/bin/bash
# Round-5 profile set (one gpurun call): the GPU test-suite, the default bench line, kernel-trace stats of the default
# bench command (2 sweeps in flight) and of the one-sweep-in-flight configuration, PMC counters of the headline kernel
# (separate passes, counters only with --kernel-trace), the other sweep variants and BASELINE configs.
# usage (GPU box, repo root):  bash profiles/run_r05.sh gpurun_out/r05
O=$1
R=$PWD
mkdir -p $O
if [ -z "$PROFILE_ONLY" ]; then  # (PROFILE_ONLY=1: only the rocprofv3 passes)
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
tail -c 400 $O/bench.json; echo
fi
cd /tmp && export TMPDIR=/tmp
B="--steps 60 --warmup 10 --no-cpu-baseline --no-pcie --no-all-finite --sustained-seconds 0"
stats() { n=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/$n -- "$@" > $R/$O/$n.out 2> $R/$O/$n.err; echo "$n rc=$?"; }
pmc() { n=$1; c=$2; shift 2; timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/$n -- "$@" > /dev/null 2> $R/$O/$n.err; echo "$n rc=$?"; }
stats bench_streams2 python3 $R/bench.py $B --streams 2 || exit 1
stats bench_streams1 python3 $R/bench.py $B --streams 1 || exit 1
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
A="--steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-all-finite --streams 1 --sustained-seconds 0"
pmc p1 "$P1" python3 $R/bench.py $A || exit 1
pmc p2 "$P2" python3 $R/bench.py $A || exit 1
pmc p3 "FETCH_SIZE GRBM_GUI_ACTIVE" python3 $R/bench.py $A || exit 1
pmc p4 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" python3 $R/bench.py $A || exit 1
cd $R
mkdir -p $O/headline && cp -r $O/p1 $O/p2 $O/p3 $O/p4 $O/headline/ && python3 profiles/pmc_summary.py $O/headline > $O/pmc_summary_headline.txt
cat $O/pmc_summary_headline.txt
[ -n "$PROFILE_ONLY" ] && { echo done; exit 0; }
cd /tmp
stats paths python3 $R/profiles/bench_paths.py cfg2 cfg4 car order3 order1
cd $R
timeout -k 10 300 python3 profiles/bench_paths.py cfg2 cfg4 car order3 order1 > $O/paths.log 2> $O/paths.err; cat $O/paths.log
timeout -k 10 300 python3 profiles/pcie_breakdown.py > $O/pcie_breakdown.log 2>&1; cat $O/pcie_breakdown.log
timeout -k 10 300 python3 profiles/bench_configs.py > $O/configs.log 2>&1; tail -12 $O/configs.log
timeout -k 10 300 python3 profiles/tap_fix_timing.py > $O/tap_fix_timing.jsonl 2> /dev/null
timeout -k 10 300 python3 profiles/api_timing.py > $O/api_timing.json 2> $O/api_timing.err; tail -c 600 $O/api_timing.json; echo
# the sweep kernel INSIDE an isolated drop-in call (clocks fall while the GPU idles between calls): kernel-trace of the call
cd /tmp
stats api_call python3 $R/profiles/api_timing.py
cd $R
echo done
