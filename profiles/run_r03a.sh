#!/bin/bash
# Round-3 check-point (one gpurun call): the GPU test-suite, the default bench line, a kernel trace of one-sweep-in-flight
# steps (for the gaps between the kernels of a sweep).   usage: bash profiles/run_r03a.sh gpurun_out/r3a
O=$1
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; rc=$?
tail -4 $O/pytest.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; rc=$?
tail -c 600 $O/bench.json
if [ $rc -ne 0 ]; then tail -20 $O/bench.err; exit $rc; fi
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-pcie > $R/$O/trace.out 2> $R/$O/trace.err
echo "trace rc=$?"
