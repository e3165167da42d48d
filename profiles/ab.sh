#!/bin/bash
# A/B of library builds on one GPU box: headline kernel time (profiles/tune.py), the other sweep variants
# (profiles/bench_paths.py) for each build under build/variants/<name>.so.
# usage (GPU box, repo root): bash profiles/ab.sh gpurun_out/ab "base s1 s2" ["paths"]
O=$1; V=$2; P=$3
mkdir -p $O
for v in $V; do
  COREG_HIP_LIB=$PWD/build/variants/$v.so timeout -k 10 240 python3 profiles/tune.py "" "" > $O/$v.tune.log 2> $O/$v.tune.err || exit 1
  echo "== $v"; cat $O/$v.tune.log
done
if [ -n "$P" ]; then
  for v in $V; do
    COREG_HIP_LIB=$PWD/build/variants/$v.so timeout -k 10 400 python3 profiles/bench_paths.py cfg2 cfg4 car order3 order1 > $O/$v.paths.log 2> $O/$v.paths.err || exit 1
    echo "== $v paths"; cat $O/$v.paths.log
  done
fi
