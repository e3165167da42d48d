#!/bin/bash
# A/B of library builds x option strings on one GPU box.  usage: bash profiles/ab2.sh <outdir> "<variants>" "<cfg1>|<cfg2>|..." [paths]
O=$1; V=$2; IFS='|' read -ra CFGS <<< "$3"; P=$4
mkdir -p $O
for rep in 1 2; do
  for v in $V; do
    COREG_HIP_LIB=$PWD/build/variants/$v.so timeout -k 10 300 python3 profiles/tune.py "${CFGS[@]}" > $O/$v.tune$rep.log 2> $O/$v.tune$rep.err || { tail -5 $O/$v.tune$rep.err; exit 1; }
    echo "== $v (rep $rep)"; cat $O/$v.tune$rep.log
  done
done
if [ -n "$P" ]; then
  for v in $V; do
    COREG_HIP_LIB=$PWD/build/variants/$v.so timeout -k 10 400 python3 profiles/bench_paths.py cfg2 cfg4 order1 > $O/$v.paths.log 2> $O/$v.paths.err || exit 1
    echo "== $v paths"; cat $O/$v.paths.log
  done
fi
