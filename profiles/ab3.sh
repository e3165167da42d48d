#!/bin/bash
# usage: bash profiles/ab3.sh <outdir> "<variants>"   -- headline (60 x 60 lags) and a 21 x 21 lag set (few batches)
O=$1; V=$2; mkdir -p $O
for rep in 1 2; do
  for v in $V; do
    COREG_HIP_LIB=$PWD/build/variants/$v.so timeout -k 10 300 python3 profiles/tune.py "" "" "" > $O/$v.a$rep.log 2> $O/$v.a$rep.err || exit 1
    COREG_HIP_LIB=$PWD/build/variants/$v.so TUNE_NLAG=21 timeout -k 10 300 python3 profiles/tune.py "" "" "" > $O/$v.b$rep.log 2> $O/$v.b$rep.err || exit 1
    echo "$v 60x60: $(awk '{print $3}' $O/$v.a$rep.log | tr '\n' ' ')  21x21: $(awk '{print $3}' $O/$v.b$rep.log | tr '\n' ' ')"
  done
done
