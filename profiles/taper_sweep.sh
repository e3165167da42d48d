#!/bin/bash
# tapered tile-group shares against equal shares for launches of 15 / 8 / 4 / 2 rounds of workgroups
O=$1; mkdir -p $O
for n in 60 45 32 21; do
  TUNE_NLAG=$n timeout -k 10 300 python3 profiles/tune.py "taper_frac=0" "taper_frac=0" "taper_frac=512" "taper_frac=384,taper_min=256" "taper_frac=768,taper_min=128" "taper_frac=256,taper_min=256" "taper_frac=0" "taper_frac=512" "" > $O/n$n.log 2> $O/n$n.err || { tail $O/n$n.err; exit 1; }
  echo "== $n x $n lags"; awk '{printf "%-34s %s\n", $1, $3}' $O/n$n.log
done
