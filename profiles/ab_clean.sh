#!/bin/bash
# usage: bash profiles/ab_clean.sh <outdir>  -- the unmasked all-finite path (clean_path) against the masked one, on the
# headline sweep with and without NaN pixels in the image to align; the first configuration of a process is a warm-up
O=$1; mkdir -p $O
TUNE_NAN=0 timeout -k 10 300 python3 profiles/tune.py "" "clean_path=1" "clean_path=0" "clean_path=1" "clean_path=0" > $O/nan0.log 2> $O/nan0.err || exit 1
timeout -k 10 300 python3 profiles/tune.py "" "clean_path=1" "clean_path=0" "clean_path=1" "clean_path=0" > $O/nan005.log 2> $O/nan005.err || exit 1
TUNE_NAN=0 TUNE_NLAG=21 timeout -k 10 300 python3 profiles/tune.py "" "clean_path=1" "clean_path=0" "clean_path=1" "clean_path=0" > $O/nan0_21.log 2> $O/nan0_21.err || exit 1
cat $O/nan0.log $O/nan005.log $O/nan0_21.log
