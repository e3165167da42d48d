#!/bin/bash
# Headline kernel time under different lag-patch / tile / pitch choices (profiles/tune.py), plans on stderr.
O=$1; mkdir -p $O
COREG_DEBUG_PLAN=1 timeout -k 10 600 python3 profiles/tune.py "" "patch_w=20" "patch_w=20,pitch=121" "patch_w=20,pitch=153" "patch_w=20,pitch=0" \
  "patch_w=12,pitch=153" "patch_w=15" "patch_w=16" "patch_w=30" "patch_w=60" "patch_w=10" "patch_w=20,tile_w=8" "patch_w=20,tile_w=16" \
  "patch_w=20,tile_w=32" "patch_w=20,tile_w=64" "patch_w=30,tile_w=16" "patch_w=60,tile_w=16" "" "patch_w=20" > $O/plan_sweep.log 2> $O/plan_sweep.err
cat $O/plan_sweep.log; grep "coreg plan" $O/plan_sweep.err | sort | uniq -c | sort -rn | head -40
