#!/bin/bash
# quick regression check of every sweep variant + the headline (one gpurun call)
O=$1; mkdir -p $O
timeout -k 10 300 python3 profiles/tune.py "taper_frac=0" "" "taper_frac=0" "" > $O/tune.log 2>$O/tune.err || exit 1
awk '{printf "%-20s %s  pre %s\n", $1, $3, $6}' $O/tune.log
timeout -k 10 400 python3 profiles/bench_paths.py cfg2 cfg4 car order3 order1 > $O/paths.log 2> $O/paths.err || exit 1
cat $O/paths.log | cut -c1-190
timeout -k 10 400 python3 profiles/bench_configs.py > $O/configs.log 2>&1; grep -v amdgpu $O/configs.log | cut -c1-200
