mkdir -p gpurun_out/r05c
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py "tests/test_gpu_fullsize.py::test_degenerate_overlaps_everywhere_stay_bounded_at_full_size" tests/test_gpu_upload.py -q -m gpu --no-header -p no:cacheprovider -s 2>&1 | grep -E "passed|failed|FAILED|degenerate sweep"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05c/s1 -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-pcie --no-all-finite --sustained-seconds 0 --streams 1 > /dev/null 2>&1
head -7 $R/gpurun_out/r05c/s1/*/*kernel_stats.csv | cut -c1-140
cd $R; python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['one_sweep_in_flight']['ms_per_step'], d['pcie_inclusive']['ms_per_step'], d['roofline']['kernel_ms'])"
