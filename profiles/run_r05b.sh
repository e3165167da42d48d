#!/bin/bash
# Round-5, second set (one gpurun call): the callers' timings (drop-in imager call, SPICE call, jitter session on one and
# on two logical devices), the partitions of multi-dimensional sweeps emulated per rank, and the multi-GPU code paths as
# far as ONE GPU allows (two gloo ranks sharing it, one RCCL rank, the in-library driver on two virtual devices).
# usage: bash profiles/run_r05b.sh gpurun_out/r05b
O=$1; mkdir -p $O
timeout -k 10 200 python3 profiles/api_timing.py > $O/api_timing.json 2> $O/api_timing.err || { tail $O/api_timing.err; exit 1; }
python3 -c "import json;d=json.load(open('$O/api_timing.json'));print({k:v for k,v in d.items() if not isinstance(v,dict)})"
timeout -k 10 200 python3 profiles/api_timing_spice.py > $O/api_timing_spice.json 2> $O/api_spice.err || { tail $O/api_spice.err; exit 1; }
python3 -c "import json;d=json.load(open('$O/api_timing_spice.json'));print(d['warm_calls'], d['first_call']['sweep_kernel_ms'])"
timeout -k 10 300 python3 profiles/jitter_bench.py > $O/jitter_bench.json 2> $O/jitter.err || { tail $O/jitter.err; exit 1; }
cat $O/jitter_bench.json; echo
COREG_VIRTUAL_DEVICES=2 timeout -k 10 300 python3 profiles/jitter_bench.py > $O/jitter_bench_virtual2.json 2> $O/jitter2.err || { tail $O/jitter2.err; exit 1; }
cat $O/jitter_bench_virtual2.json; echo
timeout -k 10 100 python3 profiles/copy_bench.py > $O/copy_bench.json 2>&1; cat $O/copy_bench.json
timeout -k 10 400 python3 profiles/partition_timing.py > $O/partition_timing.jsonl 2> $O/partition.err || { tail $O/partition.err; exit 1; }
cat $O/partition_timing.jsonl
timeout -k 10 400 python3 profiles/slice_timing.py > $O/slice_timing.log 2> $O/slice_timing.err || { tail $O/slice_timing.err; exit 1; }
cat $O/slice_timing.log
COREG_BENCH_BACKEND=gloo COREG_CPU_CORES=8 timeout -k 10 500 python3 bench.py --gpus 2 --steps 100 --warmup 10 > $O/bench_n2_gloo_one_gpu.json 2> $O/n2.err || { tail $O/n2.err; exit 1; }
tail -c 900 $O/bench_n2_gloo_one_gpu.json; echo
COREG_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_dist1_rccl.json 2> $O/dist1.err || { tail $O/dist1.err; exit 1; }
tail -c 300 $O/bench_dist1_rccl.json; echo
COREG_VIRTUAL_DEVICES=2 timeout -k 10 300 python3 bench.py --gpus 2 --launch threads --steps 50 --warmup 5 > $O/bench_threads2_virtual.json 2> $O/thr.err || { tail $O/thr.err; exit 1; }
cat $O/bench_threads2_virtual.json | cut -c1-1500; echo
COREG_BENCH_BACKEND=gloo COREG_CPU_CORES=8 timeout -k 10 500 python3 bench.py --gpus 2 --shard points --steps 50 --warmup 5 > $O/bench_n2_points_gloo_one_gpu.json 2> $O/n2p.err || { tail $O/n2p.err; exit 1; }
tail -c 600 $O/bench_n2_points_gloo_one_gpu.json; echo
timeout -k 10 120 python3 profiles/call_kernel_time.py > $O/call_kernel_time.log 2>&1; grep -v amdgpu $O/call_kernel_time.log
# closing fuzz campaign on the round's build: every order, every image form
timeout -k 10 900 python3 tests/deep_fuzz.py 2500 400000 1 1,2,3 1 > $O/deep_fuzz_400000_scale1.log 2>&1; tail -3 $O/deep_fuzz_400000_scale1.log
timeout -k 10 600 python3 tests/deep_fuzz.py 600 410000 3 1,2,3 1 > $O/deep_fuzz_410000_scale3.log 2>&1; tail -3 $O/deep_fuzz_410000_scale3.log
echo done
