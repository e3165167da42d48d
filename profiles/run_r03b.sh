#!/bin/bash
# Round-3 multi-GPU evidence obtainable on a ONE-GPU box: per-rank work of N-way sharding (emulated), bench.py with two
# gloo ranks sharing the GPU (self-verifying line), the RCCL path with one rank, the in-library driver on two virtual
# devices, three sweeps in flight.   usage: bash profiles/run_r03b.sh gpurun_out/r03b
O=$1; mkdir -p $O
timeout -k 10 400 python3 profiles/slice_timing.py > $O/slice_timing.log 2> $O/slice_timing.err || { tail $O/slice_timing.err; exit 1; }
cat $O/slice_timing.log
COREG_BENCH_BACKEND=gloo COREG_CPU_CORES=8 timeout -k 10 500 python3 bench.py --gpus 2 --steps 100 --warmup 10 > $O/bench_n2_gloo_one_gpu.json 2> $O/n2.err || { tail $O/n2.err; exit 1; }
tail -c 900 $O/bench_n2_gloo_one_gpu.json; echo
COREG_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_dist1_rccl.json 2> $O/dist1.err || { tail $O/dist1.err; exit 1; }
tail -c 300 $O/bench_dist1_rccl.json; echo
COREG_VIRTUAL_DEVICES=2 timeout -k 10 300 python3 bench.py --gpus 2 --launch threads --steps 50 --warmup 5 > $O/bench_threads2_virtual.json 2> $O/thr.err || { tail $O/thr.err; exit 1; }
cat $O/bench_threads2_virtual.json | cut -c1-1500; echo
timeout -k 10 300 python3 bench.py --streams 3 --no-cpu-baseline --no-pcie > $O/bench_streams3.json 2> $O/s3.err || { tail $O/s3.err; exit 1; }
python3 -c "import json;d=json.load(open('$O/bench_streams3.json'));print('streams3', d['value'], d['ms_per_step'])"
timeout -k 10 300 python3 bench.py --streams 1 --no-cpu-baseline --no-pcie > $O/bench_streams1.json 2> $O/s1.err
python3 -c "import json;d=json.load(open('$O/bench_streams1.json'));print('streams1', d['value'], d['ms_per_step'])"
echo done
