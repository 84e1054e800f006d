#!/bin/bash
# round 3, plain-CSR stream kernel with short-lived workgroups: the CG step built on it, kernel tests
set -e
mkdir -p gpurun_out
python bench.py --steps 20 --no-ttr --no-cpu-baseline > gpurun_out/r03_bench_c1.json 2> gpurun_out/r03_bench_c1.err
SCHWZ_STREAM_NTY=2 python bench.py --steps 20 --no-ttr --no-cpu-baseline > gpurun_out/r03_bench_c2.json 2> gpurun_out/r03_bench_c2.err
SCHWZ_STREAM_SEQ=0 python bench.py --steps 20 --no-ttr --no-cpu-baseline > gpurun_out/r03_bench_c0.json 2> gpurun_out/r03_bench_c0.err
python -m pytest tests/test_gpu_kernels.py -x -q > gpurun_out/r03_gpukernels.txt 2>&1
tail -3 gpurun_out/r03_gpukernels.txt
