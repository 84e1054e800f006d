#!/bin/bash
# round 3: whole GPU suite, then the default bench line with its new legs
mkdir -p gpurun_out
python -m pytest tests/ -x -q -m gpu > gpurun_out/r03_gputests_d.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r03_gputests_d.txt
tail -4 gpurun_out/r03_gputests_d.txt
( time python bench.py ) > gpurun_out/r03_bench_d.json 2> gpurun_out/r03_bench_d.err
echo "bench rc=$?"
tail -c 600 gpurun_out/r03_bench_d.err
