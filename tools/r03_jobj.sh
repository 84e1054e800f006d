#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python -m pytest tests/ -x -q -m gpu > $ROOT/gpurun_out/r03_gputests_j.txt 2>&1
echo "pytest rc=$?"
tail -3 $ROOT/gpurun_out/r03_gputests_j.txt
python3 $ROOT/bench.py > $ROOT/gpurun_out/r03_bench_j.json 2> $ROOT/gpurun_out/r03_bench_j.err
echo "bench rc=$?"
