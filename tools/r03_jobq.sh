#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python -m pytest tests/ -x -q -m gpu > $ROOT/gpurun_out/r03_gputests_q.txt 2>&1
echo "pytest rc=$?"
tail -3 $ROOT/gpurun_out/r03_gputests_q.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 $ROOT/bench.py > $ROOT/gpurun_out/r03_bench_q.json 2> $ROOT/gpurun_out/r03_bench_q.err
echo "bench rc=$?"
