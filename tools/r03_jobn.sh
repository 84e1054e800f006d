#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python -m pytest tests/ -x -q -m gpu > $ROOT/gpurun_out/r03_gputests_n.txt 2>&1
echo "pytest rc=$?"
tail -3 $ROOT/gpurun_out/r03_gputests_n.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 $ROOT/bench.py > $ROOT/gpurun_out/r03_bench_n.json 2> $ROOT/gpurun_out/r03_bench_n.err
echo "bench rc=$?"
python3 $ROOT/bench.py --size 192 --no-mirror --no-shapes --no-cpu-baseline --no-ttr --strong-grid= > $ROOT/gpurun_out/r03_bench_192.json 2>/dev/null
python3 $ROOT/bench.py --size 320 --no-mirror --no-shapes --no-cpu-baseline --no-ttr --strong-grid= > $ROOT/gpurun_out/r03_bench_320.json 2>/dev/null
