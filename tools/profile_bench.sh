#!/bin/bash
# rocprofv3 passes over the default bench command (run on the GPU box via gpurun):
#   1. --kernel-trace --stats       -> per-kernel time summary
#   2. --kernel-trace --pmc FETCH_SIZE   (own pass)
#   3. --kernel-trace --pmc WRITE_SIZE   (own pass)
# Outputs go to gpurun_out/prof_<tag>/ ; tools/summarize_profile.py turns them into
# profiles/<tag>_*.{csv,json}.
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-ttr --no-plain-loop --no-mirror --no-shapes --strong-grid= $BENCH_ARGS"   # BENCH_ARGS: e.g. "--strong 512,512,64" (the slab of the multi-GPU runs)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py $ARGS > $OUT/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py $ARGS > $OUT/fetch.log 2>&1 || exit 2
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/bench.py $ARGS > $OUT/write.log 2>&1 || exit 3
echo "profiles written under $OUT"
