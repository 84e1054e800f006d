#!/bin/bash
# round 3: every profile the bench line quotes, on one box: bench passes (cube, 512 x 512 x 64 slab, 1024 x 1024 x 128
# slab) and the plain-CSR SpMV per shape.  Summaries are made afterwards by tools/summarize_profile.py.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
set -x
bash $ROOT/tools/profile_bench.sh r03 || exit 1
BENCH_ARGS="--strong 512,512,64" bash $ROOT/tools/profile_bench.sh r03_slab || exit 2
BENCH_ARGS="--strong 1024,1024,128" bash $ROOT/tools/profile_bench.sh r03_c5slab || exit 3
bash $ROOT/tools/profile_plain.sh r03 || exit 4
du -sh $ROOT/gpurun_out/prof_r03*
# keep what the summaries need (the kernel traces are large)
find $ROOT/gpurun_out/prof_r03* -name "*kernel_trace.csv" -size +8M -delete
du -sh $ROOT/gpurun_out
