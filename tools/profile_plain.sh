#!/bin/bash
# rocprofv3 passes over the plain-CSR SpMV on the three per-GPU shapes (cube of configs[1], slabs of configs[2] and
# configs[4]): --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE, each its own pass.
#   tools/profile_plain.sh <tag>      -> gpurun_out/prof_<tag>_plain_<shape>/{stats,fetch,write}
# tools/summarize_profile.py --plain gpurun_out/prof_<tag>_plain <tag> turns them into profiles/<tag>_plain_*.csv
# and the spmv_stream_kernel<0, entries of profiles/traffic.json.
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for SHAPE in ${SHAPES:-256,256,256 512,512,64 1024,1024,128}; do
    KEY=$(echo $SHAPE | tr ',' 'x')
    OUT=$ROOT/gpurun_out/prof_${TAG}_plain_$KEY
    rm -rf $OUT && mkdir -p $OUT
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/plain_spmv_run.py $SHAPE > $OUT/stats.log 2>&1 || exit 1
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/tools/plain_spmv_run.py $SHAPE 6 > $OUT/fetch.log 2>&1 || exit 2
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/tools/plain_spmv_run.py $SHAPE 6 > $OUT/write.log 2>&1 || exit 3
    echo "plain SpMV passes for $SHAPE written under $OUT"
done
