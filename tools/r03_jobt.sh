#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r03_walk2d.txt
: > $OUT
python -m pytest tests/test_gpu_kernels.py -x -q -k "z_sweep or walk" > $ROOT/gpurun_out/r03_gputests_t.txt 2>&1
echo "pytest rc=$?" | tee -a $OUT
tail -4 $ROOT/gpurun_out/r03_gputests_t.txt | tee -a $OUT
python3 $ROOT/tools/walk2d_probe.py 2048 4096 6000 2>&1 | grep -v amdgpu | tee -a $OUT
