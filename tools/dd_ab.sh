#!/bin/bash
# In-box A/B of the fused direction launch's build-time switches (SCHWZ_DD, spmv_pair.hip): library builds
# lib/libschwz_hip_dd<k>.so against the default one; cube, 512 x 512 x 64 slab, 1024-wide slab.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/dd_ab.txt
: > $OUT
LIBS=${LIBS:-"libschwz_hip.so libschwz_hip_dd1.so libschwz_hip_dd2.so libschwz_hip_dd3.so libschwz_hip_dd7.so libschwz_hip.so"}
for shape in cube slab wide; do
    case $shape in cube) ARGS="";; slab) ARGS="--strong 512,512,64";; wide) ARGS="--strong 1024,1024,16";; esac
    for L in $LIBS; do
        line=$(SCHWZ_HIP_LIB=$ROOT/schwarz-lib_amd/lib/$L python3 $ROOT/bench.py --steps 20 --warmup 3 --no-ttr --no-cpu-baseline --no-plain-loop --no-mirror --no-shapes --strong-grid "" $ARGS 2>/dev/null)
        echo "$shape $L $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms/step %.4f  update %.4f ms frac %.3f  dirdot %.4f ms frac %.3f  reduction %.17g" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline_spmv"]["avg_launch_ms"], d["roofline_spmv"]["frac"], d["residual_reduction_in_timed_steps"]))')" | tee -a $OUT
    done
done
