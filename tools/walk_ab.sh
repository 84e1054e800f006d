#!/bin/bash
# In-box A/B of library builds (make variant NAME=.. DEFS=..): cube, 512 x 512 x 64 slab, optionally the configs[4] slab.
#   LIBS="libschwz_hip.so libschwz_hip_zc0.so" SHAPES="cube slab" bash tools/walk_ab.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/walk_ab.txt
: > $OUT
LIBS=${LIBS:-"libschwz_hip.so libschwz_hip_zc0.so libschwz_hip.so libschwz_hip_zc0.so"}
for shape in ${SHAPES:-cube slab}; do
    case $shape in cube) ARGS="";; slab) ARGS="--strong 512,512,64";; wide) ARGS="--strong 1024,1024,16";; c5) ARGS="--strong 1024,1024,128";; *) ARGS="--size $shape";; esac
    for LE in $LIBS; do
        L=${LE%%:*}                      # an entry may carry one environment setting: libschwz_hip.so:SCHWZ_WALK_CACHED=0
        [ "$LE" != "$L" ] && export ${LE#*:}
        line=$(SCHWZ_HIP_LIB=$ROOT/schwarz-lib_amd/lib/$L python3 $ROOT/bench.py --steps 20 --warmup 3 --no-ttr --no-cpu-baseline --no-plain-loop --no-mirror --no-shapes --strong-grid "" $ARGS 2>/dev/null) || exit 1
        [ "$LE" != "$L" ] && { v=${LE#*:}; unset ${v%%=*}; }
        echo "$shape $LE $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms/step %.4f  update %.4f ms frac %.3f  dirdot %.4f ms frac %.3f  reduction %.17g" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline_spmv"]["avg_launch_ms"], d["roofline_spmv"]["frac"], d["residual_reduction_in_timed_steps"]))')" | tee -a $OUT
    done
done
