#!/bin/bash
# segment length of the update launch's table (SCHWZ_SWEEP_L) with the fused launch on its own table
for rep in 1 2; do for l in auto 16 24 32 48 64; do
  if [ $l = auto ]; then e="SCHWZ_X=0"; else e="SCHWZ_SWEEP_L=$l"; fi
  echo "L=$l: $(env $e python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ttr --no-plain-loop $BENCH_ARGS 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4f ms/step  upd %.4f ms  dirdot %.4f ms" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline_spmv"]["avg_launch_ms"]))')"
done; done
