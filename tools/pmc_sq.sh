#!/bin/bash
# SQ / TCC counter passes (no TA/TCP sets: they hung the profiler on this pool) over the SpMV
# launches of tools/spmv_probe.py.   tools/pmc_sq.sh <tag> <variants>
set -o pipefail
TAG=${1:-pmcsq}
VARS=${2:-0}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/set$i -- python3 $ROOT/tools/spmv_probe.py --only-spmv --variants $VARS --reps 5 > $OUT/set$i.log 2>&1 || { echo "set $i failed"; tail -5 $OUT/set$i.log; break; }
  echo "set $i done"
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/set*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        if "spmv" not in k: continue
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        d = dur[k]; d[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); d[1] += 1
for k in sorted(acc):
    print("==", k, "avg dur us %.1f" % (dur[k][0] / dur[k][1] / 1e3))
    for c in sorted(acc[k]):
        s, n = acc[k][c]
        print("   %-36s %16.1f" % (c, s / n))
PY
