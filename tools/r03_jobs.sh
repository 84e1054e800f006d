#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for ord in 1 0; do
  export SCHWZ_STREAM_ORDER=$ord
  rm -rf /tmp/fp_$ord
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fp_$ord -- python3 $ROOT/tools/plain_spmv_run.py 1024,1024,32 6 > /tmp/fp_$ord.log 2>&1
  python3 - <<PY
import csv, glob
tot=[0,0]
for f in glob.glob('/tmp/fp_$ord/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'spmv_stream_kernel' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE':
            tot[0]+=float(r['Counter_Value']); tot[1]+=1
print('order $ord: FETCH_SIZE x2 per launch = %.3f GB over %d launches' % (2*tot[0]/max(tot[1],1)*1024/1e9, tot[1]))
PY
done
