#!/bin/bash
# Diagnostic (GPU box): kernel trace of the N>1 code path with a world of one (RCCL gather every step) next to the plain path.
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_dist; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="--steps 300 --warmup 100 --no-cpu-baseline --no-also --no-host-rate --no-steady"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/plain -- python3 $ROOT/bench.py $A > $OUT/plain.log 2>&1
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 JB_BENCH_FORCE_DIST=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dist -- python3 $ROOT/bench.py --gpus 1 $A > $OUT/dist.log 2>&1
cd $ROOT
for d in plain dist; do echo "== $d"; grep '^{' $OUT/$d.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])"; f=$(find $OUT/$d -name '*kernel_stats.csv' | head -1); head -6 $f | cut -c1-200; done
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_dist/dist/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'jb_step_kernel' in r['Kernel_Name']]
i0, i1 = idx[250], idx[254]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1 + 1]:
    print("%-40s q%s s%s start %8.1f us dur %7.1f us" % (r['Kernel_Name'][:40], r.get('Queue_Id'), r.get('Stream_Id'), (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
PY
