#!/bin/bash
# one-off PMC pass: tools/pmc_extra.sh "<counters>" tag
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof/pmc_$2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-also --no-host-rate $EXTRA > $OUT.log 2>&1 || { tail -5 $OUT.log; exit 0; }
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc, n = {}, {}
for r in csv.DictReader(open(f)):
    if "jb_step_kernel" not in r["Kernel_Name"]: continue
    k = r["Counter_Name"]; acc[k] = acc.get(k, 0) + float(r["Counter_Value"]); n[k] = n.get(k, 0) + 1
for k in acc: print("%-28s %.4g" % (k, acc[k] / n[k]))
PY
