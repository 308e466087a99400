cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
ROOT=$(pwd)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --list-avail > $ROOT/gpurun_out/r5/avail.txt 2>&1 )
grep -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_WAIT_INST[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*\|SQ_BUSY[A-Z_0-9]*" gpurun_out/r5/avail.txt | sort -u | tr '\n' ' ' > gpurun_out/r5/avail_sq.txt
cat gpurun_out/r5/avail_sq.txt; echo
for lib in "$@"; do
  OUT=$ROOT/gpurun_out/prof/ic_$(basename $lib .so); rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp && JITTERBUG_HIP_LIB=$ROOT/$lib timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 300 --warmup 100 --no-cpu-baseline --no-also --no-host-rate --no-steady > $OUT.log 2>&1 )
  python3 - "$OUT" "$lib" <<'PY'
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not fs: print(sys.argv[2], "no counters"); sys.exit(0)
acc, n = {}, {}
for r in csv.DictReader(open(fs[0])):
    if "jb_step_kernel" not in r["Kernel_Name"]: continue
    k = r["Counter_Name"]; acc[k] = acc.get(k, 0) + float(r["Counter_Value"]); n[k] = n.get(k, 0) + 1
v = {k: acc[k] / n[k] for k in acc}
print("%-26s" % sys.argv[2], " ".join("%s %.3f M" % (k, v[k] / 1e6) for k in sorted(v)))
PY
done
