#!/bin/bash
# tools/ab_seeds.sh "seed list" lib1.so lib2.so ...  : mean launch time per build over several seeds (builds whose numerics
# differ put different robots on the floor, which moves the slowest-wave time by a few % for any single seed)
SEEDS=$1; shift
for lib in "$@"; do
  tot=0; k=0; line=""
  for sd in $SEEDS; do
    JITTERBUG_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps ${STEPS:-400} --warmup 50 --no-cpu-baseline --no-also --no-host-rate --no-steady --seed $sd > gpurun_out/abs.json || exit 1
    ms=$(python -c "
import json, sys
d = json.loads(open('gpurun_out/abs.json').read().strip().split('\n')[-1])
if not d['finite'] or d['solver_cap_hits'] > 2000: sys.exit('NON-FINITE or solver-cap storm: finite=%s cap=%s (a timing of a broken build is meaningless)' % (d['finite'], d['solver_cap_hits']))
print(d['roofline']['launch_ms'])") || exit 1
    line="$line $ms"; tot=$(python -c "print($tot + $ms)"); k=$((k+1))
  done
  python -c "print('%-28s mean launch %.4f ms -> %.3f M env-steps/s   [%s ]' % ('$lib', $tot/$k, 4096/($tot/$k)/1e3, '$line'))"
done
