#!/bin/bash
# Register / scratch / occupancy report of every kernel in jb_api.hip (compiler remarks; no GPU needed).
#   tools/kernel_resources.sh [extra hipcc flags]
cd "$(dirname "$0")/.." || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -disable-vector-combine -Wno-unused-value -c --cuda-device-only \
    -Rpass-analysis=kernel-resource-usage "$@" -o /tmp/jb_res.o jitterbug_amd/csrc/jb_api.hip 2>&1 | python3 -c '
import re, sys
name, rows = None, {}
for line in sys.stdin:
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", m.group(1)); rows[name] = []
        continue
    m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill): (\S+)", line)
    if m and name:
        rows[name].append("%s %s" % (m.group(1).split(" ")[0], m.group(2)))
for k in sorted(rows):
    print("%-60s %s" % (k[:60], " | ".join(rows[k])))
'
