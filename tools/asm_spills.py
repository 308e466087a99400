"""Where do a step kernel's register spills sit - in the substep loop (hot) or only around it (once per control step / per launch)?

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -disable-vector-combine -w -S --cuda-device-only -o /tmp/jb.s jitterbug_amd/csrc/jb_api.hip
    python tools/asm_spills.py /tmp/jb.s [kernel-name-substring]

For every step kernel: static instruction count, the scratch_ / v_writelane / v_readlane operations and the packed fp32 instructions
(v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) split by the loop they are in.
Loops are found from backward branches (label .LBBn_m defined above its s_cbranch / s_branch); the SUBSTEP loop is taken to be the
largest loop nested inside the outermost one (the control-step loop)."""
import re
import sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "jb_step_kernel"
lines = open(path).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\S+:", l)]
ends = [i for i, l in enumerate(lines) if l.startswith(".Lfunc_end")]
is_inst = re.compile(r"^\s+(v_|s_|ds_|global_|scratch_|buffer_|flat_)")
for (i0, name), i1 in zip(starts, ends):
    if want not in name:
        continue
    body = lines[i0:i1]
    inst_idx = [j for j, l in enumerate(body) if is_inst.match(l)]
    labels = {}
    for j, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = j
    loops = []
    for j, l in enumerate(body):
        m = re.match(r"^\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < j:
            loops.append((labels[m.group(1)], j))
    loops.sort(key=lambda ab: ab[0] - ab[1])          # largest first
    outer = loops[0] if loops else (0, len(body))
    inner = [ab for ab in loops[1:] if ab[0] >= outer[0] and ab[1] <= outer[1]]
    sub = inner[0] if inner else outer

    # the cold block of the substep loop - the line-searched second contact solve (jb_sim.hpp substep_impl) - is bracketed by comments in the ISA
    cb = [j for j, l in enumerate(body) if "jb-cold-solve-begin" in l]
    ce = [j for j, l in enumerate(body) if "jb-cold-solve-end" in l]
    cold = (cb[0], ce[-1]) if cb and ce else None

    def count(pred, lo, hi, skip=None):
        return sum(1 for j in inst_idx if lo <= j <= hi and pred(body[j]) and not (skip and skip[0] <= j <= skip[1]))

    def report(lo, hi, skip=None):
        return "insts %6d  scratch %4d  writelane %4d  readlane %4d  accvgpr %4d  v_pk_* %4d" % (
            count(lambda l: True, lo, hi, skip), count(lambda l: "scratch_" in l, lo, hi, skip), count(lambda l: "v_writelane" in l, lo, hi, skip),
            count(lambda l: "v_readlane" in l, lo, hi, skip), count(lambda l: "v_accvgpr" in l, lo, hi, skip), count(lambda l: re.match(r"^\s+v_pk_(fma|mul|add)_f32", l) is not None, lo, hi, skip))

    short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:44]
    print("%-44s whole        %s" % (short, report(0, len(body))))
    print("%-44s step loop    %s" % ("", report(*outer)))
    print("%-44s substep loop %s" % ("", report(*sub)))
    if cold:
        print("%-44s  - hot part  %s   (the loop without its cold block: the line-searched second solve)" % ("", report(sub[0], sub[1], cold)))
        print("%-44s  - cold block %s" % ("", report(*cold)))
