"""Diagnostic (needs a -DJB_WAVE_STATS build): where the hardware puts the workgroups of a step launch - which workgroup indices share a
SIMD when two waves per SIMD are resident (LEAN kernels) - and how long each wave lived next to its SIMD-mate.

    JITTERBUG_HIP_LIB=ab_build/libjb_ws.so python tools/debug/placement.py [n_envs] [flags] [augmented]
"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jitterbug_amd.vec_env import JitterbugVecEnv
from jitterbug_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 2
aug = len(sys.argv) > 3 and sys.argv[3] == "augmented"
dev = torch.device("cuda", 0)
env = JitterbugVecEnv(n, "move_to_pose", seed=0, flags=flags | 32, stream=torch.cuda.current_stream(dev).cuda_stream)      # index order: workgroup b = wave b
if aug:
    env.randomise_models(seed=1000, return_params=False)
env.reset_device()
g = torch.Generator(device=dev); g.manual_seed(1234)
tape = torch.rand((200, n), generator=g, device=dev) * 2 - 1
obs = torch.zeros((n, env.obs_dim), device=dev); rew = torch.zeros((n,), device=dev); done = torch.zeros((n,), device=dev, dtype=torch.uint8)
L = _lib.load()
L.jb_debug_wave_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
for k in range(120):
    env.step_device(tape[k].data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
torch.cuda.synchronize(dev)
wc = env.wave_clocks() * 1e3
big = np.zeros((n * 5, 16), dtype=np.uint64)
e = L.jb_debug_wave_stats(env._h, big.ctypes.data, n * 5)
nw = (n + e - 1) // e
w = big[n:].reshape(-1, 64)[:nw, 63]
blk = (w >> np.uint64(40)).astype(np.int64); xcc = ((w >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64); hw = (w & np.uint64(0xffffffff)).astype(np.int64)
wave_slot = hw & 0xf; simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
print("kernel variant %s, %d waves; XCCs %s, SEs %s, SHs %s, CUs %s, SIMDs %s, wave slots %s" % (env.kernel_variant, nw, sorted(set(xcc)), sorted(set(se)), sorted(set(sh)), sorted(set(cu)), sorted(set(simd)), sorted(set(wave_slot))))
print("first 40 workgroups: (xcc, se, sh, cu, simd, slot)")
for b in range(40):
    i = int(np.where(blk == b)[0][0])
    print("  wg %4d -> xcc %d se %d sh %d cu %2d simd %d slot %d   life %.3f ms" % (b, xcc[i], se[i], sh[i], cu[i], simd[i], wave_slot[i], wc[i]))
key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd)
from collections import defaultdict
d = defaultdict(list)
for i in range(nw):
    d[int(key[i])].append(int(blk[i]))
cnt = np.bincount([len(v) for v in d.values()])
print("SIMDs used %d; waves per SIMD histogram %s" % (len(d), cnt.tolist()))
diffs = [abs(v[0] - v[1]) for v in d.values() if len(v) == 2]
if diffs:
    vals, c = np.unique(diffs, return_counts=True)
    o = np.argsort(-c)[:8]
    print("index distance between the two workgroups of a SIMD (last launch): " + ", ".join("%d x%d" % (vals[j], c[j]) for j in o))
    pair_sum = np.array([wc[np.where(blk == v[0])[0][0]] + wc[np.where(blk == v[1])[0][0]] for v in d.values() if len(v) == 2])
    print("per-SIMD sum of the two wave lives: mean %.3f max %.3f ms; launch ~ max wave life %.3f" % (pair_sum.mean(), pair_sum.max(), wc.max()))
env.close()
