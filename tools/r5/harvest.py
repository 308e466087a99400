"""Diagnostic (a -DJB_CAPTURE build named by JITTERBUG_HIP_LIB): run fused episodes and dump the entry states of the substeps whose contact
solve stayed unconverged even after the line-searched pass.   python tools/r5/harvest.py out.npy [episodes]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jitterbug_amd.vec_env import JitterbugVecEnv
from jitterbug_amd import _lib
out = sys.argv[1]; E = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
recs = []
for label, n, kw, rnd, const in (("uniform", 4096, {}, False, False), ("flat", 4096, {}, False, True)):
    env = JitterbugVecEnv(n, "move_from_origin", seed=1, **kw)
    g = torch.Generator(device=dev); g.manual_seed(5)
    tape = torch.rand((1000, n), generator=g, device=dev) * 2 - 1
    if const: tape.fill_(1.0)
    env.reset_device()
    for e in range(E):
        env.step_many_device(1000, tape.data_ptr())
    env.synchronize()
    L = _lib.load()
    buf = np.zeros((256, 64), dtype=np.float32)
    L.jb_debug_captured.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    cnt = L.jb_debug_captured(env._h, buf.ctypes.data, 256)
    sc, ep, cap = env.counters()
    print(label, "unconverged", float(cap.sum()), "captured", cnt, "resolved", env.solver_stats())
    recs.append(buf[:min(cnt, 256)])
    env.close()
np.save(out, np.concatenate(recs))
