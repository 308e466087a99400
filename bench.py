#!/usr/bin/env python
"""bench.py — env steps/s of the MI355X-native Jitterbug stepper (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one control step (50 physics substeps + reward + observation, in-kernel auto-reset) of every
environment of the batch.  Workload at N=1: BASELINE.json configs[2] — move_from_origin, 4096 envs, full
contact solve (configs[1], contacts off, is reported alongside in "also").  For N>1 (launched by
torch.distributed.run, one rank per GPU) each rank steps its own 4096 envs (weak scaling; env indices are
global so results do not depend on the split) and rank 0 gathers [N_local, D+2] obs/reward/done rows over
RCCL every step, inside the timed region.

Inputs (actions) are resident in HBM before the timed region; the timed region is K launches bracketed by
barrier + torch.cuda.synchronize(); rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 317       # read qpos 64 + qvel 60 + ctrl 4; write qpos 64 + qvel 60 + obs 60 + reward 4 + done 1
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8.0 TB/s spec
N_ENVS_PER_GPU = 4096
TASK = "move_from_origin"


def usable_cores():
    """Host threads this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU box hands out a share)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return n


def cpu_baseline(n_cores, task, budget_s=12.0):
    """The CPU fp64 oracle (a port/restatement, NOT MuJoCo) on the host cores, on a bounded sample of the same workload."""
    import numpy as np
    from jitterbug_amd import model
    from oracle import oracle as O
    P = model.default_params()
    TASK = task
    # one core first (SURVEY.md §8d asks for the 1-core rate next to the all-core one): 16 envs x a few control steps
    e1 = O.OracleEnv(16, TASK, P, seed=0)
    e1.reset()
    r1 = np.random.default_rng(1)
    e1.step(r1.uniform(-1, 1, size=16), nthreads=1)
    t0 = time.perf_counter()
    k1 = 0
    while time.perf_counter() - t0 < 2.0:
        e1.step(r1.uniform(-1, 1, size=16), nthreads=1)
        k1 += 1
    one_core = 16 * k1 / (time.perf_counter() - t0)
    n = 64 * n_cores
    env = O.OracleEnv(n, TASK, P, seed=0)
    env.reset()
    rng = np.random.default_rng(0)
    t0 = time.perf_counter()
    env.step(rng.uniform(-1, 1, size=n), nthreads=n_cores)
    dt = time.perf_counter() - t0
    steps = int(max(3, min(200, budget_s / max(dt, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(steps):
        env.step(rng.uniform(-1, 1, size=n), nthreads=n_cores)
    dt = time.perf_counter() - t0
    return {"value": n * steps / dt, "unit": "env steps/s", "cores": n_cores, "kind": "port", "one_core_value": one_core,
            "sample": "%d envs x %d control steps of %s, fp64 oracle (oracle/jb_oracle.c: Newton contact solve), OpenMP over envs" % (n, steps, TASK)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=N_ENVS_PER_GPU)
    ap.add_argument("--contacts", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true")
    ap.add_argument("--max-newton", type=int, default=12)
    ap.add_argument("--envs-per-wave", type=int, default=0)
    ap.add_argument("--no-rank-one", action="store_true", help="diagnostic: JB_FLAG_NO_RANK_ONE (every Newton pass is a full pass)")
    ap.add_argument("--actions", default="uniform", help="uniform (default, the metric's workload) | const1 (motor flat out: about half the robots tip over - diagnostic)")
    ap.add_argument("--seed", type=int, default=0, help="reset / action stream seed (the committed numbers use 0)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the measured path); gloo is a CPU-staged rehearsal of the N>1 control flow")
    ap.add_argument("--task", default=TASK, help="default move_from_origin (the BASELINE metric); move_to_pose is BASELINE configs[3]'s task")
    ap.add_argument("--augmented", action="store_true", help="one randomised model per env (BASELINE configs[4], augment_Jitterbug semantics)")
    args = ap.parse_args()
    task = args.task

    import numpy as np
    import torch
    from jitterbug_amd import model
    from jitterbug_amd.vec_env import JitterbugVecEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks (WORLD_SIZE=%d)" % (args.gpus, args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # JB_BENCH_DEVICE: rehearsal of the N>1 control flow on a box with one GPU (every rank on that device, gloo backend)
    dev_index = int(os.environ.get("JB_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # JB_BENCH_FORCE_DIST=1 (under torch.distributed.run with one rank): take the N>1 code path - process group, per-step
    # gather, barrier, max-reduction - with a world of one, to exercise the real RCCL calls on a one-GPU box
    force_dist = world == 1 and os.environ.get("JB_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)          # RCCL over xGMI
        else:
            dist.init_process_group(args.dist_backend)
    local_rank = dev_index

    n = args.envs_per_gpu
    D = model.OBS_DIM[task]
    K, W = args.steps, args.warmup

    def make_env(contacts):
        # the handle launches on torch's current stream so torch.cuda.Event brackets exactly these kernels
        env = JitterbugVecEnv(n, task, seed=args.seed, device_id=local_rank, contacts=bool(contacts), env_offset=rank * n, max_newton=args.max_newton, envs_per_wave=args.envs_per_wave, flags=1 if args.no_rank_one else 0,
                              stream=torch.cuda.current_stream(dev).cuda_stream)
        if args.augmented:           # BASELINE configs[4]: one randomised model per env, generated on the device (keyed by the global env index)
            env.randomise_models(seed=1000, min_mass_clearance=1e-3, return_params=False)
        return env

    def run(contacts, steps, warmup, gather):
        env = make_env(contacts)
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + rank + 7919 * args.seed)
        actions = torch.rand((steps + warmup, n), generator=g, device=dev, dtype=torch.float32) * 2 - 1
        if args.actions == "const1":
            actions.fill_(1.0)
        obs = torch.empty((n, D), device=dev, dtype=torch.float32)
        rew = torch.empty((n,), device=dev, dtype=torch.float32)
        done = torch.empty((n,), device=dev, dtype=torch.uint8)
        # N > 1: the step kernel writes packed rows [obs | reward | done] itself (jb_step_rows_device) and rank 0 gathers them
        # every step.  Two row buffers alternate so that the gather of step t (on RCCL's stream) overlaps the kernel of step
        # t+1; a buffer is rewritten only after the gather that read it has been waited for (three buffers with RCCL, so that the
        # step kernel never waits for the gather issued just before it).
        nccl = gather and args.dist_backend == "nccl"
        NB = 3 if nccl else 2                   # row buffers in flight
        rows = [torch.empty((n, D + 2), device=dev, dtype=torch.float32) for _ in range(NB)] if gather else None
        stage = [torch.empty((n, D + 2), dtype=torch.float32).pin_memory() for _ in range(2)] if (gather and not nccl) else None
        gathered = None
        if gather and rank == 0:
            gathered = [[torch.empty((n, D + 2), device=dev if nccl else "cpu", dtype=torch.float32) for _ in range(world)] for _ in range(NB)]
        pending = [None] * NB
        env.reset_device(None, obs.data_ptr())

        # RCCL's gather kernel cannot share a SIMD with a step-kernel wave (one wave per SIMD, ~460 registers), so a gather issued
        # right behind step t delays the start of step t+1 by its own duration.  It is therefore issued one step LATE, from a
        # side stream that waits only for step t's event: by then step t+1 occupies the device and the gather runs in its tail,
        # on the SIMDs whose waves have already finished.
        side = torch.cuda.Stream(device=dev) if nccl else None
        step_done = [torch.cuda.Event() for _ in range(NB)] if nccl else None
        late = [None]                               # index of the step whose rows still have to be sent

        def send(j):
            b = j % NB
            with torch.cuda.stream(side):
                side.wait_event(step_done[b])
                pending[b] = dist.gather(rows[b], gathered[b] if rank == 0 else None, dst=0, async_op=True)

        def one(i):
            if not gather:
                env.step_device(actions[i].data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
                return
            b = i % NB
            if pending[b] is not None:
                pending[b].wait()                   # (current stream waits: the buffer is free to be rewritten)
                pending[b] = None
            env.step_rows_device(actions[i].data_ptr(), rows[b].data_ptr())
            if nccl:
                step_done[b].record()
                if late[0] is not None:
                    send(late[0])
                late[0] = i
            else:                                   # rehearsal: stage through the host
                stage[b].copy_(rows[b])
                dist.gather(stage[b], gathered[b] if rank == 0 else None, dst=0)

        def drain():
            if nccl and late[0] is not None:
                send(late[0])
                late[0] = None
            for b in range(NB):
                if pending[b] is not None:
                    pending[b].wait()
                    pending[b] = None

        for i in range(warmup):
            one(i)
        drain()
        finite_warm = bool(torch.isfinite(rows[(warmup - 1) % NB] if (gather and warmup) else obs).all().item()) if warmup else True
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for i in range(warmup, warmup + steps):
            one(i)
        ev1.record()
        drain()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        dev_ms = ev0.elapsed_time(ev1)
        sc, ep, cap = env.counters()
        last = rows[(warmup + steps - 1) % NB][:, :D] if gather else obs
        finite = finite_warm and bool(torch.isfinite(last).all().item())      # checked after the warm-up and after the timed steps
        finite = finite and float(cap.max()) < 1000.0                         # ... and no env ever ended a step non-finite (kernel-side flag)
        if gather and rank == 0:                    # the gathered block of the last step really holds every rank's rows
            gl = gathered[(warmup + steps - 1) % NB]
            finite = finite and all(bool(torch.isfinite(x).all().item()) for x in gl) and bool((gl[0].to(dev) == rows[(warmup + steps - 1) % NB]).all().item())
        env.close()
        return wall, dev_ms, float(cap.sum()), finite

    wall, dev_ms, cap_hits, finite = run(args.contacts, K, W, gather=(dist is not None))
    t = torch.tensor([wall], device=dev if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max = float(t.item())
    total_envs = n * world
    value = total_envs * K / wall_max

    also = None
    if not args.no_also and world == 1:
        w2, d2, _, _ = run(1 - args.contacts, max(50, K // 5), 20, gather=False)
        also = {"workload": "%s N_envs=%d contacts %s (BASELINE configs[%d])" % (task, n, "off" if args.contacts else "on", 1 if args.contacts else 2),
                "value": n * max(50, K // 5) / w2, "unit": "env steps/s"}

    if rank == 0:
        # HBM traffic per launch from the rocprofv3 PMC passes of the same workload (FETCH_SIZE + WRITE_SIZE, separate runs;
        # profiles/r01_pmc_summary.md).  It cannot be collected inside this process, so it is read from the committed summary.
        traffic = None
        issue = None
        try:
            raw = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_raw.json")))
            if n == N_ENVS_PER_GPU and args.contacts and task == TASK and not args.augmented:
                traffic = (raw["FETCH_SIZE_KB"] + raw["WRITE_SIZE_KB"]) * 1024.0
                sq = raw["sq"]
                issue = {"valu_issue_frac_of_wave_life": sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"],
                         "any_issue_frac_of_wave_life": sq["SQ_ACTIVE_INST_ANY"] / sq["SQ_WAVE_CYCLES"],
                         "waitcnt_frac_of_wave_life": sq.get("SQ_WAIT_ANY", 0.0) / sq["SQ_WAVE_CYCLES"],
                         "source": "profiles/r01_pmc_raw.json (rocprofv3 PMC passes of this workload)"}
        except Exception:
            pass
        launch_s = dev_ms * 1e-3 / K                 # average duration of one jb_step_kernel launch, from HIP events on its stream
        # 317 B for move_from_origin (D=15); other tasks add 4 B per extra obs entry and the 12 B target read; per-env models add
        # the lane constant table (202 x 4 floats) read once per step
        algo_bytes = ALGO_BYTES_PER_ENV_STEP + 4 * (D - 15) + (12 if task != TASK else 0) + (202 * 4 * 4 if args.augmented else 0)
        achieved = algo_bytes * n / launch_s / 1e9
        res = {
            "metric": "env steps/s at N_envs=%d, %s" % (n, task),
            "value": value, "unit": "env steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall_max * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, N_envs=%d per GPU, %s, 50 substeps/step, in-kernel auto-reset (BASELINE configs[%d])"
                                   % (task + (", one randomised model per env" if args.augmented else ""), n, "full Newton contact solve" if args.contacts else "contacts off", 2 if args.contacts else 1),
                       "global_envs": total_envs, "parallelism": "env-sharded x%d%s" % (world, (", %s gather of [N,D+2] rows to rank 0 every step, double-buffered" % ("RCCL" if args.dist_backend == "nccl" else args.dist_backend + " (rehearsal)")) if world > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "kernel": "jb_step_kernel", "launch_ms": launch_s * 1e3, "algorithmic_bytes_per_launch": algo_bytes * n,
                         "note": "path is fp32-VALU/latency bound, not HBM bound (SURVEY.md §8d): 317 B per env step vs ~1e5-1e6 dependent flops",
                         "issue": issue},
            "solver_cap_hits": cap_hits, "finite": finite,
        }
        if also:
            res["also"] = also
        if world == 1 and not args.no_cpu_baseline:
            cores = usable_cores()
            res["cpu_baseline"] = cpu_baseline(cores, task)
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
