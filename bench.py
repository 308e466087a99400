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

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts ITSELF: the parent spawns
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process (the counterpart of the reference's
SubprocVecEnv spawning its workers from one command, benchmarks/benchmark.py:146-171), never imports torch or touches a
GPU, relays rank 0's JSON line and exits with the child's return code.
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
    n = N_ENVS_PER_GPU           # the metric's batch (4096 envs), over as many control steps of the episode as the time budget allows
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
            "sample": "%d envs (the metric's batch) x %d control steps from the reset (of the episode's 1000) of %s, fp64 oracle (oracle/jb_oracle.c: Newton contact solve), OpenMP over envs" % (n, steps, TASK)}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n_gpus, argv):
    """Parent of a `--gpus N` run started without torch.distributed.run: spawn the ranks as a CHILD process tree (no exec, no
    torch import, no HIP call here), pass their stderr through, print rank 0's single JSON line, return the child's rc.
    A watchdog (JB_BENCH_LAUNCH_TIMEOUT seconds, default 1500) kills the child's whole process group if the ranks neither finish nor
    fail - a rendezvous or a collective that never completes must not hang the caller; the run then returns 124 and prints no line."""
    import signal
    import subprocess
    import threading
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    # What the launcher sets for the ranks, and why:
    #   HSA_ENABLE_IPC_MODE_LEGACY=0   dmabuf IPC: the host driver of this pool supports only that, RCCL across processes fails with
    #                                  hipIpcGetMemHandle: invalid argument otherwise (already exported on the boxes; kept for a bare environment)
    #   (MASTER_ADDR/PORT, RANK, LOCAL_RANK, WORLD_SIZE come from torch.distributed.run; nothing else is touched)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, start_new_session=True)
    limit = float(os.environ.get("JB_BENCH_LAUNCH_TIMEOUT", "1500"))
    timed_out = []

    def kill_group():
        timed_out.append(True)
        try:
            os.killpg(child.pid, signal.SIGKILL)          # the exact process group this function started (start_new_session), nothing by pattern
        except (ProcessLookupError, PermissionError, AttributeError):
            pass
    watchdog = threading.Timer(limit, kill_group)
    watchdog.daemon = True
    if hasattr(child, "pid"):
        watchdog.start()
    line = None
    for out in child.stdout:
        if out.startswith("{") and '"metric"' in out:
            line = out.strip()                                # rank 0's result (only rank 0 prints one)
        else:
            sys.stderr.write(out)                             # anything else the ranks print is not the result
    rc = child.wait()
    watchdog.cancel()
    if timed_out:
        sys.stderr.write("bench.py: the ranks did not finish within %.0f s (JB_BENCH_LAUNCH_TIMEOUT): killed\n" % limit)
        return 124
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks exited cleanly but printed no result line\n")
        rc = 1
    if line is not None and '"degraded": true' in line:
        # a DEGRADED line (no data-path collective ran: `value` is null; the ranks exit 3, which torch.distributed.run reports as 1): relayed, rc 3
        print(line, flush=True)
        return 3
    if rc == 0:
        print(line, flush=True)
    return rc


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=N_ENVS_PER_GPU)
    ap.add_argument("--contacts", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true")
    ap.add_argument("--no-host-rate", action="store_true", help="skip the host-buffer (jb_step) rate measured next to the headline")
    ap.add_argument("--max-newton", type=int, default=0, help="checks of the active set before a substep goes to the line-searched solve (0: the library's default, 12)")
    ap.add_argument("--envs-per-wave", type=int, default=0)
    ap.add_argument("--no-rank-one", action="store_true", help="diagnostic: JB_FLAG_NO_RANK_ONE (every Newton pass is a full pass)")
    ap.add_argument("--no-reorder", action="store_true", help="diagnostic: JB_FLAG_NO_REORDER (never launch the waves longest-first)")
    ap.add_argument("--no-spread", action="store_true", help="diagnostic: JB_FLAG_NO_SPREAD (contact sweeps never in spread mode)")
    ap.add_argument("--no-pair", action="store_true", help="diagnostic: JB_FLAG_NO_PAIR (floor contacts only, also for per-env models: what rounds 1-2 simulated)")
    ap.add_argument("--lean", action="store_true", help="force the two-waves-per-SIMD kernel variant (JB_FLAG_LEAN); by default the product picks it from the per-GPU batch (jitterbug_amd.variants)")
    ap.add_argument("--no-lean", action="store_true", help="never the LEAN variant")
    ap.add_argument("--actions", default="uniform", help="uniform (default, the metric's workload) | const1 (motor flat out: about half the robots tip over - diagnostic)")
    ap.add_argument("--seed", type=int, default=0, help="reset / action stream seed (the committed numbers use 0)")
    ap.add_argument("--dist-backend", default="nccl", help="--collective torch only: nccl (= RCCL through PyTorch); gloo is a CPU-staged rehearsal of the N>1 control flow")
    ap.add_argument("--collective", default="torch", choices=["cabi", "torch"],
                    help="N > 1: who moves the rows.  torch (default): torch.distributed's gather over RCCL - the path PyTorch has run on many ranks; cabi: the library's own RCCL binding "
                         "(jb_comm_init / jb_scatter_actions_device / jb_gather_rows_device; torch.distributed then carries only control traffic over gloo) - it has never seen two GPUs, so it is "
                         "opt-in until a multi-GPU box has run it.  If the chosen path fails the other is tried, then the shards are timed without any collective and the line says degraded")
    ap.add_argument("--actions-from", default="local", choices=["local", "rank0"],
                    help="N > 1: where a step's actions come from.  local (default): resident on every rank before the timed region; rank0: rank 0 holds the actions of the whole batch and "
                         "every step scatters them (jb_scatter_actions_device / torch.distributed.scatter): the timed region is then scatter -> step -> gather, the reference's whole per-step "
                         "round trip (benchmarks/benchmark.py:161-171); the line says which it timed")
    ap.add_argument("--task", default=TASK, help="default move_from_origin (the BASELINE metric); move_to_pose is BASELINE configs[3]'s task")
    ap.add_argument("--augmented", action="store_true", help="one randomised model per env (BASELINE configs[4], augment_Jitterbug semantics)")
    ap.add_argument("--no-steady", action="store_true", help="skip the steady-state (steps 100-400) and full-episode blocks measured next to the headline")
    args = ap.parse_args(argv)
    # which step kernel: resolved by the PRODUCT (jitterbug_amd.variants - this script holds no threshold of its own); --lean / --no-lean force one
    variant = "lean" if args.lean else "ordinary" if args.no_lean else "auto"
    task = args.task
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus, argv)          # before anything imports torch or initialises HIP in this process

    import numpy as np
    import torch
    from jitterbug_amd import model, variants
    from jitterbug_amd.vec_env import JitterbugVecEnv
    args.lean = variants.resolve(variant, args.envs_per_gpu, args.augmented) == "lean"
    used_variant = []          # what jb_kernel_variant reports for the handles of the timed runs

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
    ctl = None                 # control group (gloo, CPU tensors): barriers, error agreement and the max over ranks never depend on RCCL being healthy
    dist_notes = []
    nccl_group = [None]        # --collective torch when the default group is gloo: an RCCL subgroup, built on first use
    if world > 1 or force_dist:
        import torch.distributed as dist
        if args.collective == "cabi" or args.dist_backend != "nccl":
            # control traffic (barriers, the max over ranks, failure agreement, the 128-byte communicator id) over gloo; the rows travel through
            # the library's own RCCL communicator (cabi) or, --dist-backend gloo, staged through the host (rehearsal)
            dist.init_process_group("gloo")
        else:
            try:
                dist.init_process_group("nccl", device_id=dev)          # RCCL over xGMI (device_id: the communicator is built here, errors surface here)
                ctl = dist.new_group(backend="gloo")
            except Exception as e:
                dist_notes.append("RCCL process group through PyTorch failed (%s: %s)" % (type(e).__name__, str(e)[:200]))
                try:
                    dist.destroy_process_group()
                except Exception:
                    pass
                dist.init_process_group("gloo")
                args.dist_backend = "gloo-control-only"
    local_rank = dev_index
    # the data paths to try, in order: what was asked for, then the other RCCL path; after that the shards are timed alone and the line is DEGRADED
    gather_modes = []
    if dist is not None:
        if args.dist_backend == "gloo":
            gather_modes = ["torch"]                     # rehearsal: torch.distributed over gloo
        elif args.dist_backend == "gloo-control-only":
            gather_modes = ["cabi"]                      # PyTorch's RCCL did not come up: the library's own may
        else:
            gather_modes = ["cabi", "torch"] if args.collective == "cabi" else ["torch", "cabi"]
    use_gather = gather_modes[0] if gather_modes else None

    class RanksDisagree(RuntimeError):
        pass

    def rank_max(x):
        t = torch.tensor([x], dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=ctl)
        return float(t.item())

    def sync_point(failure=None):
        """barrier + agreement: every rank waits here (device idle first); raises on EVERY rank if any rank brought a failure"""
        try:
            torch.cuda.synchronize(dev)
        except Exception as e:
            failure = failure or e
        if dist is None:
            if failure is not None:
                raise failure
            return
        bad = rank_max(0.0 if failure is None else 1.0)
        if bad:
            raise RanksDisagree("a rank failed: %s" % (("%s: %s" % (type(failure).__name__, str(failure)[:300])) if failure is not None else "another rank"))

    n = args.envs_per_gpu
    D = model.OBS_DIM[task]
    K, W = args.steps, args.warmup

    def run(contacts, steps, warmup, gather):
        """One timed run.  Every rank passes the same three sync points whatever happens to it, so ranks whose gather path raises (its
        first run over real links, say) leave this run together - RanksDisagree - and nobody is left in a barrier.  (A rank that fails
        alone INSIDE a collective still leaves the others waiting in it: that ends at the process group's timeout / the launcher's
        watchdog, without a line.)"""
        failure = None
        try:
            setup = run_setup(contacts, steps, warmup, gather)
        except Exception as e:
            failure, setup = e, None
        try:
            sync_point(failure)
        except Exception:
            close_quietly(setup)
            raise
        env, sh, one, drain, obs, last = setup
        t0 = time.perf_counter()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        try:
            ev0.record()
            for i in range(warmup, warmup + steps):
                one(i)
            ev1.record()
            drain()
            torch.cuda.synchronize(dev)
        except Exception as e:
            failure = e
        wall = time.perf_counter() - t0
        try:
            sync_point(failure)
        except Exception:
            close_quietly(setup)          # a failed attempt leaves no communicator, stream or handle behind before the next mode is tried
            raise
        try:
            out = run_finish(env, sh, obs, last, wall, ev0.elapsed_time(ev1))
        except Exception as e:
            failure, out = e, None
            close_quietly(setup)
        sync_point(failure)
        return out

    def close_quietly(setup):
        if not setup:
            return
        env, sh = setup[0], setup[1]
        for obj in (sh, env):
            try:
                if obj is not None:
                    obj.close()
            except Exception:
                pass

    def run_setup(contacts, steps, warmup, gather):
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + rank + 7919 * args.seed)
        actions = torch.rand((steps + warmup, n), generator=g, device=dev, dtype=torch.float32) * 2 - 1
        if args.actions == "const1":
            actions.fill_(1.0)
        env_kw = dict(contacts=bool(contacts), max_newton=args.max_newton, envs_per_wave=args.envs_per_wave, variant=variant, per_env_model=args.augmented,
                      flags=(1 if args.no_rank_one else 0) | (8 if args.no_pair else 0) | (16 if args.no_spread else 0) | (32 if args.no_reorder else 0))
        if gather and os.environ.get("JB_BENCH_TEST_FAIL_GATHER") == "1":          # (every rank: an error a collective reports on all of them)
            raise RuntimeError("injected failure of the gather path on rank %d (tests/test_bench_launch.py)" % rank)
        if gather:
            # N > 1: the PRODUCT's sharded env (jitterbug_amd/distributed.py), pipelined: the step kernel writes packed rows
            # [obs | reward | done] itself and rank 0 gathers them every step over RCCL, one step late from a side stream, three row
            # buffers rotating (see ShardedJitterbugEnv).  Actions are resident on every rank (local_actions).
            from jitterbug_amd.distributed import ShardedJitterbugEnv
            sh = ShardedJitterbugEnv(n * world, task, seed=args.seed, device=dev, pipeline_depth=2, collective=gather, group=torch_group(gather), **env_kw)
            env = sh.env
        else:
            # the handle launches on torch's current stream so torch.cuda.Event brackets exactly these kernels
            sh = None
            env = JitterbugVecEnv(n, task, seed=args.seed, device_id=local_rank, env_offset=rank * n, stream=torch.cuda.current_stream(dev).cuda_stream, **env_kw)
        if args.augmented:           # BASELINE configs[4]: one randomised model per env, generated on the device (keyed by the global env index)
            env.randomise_models(seed=1000, return_params=False)          # every draw of the reference's distribution is kept (robots whose mass touches a leg included: PAIR kernel)
        used_variant.append(env.kernel_variant)
        obs = torch.empty((n, D), device=dev, dtype=torch.float32)
        rew = torch.empty((n,), device=dev, dtype=torch.float32)
        done = torch.empty((n,), device=dev, dtype=torch.uint8)
        env.reset_device(None, obs.data_ptr())
        last = [None]

        actions_all = None
        if sh is not None and args.actions_from == "rank0" and rank == 0:          # the whole batch's actions on rank 0 (resident in ITS HBM), scattered every step
            actions_all = torch.rand((steps + warmup, n * world), generator=g, device=dev, dtype=torch.float32) * 2 - 1
            actions_all[:, :n] = actions

        def one(i):
            if sh is None:
                env.step_device(actions[i].data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
            else:
                # a PendingRows handle for step i-1: nothing waits unless asked
                r = sh.step(local_actions=actions[i]) if args.actions_from == "local" else sh.step(None if actions_all is None else actions_all[i])
                if r is not None:
                    last[0] = r

        def drain():
            if sh is not None:
                r = sh.flush()
                if r is not None:
                    last[0] = r

        for i in range(warmup):
            one(i)
        drain()
        finite_warm[0] = bool(torch.isfinite(sh.last_local_rows() if (sh is not None and warmup) else obs).all().item()) if warmup else True
        return env, sh, one, drain, obs, last

    finite_warm = [True]

    def torch_group(mode):
        """the process group the sharded env's torch.distributed calls use: the default one, or - rows through PyTorch's RCCL while the default
        group is gloo - an nccl subgroup built on first use (collective: every rank comes here together)"""
        if mode != "torch" or dist.get_backend() != "gloo" or args.dist_backend == "gloo":
            return None
        if nccl_group[0] is None:
            nccl_group[0] = dist.new_group(backend="nccl")
        return nccl_group[0]

    def run_finish(env, sh, obs, last, wall, dev_ms):
        sc, ep, cap = env.counters()
        lastrows = sh.last_local_rows()[:, :D] if sh is not None else obs
        finite = finite_warm[0] and bool(torch.isfinite(lastrows).all().item())      # checked after the warm-up and after the timed steps
        finite = finite and float(cap.max()) < 1000.0                             # ... and no env ever ended a step non-finite (kernel-side flag)
        if sh is not None and rank == 0:                # the gathered block of the last step really holds every rank's rows
            ob_all, rw_all, dn_all = last[0].get()
            finite = finite and ob_all.shape[0] == n * world and bool(torch.isfinite(ob_all).all().item()) and bool((ob_all[:n] == sh.last_local_rows()[:, :D]).all().item())
        if sh is not None:
            sh.close()          # flush, both streams idle, the library's communicator destroyed (cabi), then the env
        else:
            env.close()
        return wall, dev_ms, float(cap.sum()), finite

    measured = None
    while True:
        try:
            measured = run(args.contacts, K, W, gather=use_gather)
            break
        except RanksDisagree as e:
            if use_gather is None:
                raise
            # the data path failed somewhere: try the next one; the shards do not need any to advance - in the end they are timed alone, and
            # the line says so in a field a driver can read (`degraded`)
            dist_notes.append("row gather through %s failed (%s)" % ({"cabi": "the library's RCCL binding (jb_gather_rows_device)", "torch": "torch.distributed"}[use_gather], str(e)[:300]))
            gather_modes = gather_modes[1:]
            use_gather = gather_modes[0] if gather_modes else None
    wall, dev_ms, cap_hits, finite = measured
    degraded = dist is not None and use_gather is None
    if dist is not None:
        dist_notes.append("NO data-path collective: the ranks were timed as independent shards" if degraded else
                          "rows gathered through %s" % {"cabi": "the library's own RCCL communicator (jb_comm_init / jb_gather_rows_device), torch.distributed carries control traffic only (gloo)",
                                                         "torch": "torch.distributed (%s)" % ("RCCL" if args.dist_backend != "gloo" else "gloo, staged through the host: a rehearsal")}[use_gather])
    wall_max = rank_max(wall)
    total_envs = n * world
    value = total_envs * K / wall_max

    # Next to the headline window (whatever --steps/--warmup the caller chose: with the driver's 5 + 20 that is steps 5-25 of an
    # episode, every robot still upright), the same workload in the same process over two fixed windows:
    #   steady        100 warm-up + 300 timed steps (steps 100-400 of the episode; the window the committed rocprofv3 profile covers)
    #   full_episode  0 + 1000 steps from the reset, including the in-kernel auto-reset of the last step
    steady = full_episode = None
    if not args.no_steady:
        ws, ds, _, fs = run(args.contacts, 300, 100, gather=use_gather)
        ws = rank_max(ws)
        steady = {"value": total_envs * 300 / ws, "unit": "env steps/s", "ms_per_step": ws * 1e3 / 300, "launch_ms": ds / 300, "steps": 300, "warmup": 100,
                  "window": "steps 100-400 of the episode (some robots have tipped over by then; the window of profiles/r06_*)", "finite": fs}
        if world == 1 and dist is None:
            wf, df, _, ff = run(args.contacts, 1000, 0, gather=False)
            full_episode = {"value": total_envs * 1000 / wf, "unit": "env steps/s", "ms_per_step": wf * 1e3 / 1000, "launch_ms": df / 1000, "steps": 1000, "warmup": 0,
                            "window": "steps 0-1000: one whole episode from the reset, auto-reset included", "finite": ff}

    # The fused K-step rollout (jb_step_many_device: one launch of K control steps, every wave keeps its envs for all of them), next to the
    # per-step numbers above and on the same workload / seed / action tape: K = 1000 (a whole episode in one launch), K = 100 (ten
    # launches), the in-kernel heuristic policy instead of the tape, and the motor-flat-out tape.  Its bound is the slowest wave's SUM
    # over the K steps, measured by the kernel itself (jb_wave_clocks: s_memrealtime per wave) and reported against the mean wave.
    rollout_fused = None
    if not args.no_steady and world == 1 and dist is None:
        def fused(k_launch, source, total=1000, episodes=1):
            g = torch.Generator(device=dev)
            g.manual_seed(1234 + rank + 7919 * args.seed)
            tape = torch.rand((total, n), generator=g, device=dev, dtype=torch.float32) * 2 - 1
            if source == "const1":
                tape.fill_(1.0)
            env = JitterbugVecEnv(n, task, seed=args.seed, device_id=local_rank, env_offset=rank * n, stream=torch.cuda.current_stream(dev).cuda_stream,
                                  contacts=bool(args.contacts), max_newton=args.max_newton, envs_per_wave=args.envs_per_wave, variant=variant, per_env_model=args.augmented,
                                  flags=(1 if args.no_rank_one else 0) | (8 if args.no_pair else 0) | (16 if args.no_spread else 0) | (32 if args.no_reorder else 0))
            if args.augmented:
                env.randomise_models(seed=1000, return_params=False)
            rew = torch.empty((total, n), device=dev, dtype=torch.float32)
            obs = torch.empty((n, D), device=dev, dtype=torch.float32)
            done = torch.empty((n,), device=dev, dtype=torch.uint8)
            env.reset_device(None, obs.data_ptr())
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            per_episode = []
            for ep_i in range(episodes):          # consecutive episodes (the tape repeats; resets draw new poses): a fused launch lasts as long as
                te = time.perf_counter()          # the wave whose robot tips over FIRST, so one episode's figure depends on that robot's luck
                for k0 in range(0, total, k_launch):
                    env.step_many_device(k_launch, None if source == "policy" else tape[k0:k0 + k_launch].data_ptr(), rewards_ptr=rew[k0:k0 + k_launch].data_ptr(),
                                         obs_last_ptr=obs.data_ptr(), done_last_ptr=done.data_ptr())
                torch.cuda.synchronize(dev)
                per_episode.append((time.perf_counter() - te) * 1e3 / total)
            wall = (time.perf_counter() - t0) / episodes
            wc = env.wave_clocks()            # of the last launch
            sc, ep, cap = env.counters()
            ok = bool(torch.isfinite(obs).all().item()) and bool(torch.isfinite(rew).all().item()) and float(cap.max()) < 1000.0 and bool(done.all().item())
            epw = env.envs_per_wave
            kv = env.kernel_variant
            env.close()
            out = {"value": n * total / wall, "unit": "env steps/s", "ms_per_step": wall * 1e3 / total, "launches": total // k_launch, "steps_per_launch": k_launch,
                   "actions": {"tape": "uniform action tape [K,N] resident in HBM (the headline's actions)", "policy": "heuristic policy evaluated in the kernel (no action buffer)",
                               "const1": "motor flat out (about half the robots tip over)"}[source],
                   "window": "steps 0-%d: %d whole episode(s) from the reset, auto-reset included" % (total, episodes), "finite": ok, "kernel_variant": kv,
                   "ms_per_step_by_episode": per_episode}
            if k_launch == total:
                simds = 1024.0
                out["wave_clock"] = {"mean_wave_ms_per_step": 1e3 * float(wc.mean()) / total, "slowest_wave_ms_per_step": 1e3 * float(wc.max()) / total,
                                     "p99_wave_ms_per_step": 1e3 * float(np.percentile(wc, 99)) / total, "mean_over_slowest": float(wc.mean() / wc.max()),
                                     "ceiling_mean_wave": epw * min(simds, len(wc)) / (float(wc.mean()) / total),
                                     "value_over_ceiling": (n * total / wall) / (epw * min(simds, len(wc)) / (float(wc.mean()) / total)),
                                     "what": "per-wave lifetimes of the launch (s_memrealtime): the launch lasts as long as its slowest wave; `ceiling_mean_wave` = envs per wave / mean wave time x waves in flight - what the launch would reach if no wave were slower than the mean.  A robot that has tipped over stays tipped for the rest of its episode (tools/tip_persistence.py), so the slowest wave is the one whose robot tipped first"}
            return out
        rollout_fused = {"k1000": fused(1000, "tape", episodes=3), "k100": fused(100, "tape", episodes=3), "k1000_policy": fused(1000, "policy"), "k1000_const1": fused(1000, "const1"),
                         "per_step_full_episode": None if full_episode is None else full_episode["value"],
                         "what": "jb_step_many_device: K control steps per launch, bit-identical to K single-step launches (tests/test_gpu_rollout.py); `value` above stays the per-step path"}

    # N > 1: the same fused rollout across the shards (ShardedJitterbugEnv.rollout: K steps in one launch per rank, then ONE gather of the
    # [K, N_local, D+2] blocks - a hundred times fewer, a hundred times larger collectives than the per-step path).  Next to the headline,
    # never instead of it; a failure here is recorded, it does not take the line down.
    if not args.no_steady and dist is not None and use_gather:
        # every rank passes the same sync points whatever happens to it (like run()): a rank that fails alone brings its failure to the next
        # sync point instead of leaving the others in it
        failure, sh, tape = None, None, None
        try:
            from jitterbug_amd.distributed import ShardedJitterbugEnv
            sh = ShardedJitterbugEnv(n * world, task, seed=args.seed, device=dev, pipeline_depth=1, variant=variant, per_env_model=args.augmented, collective=use_gather, group=torch_group(use_gather),
                                     contacts=bool(args.contacts), max_newton=args.max_newton, envs_per_wave=args.envs_per_wave)
            if args.augmented:
                sh.env.randomise_models(seed=1000, return_params=False)
            g = torch.Generator(device=dev)
            g.manual_seed(1234 + rank + 7919 * args.seed)
            tape = torch.rand((1000, n), generator=g, device=dev, dtype=torch.float32) * 2 - 1
            sh.env.reset_device()
        except Exception as e:
            failure = e
        try:
            sync_point(failure)
            t0 = time.perf_counter()
            last = None
            try:
                for k0 in range(0, 1000, 100):
                    last = sh.rollout(100, local_actions=tape[k0:k0 + 100])
            except Exception as e:
                failure = e
            sync_point(failure)
            wf = rank_max(time.perf_counter() - t0)
            ok = True if rank != 0 else bool(torch.isfinite(last[0]).all().item()) and last[0].shape[1] == n * world
            rollout_fused = {"k100_sharded": {"value": total_envs * 1000 / wf, "unit": "env steps/s", "ms_per_step": wf * 1e3 / 1000, "launches": 10, "steps_per_launch": 100,
                                              "gathers": 10, "rows_per_gather": [100, n, D + 2], "finite": ok, "collective": use_gather,
                                              "what": "ShardedJitterbugEnv.rollout(100): one fused 100-step launch per rank, then one gather of [100, N_local, D+2] rows to rank 0; x 10 = a whole episode"}}
            sh.close()
        except Exception as e:            # (RanksDisagree on every rank, or a local error after the last sync point: recorded, the headline stands)
            rollout_fused = {"k100_sharded": {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}}
            try:
                if sh is not None:
                    sh.close()
            except Exception:
                pass

    def config_label(contacts):
        """which BASELINE.json config this workload is, if any"""
        if args.augmented:
            return "BASELINE configs[4]'s per-GPU shard" if n == 8192 else "augmented models, not a BASELINE size"
        if task == "move_to_pose" and n == N_ENVS_PER_GPU and contacts:
            return "BASELINE configs[3]'s per-GPU shard"
        if task == TASK and n == N_ENVS_PER_GPU:
            return "BASELINE configs[2]" if contacts else "BASELINE configs[1]"
        return "not a BASELINE config"

    also = None
    if not args.no_also and world == 1:
        w2, d2, _, _ = run(1 - args.contacts, max(50, K // 5), 20, gather=False)
        also = {"workload": "%s N_envs=%d contacts %s (%s)" % (task, n, "off" if args.contacts else "on", config_label(1 - args.contacts)),
                "value": n * max(50, K // 5) / w2, "unit": "env steps/s"}

    host_rate = None
    if world == 1 and not args.no_host_rate:
        # the reference-shaped entry point jb_step (host pointers: H2D actions, D2H obs/reward/done, stream sync per step) on the same
        # workload, for the record next to `value` (which has inputs resident in HBM): the PCIe-inclusive rate
        import numpy as np
        env = JitterbugVecEnv(n, task, seed=args.seed, device_id=local_rank, contacts=bool(args.contacts), max_newton=args.max_newton, envs_per_wave=args.envs_per_wave)
        if args.augmented:
            env.randomise_models(seed=1000, return_params=False)          # every draw of the reference's distribution is kept (robots whose mass touches a leg included: PAIR kernel)
        env.reset()
        rs = np.random.default_rng(0)
        acts = rs.uniform(-1, 1, size=(64, n)).astype(np.float32)
        kh = min(K, 300)
        for i in range(min(W, 50)):
            env.step(acts[i % 64])
        t0 = time.perf_counter()
        for i in range(kh):
            env.step(acts[i % 64])
        host_rate = {"value": n * kh / (time.perf_counter() - t0), "unit": "env steps/s", "steps": kh,
                     "what": "jb_step with host buffers through the Python VecEnv (H2D actions + kernel + D2H obs/reward/done + sync + numpy copies per step)"}
        # the same through step_async / step_wait (jb_step_async: pinned staging, the copies and the kernel queued at once, an event to wait on) with
        # the caller's own work between the halves - here: drawing the next step's actions, what a policy would be doing
        # A caller with work of its own between the halves (the learner's bookkeeping: here 0.3 ms of numpy per step, independent of the
        # step's results): with step() the two add up, with step_async / step_wait the GPU steps underneath
        Wm = rs.normal(size=(256, 256)).astype(np.float32)

        def host_work():
            x = Wm
            t_end = time.perf_counter() + 0.0003
            while time.perf_counter() < t_end:
                x = np.tanh(x @ Wm * 1e-2)
            return x
        t0 = time.perf_counter()
        for i in range(kh):
            env.step(acts[i % 64]); host_work()
        sync_with_work = n * kh / (time.perf_counter() - t0)
        env.step_async(acts[0]); env.step_wait()
        t0 = time.perf_counter()
        for i in range(kh):
            env.step_async(acts[i % 64])
            host_work()
            env.step_wait(copy=False)
        host_rate["async"] = {"value": n * kh / (time.perf_counter() - t0), "unit": "env steps/s", "steps": kh, "sync_with_the_same_host_work": sync_with_work, "host_work_ms_per_step": 0.3,
                              "what": "step_async -> 0.3 ms of the caller's own numpy work -> step_wait(copy=False) (jb_step_async / jb_step_wait: pinned buffers, copies and kernel queued at once, an event to wait on, results read in place) against step() followed by the same work"}
        env.close()

    if rank == 0:
        # HBM traffic and issue counters per launch come from rocprofv3 PMC passes (separate runs: FETCH_SIZE, WRITE_SIZE, SQ_*), which
        # cannot be collected inside this process.  They are read from the committed summary ONLY when it was collected on this very
        # build (sha256 of the library) and this workload; otherwise they are null rather than stale.
        import hashlib
        from jitterbug_amd import _lib
        lib_sha = hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()
        traffic = None
        compute = None
        PROFILE = "profiles/r06_pmc_raw.json"
        prof_note = "no PMC summary for this build/workload (tools/collect_profiles.sh + tools/summarise_profiles.py write one)"
        try:
            raw = json.load(open(os.path.join(ROOT, PROFILE)))
            if raw.get("lib_sha256") != lib_sha:
                prof_note = PROFILE + " was collected on another build of libjitterbug_hip.so: not quoted"
            elif n == N_ENVS_PER_GPU and args.contacts and task == TASK and not args.augmented and world == 1:
                traffic = (raw["FETCH_SIZE_KB"] + raw["WRITE_SIZE_KB"]) * 1024.0
                sq = raw["sq"]
                waves = sq["SQ_WAVES"]
                simds = 1024.0
                k_s = raw["kernel_avg_ns"] * 1e-9
                compute = {"bound": "fp32 VALU issue of one wave per SIMD",
                           "window": "steps 100-400 (bench.py --steps 300 --warmup 100 under rocprofv3: the `steady` block's window, NOT the headline's)",
                           "kernel_avg_ms": raw["kernel_avg_ns"] * 1e-6,
                           "waves_per_launch": waves, "waves_per_simd": waves / simds,
                           "valu_insts_per_launch": sq["SQ_INSTS_VALU"],
                           "valu_issue_frac_of_wave_life": sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"],
                           "any_issue_frac_of_wave_life": sq["SQ_ACTIVE_INST_ANY"] / sq["SQ_WAVE_CYCLES"],
                           "waitcnt_frac_of_wave_life": sq.get("SQ_WAIT_ANY", 0.0) / sq["SQ_WAVE_CYCLES"],
                           "mean_wave_life_ms": raw.get("mean_wave_life_ms"),
                           "mean_wave_life_over_launch": raw.get("mean_wave_life_ms") / (raw["kernel_avg_ns"] * 1e-6) if raw.get("mean_wave_life_ms") else None,
                           # issue slots: a SIMD can issue one wave64 VALU instruction per 2 cycles (two waves resident), a lone wave one per 4
                           "valu_issue_slot_frac_of_chip": sq["SQ_INSTS_VALU"] * 2.0 / (simds * k_s * raw.get("clock_hz", 2.4e9)),
                           "source": PROFILE + " (rocprofv3 PMC passes of this build and workload)"}
                if all(k in sq for k in ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_TRANS_F32")):
                    # fp32 operations per launch: wave-instructions by type (an FMA counts 2) x the lanes that really executed them
                    wave_ops = 2.0 * sq["SQ_INSTS_VALU_FMA_F32"] + sq["SQ_INSTS_VALU_ADD_F32"] + sq["SQ_INSTS_VALU_MUL_F32"] + sq["SQ_INSTS_VALU_TRANS_F32"]
                    lanes = sq["SQ_THREAD_CYCLES_VALU"] / sq["SQ_ACTIVE_INST_VALU"] if sq.get("SQ_THREAD_CYCLES_VALU") and sq.get("SQ_ACTIVE_INST_VALU") else None
                    compute["packed_note"] = "since round 6 a fifth of the VALU instructions are packed (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two fp32 operations per lane); the SQ_INSTS_VALU_*_F32 counters count each ONCE, so the FLOP figures below are lower bounds"
                    compute.update({"fp32_wave_insts_per_launch": {"fma": sq["SQ_INSTS_VALU_FMA_F32"], "add": sq["SQ_INSTS_VALU_ADD_F32"], "mul": sq["SQ_INSTS_VALU_MUL_F32"], "trans": sq["SQ_INSTS_VALU_TRANS_F32"]},
                                    "mean_active_lanes_per_valu_inst": lanes,
                                    "fp32_tflops_full_wave": wave_ops * 64.0 / k_s / 1e12,          # rocprof-compute's convention: every instruction counted at 64 lanes
                                    "fp32_tflops_active_lanes": (wave_ops * lanes / k_s / 1e12) if lanes else None,
                                    "active_lanes_note": "active lanes are NOT useful lanes: since round 3 lane group 1 repeats the main lanes' phase A / C (it needs the joint-space system for the damping factorisation it takes off the critical path) and the solves run unmasked in all four groups; since round 5 lane groups 2 and 3 run phase A / C on the motor body and the root body's own mass (group 3 repeating group 2 for the replica); one env's physics occupies 4 + 2 of its 16 lanes",
                                    "fp32_peak_tflops": 157.3,
                                    "fp32_frac_of_peak_active_lanes": (wave_ops * lanes / k_s / 1e12 / 157.3) if lanes else None})
                prof_note = "%s, same build (sha256 %s...)" % (PROFILE, lib_sha[:12])
            else:
                prof_note = PROFILE + " covers move_from_origin, N=4096, contacts on, 1 GPU only"
        except Exception as e:
            prof_note = "no usable %s (%s)" % (PROFILE, type(e).__name__)
        launch_s = dev_ms * 1e-3 / K                 # HIP events on the kernel's stream around the K timed launches
        # 317 B for move_from_origin (D=15); other tasks add 4 B per extra obs entry and the 12 B target read; per-env models add
        # the lane constant table (202 x 4 floats) read once per step
        algo_bytes = ALGO_BYTES_PER_ENV_STEP + 4 * (D - 15) + (12 if task != TASK else 0) + (202 * 4 * 4 if args.augmented else 0)
        achieved = algo_bytes * n / launch_s / 1e9
        gather_txt = ""
        if degraded:
            gather_txt = ", NO data-path collective in this run (the shards are independent; see dist_notes)"
        elif world > 1 or dist is not None:
            how = {"cabi": "the library's own RCCL communicator (jb_gather_rows_device: grouped ncclSend / ncclRecv)", "torch": "torch.distributed (%s)" % ("RCCL" if args.dist_backend != "gloo" else "gloo rehearsal, staged through the host")}[use_gather]
            gather_txt = ", gather of [N,D+2] rows to rank 0 every step through %s, issued one step late from a side stream, three row buffers (jitterbug_amd.distributed.ShardedJitterbugEnv, pipeline_depth=2)" % how
            gather_txt += {"local": "; actions resident on every rank (the timed region is step -> gather)",
                           "rank0": "; actions of the whole batch on rank 0, scattered every step (the timed region is scatter -> step -> gather: the whole per-step round trip)"}[args.actions_from]
        res = {
            "metric": "env steps/s at N_envs=%d, %s" % (n, task),
            "value": None if degraded else value, "unit": "env steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall_max * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, N_envs=%d per GPU, %s, 50 substeps/step, in-kernel auto-reset (%s)"
                                   % (task + (", one randomised model per env" if args.augmented else ""), n, ("full Newton contact solve" if args.contacts else "contacts off") + {"ordinary": "", "pair": ", PAIR kernel variant (floor + mass / upper-leg contact)", "lean": ", LEAN kernel variant (two waves per SIMD)", "lean_pair": ", LEAN + PAIR kernel variant (two waves per SIMD, one model per env)"}[used_variant[0]], config_label(args.contacts)),
                       "global_envs": total_envs, "parallelism": "env-sharded x%d%s" % (world, gather_txt)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "kernel": "jb_step_kernel", "launch_ms": launch_s * 1e3,
                         "window": "the headline's: steps %d-%d of an episode from the reset (achieved / frac / launch_ms); `steady` below repeats them for steps 100-400, the window of `traffic` and `compute`" % (W, W + K),
                         "steady": None if steady is None else {"launch_ms": steady["launch_ms"], "achieved": algo_bytes * n / (steady["launch_ms"] * 1e-3) / 1e9,
                                                                  "frac": algo_bytes * n / (steady["launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "window": "steps 100-400"},
                         "launch_ms_is": "average jb_step_kernel launch" if (dist is None or not use_gather) else "average step INCLUDING the stream waits on the row gather (N > 1 path), not the bare kernel",
                         "algorithmic_bytes_per_launch": algo_bytes * n,
                         "note": "path is fp32-VALU/latency bound, not HBM bound (SURVEY.md §8d): 317 B per env step vs ~1e5-1e6 dependent flops",
                         "compute": compute, "profile": prof_note},
            "solver_cap_hits": cap_hits, "finite": finite, "lib_sha256": lib_sha, "src_sha256": _lib.load().jb_source_sha256().decode(), "kernel_variant": used_variant[0], "dist_notes": dist_notes,
            "data_path_collective": (None if dist is None else (use_gather or False)), "degraded": degraded, "actions_from": (None if dist is None else args.actions_from),
        }
        if degraded:          # a multi-GPU line without the gather is NOT the multi-GPU result: no headline value, the shards' rate on the side, rc 3
            res["value_independent_shards"] = value
            res["degraded_why"] = "every data path for the per-step row gather failed (dist_notes); `value_independent_shards` is communication-free and must not be read as scaling"
        res["window"] = "steps %d-%d of an episode from the reset" % (W, W + K)
        if steady:
            res["steady"] = steady
        if full_episode:
            res["value_full_episode"] = full_episode["value"]
            res["full_episode"] = full_episode
        if rollout_fused:
            res["rollout_fused"] = rollout_fused
        if also:
            res["also"] = also
        if host_rate:
            res["host_buffer_rate"] = host_rate
        if world == 1 and not args.no_cpu_baseline:
            cores = usable_cores()
            res["cpu_baseline"] = cpu_baseline(cores, task)
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()
    return 3 if (dist is not None and use_gather is None) else 0


if __name__ == "__main__":
    sys.exit(main() or 0)
