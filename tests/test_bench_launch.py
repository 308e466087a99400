"""bench.py as the driver starts it: `python bench.py --gpus N` launches its own ranks (the reference's vectorisation spawns its
workers from one command too: benchmarks/benchmark.py:146-171, SubprocVecEnv), and the line it prints names the BASELINE workload
and the library it ran."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")

_PARENT_PROBE = r'''
import json, subprocess, sys
sys.path.insert(0, %(root)r)
seen = {}
class FakePopen:
    pid = 2 ** 22 + 12345          # (no such process: the watchdog is cancelled long before it could fire anyway)
    def __init__(self, cmd, stdout=None, env=None, text=None, **kw):
        seen["new_session"] = kw.get("start_new_session")
        seen["cmd"] = cmd; seen["env_ipc"] = (env or {}).get("HSA_ENABLE_IPC_MODE_LEGACY")
        seen["torch_loaded_at_spawn"] = any(m == "torch" or m.startswith("torch.") for m in sys.modules)
        self.stdout = iter(["some rank chatter\n", json.dumps({"metric": "env steps/s", "n_gpus": 2, "value": 1.0}) + "\n"])
    def wait(self):
        return %(rc)d
subprocess.Popen = FakePopen
import bench
rc = bench.main(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dist-backend", "gloo"])
seen["rc"] = rc
seen["torch_loaded_after"] = any(m == "torch" or m.startswith("torch.") for m in sys.modules)
sys.stderr.write("PROBE " + json.dumps(seen) + "\n")
sys.exit(rc)
'''


def _probe(rc):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-c", _PARENT_PROBE % {"root": ROOT, "rc": rc}], capture_output=True, text=True, env=env, timeout=120)
    seen = json.loads([l for l in p.stderr.splitlines() if l.startswith("PROBE ")][-1][6:])
    return p, seen


def test_gpus_n_parent_spawns_the_ranks_without_touching_torch_or_the_gpu():
    p, seen = _probe(0)
    assert p.returncode == 0
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-9] == os.path.abspath(BENCH) and cmd[-8:] == ["--gpus", "2", "--steps", "3", "--warmup", "1", "--dist-backend", "gloo"]
    assert seen["torch_loaded_at_spawn"] is False and seen["torch_loaded_after"] is False      # the parent never initialises HIP: it never even imports torch
    assert seen["env_ipc"] == "0" and seen["new_session"] is True          # own process group: the watchdog can end exactly these ranks
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2                              # exactly rank 0's line, chatter goes to stderr
    assert "some rank chatter" in p.stderr


def test_gpus_n_parent_returns_the_childs_failure_and_prints_no_result():
    p, seen = _probe(3)
    assert p.returncode == 3 and seen["rc"] == 3
    assert p.stdout.strip() == ""


def test_no_exec_in_bench():
    src = open(BENCH).read()
    assert "os.exec" not in src and "execv" not in src


@pytest.mark.gpu
def test_bench_line_names_the_baseline_workload_and_the_loaded_library():
    p = subprocess.run([sys.executable, BENCH, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-also", "--no-host-rate"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert "BASELINE configs[2]" in d["config"]["workload"] and "move_from_origin" in d["config"]["workload"] and "N_envs=4096" in d["config"]["workload"]
    from jitterbug_amd import _lib
    assert d["lib_sha256"] == hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["finite"] is True
    assert d["steady"]["steps"] == 300 and d["steady"]["warmup"] == 100 and d["steady"]["value"] > 0
    assert d["value_full_episode"] > 0 and d["roofline"]["steady"]["frac"] > 0
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-12
    # the kernel the line names is the kernel the handle launched (jb_kernel_variant), not a flag the script believed in
    assert d["kernel_variant"] == "ordinary" and "LEAN" not in d["config"]["workload"]
    # the fused rollout block next to the per-step headline: same workload, one launch per K steps, bounded by its slowest wave
    rf = d["rollout_fused"]
    for key in ("k1000", "k100", "k1000_policy", "k1000_const1"):
        assert rf[key]["finite"] is True and rf[key]["value"] > 0 and rf[key]["kernel_variant"] == "ordinary"
    assert rf["k1000"]["launches"] == 1 and rf["k100"]["launches"] == 10
    wc = rf["k1000"]["wave_clock"]
    assert 0 < wc["mean_over_slowest"] <= 1 and rf["k1000"]["value"] <= wc["ceiling_mean_wave"] * 1.02
    assert rf["k1000"]["value"] > 1.05 * d["value_full_episode"]          # what the fused launch is for (measured: 6.3 -> 7.5-8.3 M env-steps/s: it lasts as long as the wave whose robot tips over first)
    assert len(rf["k1000"]["ms_per_step_by_episode"]) == 3


@pytest.mark.gpu
def test_bench_resolves_the_kernel_variant_in_the_product_and_reports_the_one_launched():
    """No threshold lives in bench.py: `variant="auto"` is resolved by jitterbug_amd.variants (more than 4096 envs per GPU -> the two-waves-per-SIMD
    kernel; one model per env: from 8192), and the line reports jb_kernel_variant of the handle that was timed."""
    src = open(BENCH).read()
    assert "8192" not in src.split("def main")[1].split("import numpy")[0] and "16384" not in src.split("def main")[1].split("import numpy")[0]
    p = subprocess.run([sys.executable, BENCH, "--envs-per-gpu", "8192", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-also", "--no-host-rate", "--no-steady"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["kernel_variant"] == "lean" and "LEAN kernel variant" in d["config"]["workload"] and d["finite"] is True
    p = subprocess.run([sys.executable, BENCH, "--envs-per-gpu", "8192", "--augmented", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-also", "--no-host-rate", "--no-steady"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["kernel_variant"] == "lean_pair" and "LEAN + PAIR kernel variant" in d["config"]["workload"] and d["finite"] is True
    p = subprocess.run([sys.executable, BENCH, "--envs-per-gpu", "2048", "--augmented", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-also", "--no-host-rate", "--no-steady"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["kernel_variant"] == "pair" and "PAIR kernel variant" in d["config"]["workload"] and d["finite"] is True


@pytest.mark.gpu
def test_bench_gpus_2_starts_itself_on_one_gpu_over_gloo():
    """`python bench.py --gpus 2` (no torch.distributed.run in front) must produce one line with n_gpus 2.  On the
    one-GPU test box both ranks share device 0 (JB_BENCH_DEVICE) and the rows go over gloo; on an 8-GPU node the same command with the
    default backend is RCCL over xGMI."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["JB_BENCH_DEVICE"] = "0"
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dist-backend", "gloo", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-steady"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_envs"] == 8192 and d["finite"] is True and d["scaling"] == "weak"


@pytest.mark.gpu
def test_bench_gpus_2_a_failing_gather_gives_a_degraded_line_not_a_result():
    """The first 8-GPU run is also the first run of the row gather across real ranks.  If every gather path raises, every rank must leave
    together (no rank left behind in a barrier); the shards are then timed without the data-path collective - which they never needed to
    advance - but that is NOT the multi-GPU result: the line says `degraded`, carries no headline `value` (the shards' rate sits in
    `value_independent_shards`) and the run exits 3, so that a driver reading `value` / `n_gpus` cannot book it as scaling."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["JB_BENCH_DEVICE"] = "0"
    env["JB_BENCH_TEST_FAIL_GATHER"] = "1"
    env["JB_BENCH_LAUNCH_TIMEOUT"] = "150"
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dist-backend", "gloo", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-steady"],
                       capture_output=True, text=True, timeout=240, cwd=ROOT, env=env)
    assert p.returncode == 3, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    d = json.loads(lines[0])
    assert len(lines) == 1 and d["n_gpus"] == 2 and d["finite"] is True
    assert d["degraded"] is True and d["data_path_collective"] is False and d["value"] is None and d["value_independent_shards"] > 0
    assert any("injected failure" in x for x in d["dist_notes"]) and "NO data-path collective" in d["dist_notes"][-1] and "NO data-path collective" in d["config"]["parallelism"]


def test_gpus_n_parent_watchdog_kills_ranks_that_never_finish():
    """A rendezvous or collective that never completes must not hang the caller (the first 8-GPU run is also the first run of RCCL across
    real ranks): the parent kills the process group it started after JB_BENCH_LAUNCH_TIMEOUT and returns 124 without a result line."""
    probe = r'''
import os, subprocess, sys, time
sys.path.insert(0, %(root)r)
real = subprocess.Popen
def sleepy(cmd, **kw):
    return real([sys.executable, "-c", "import time; print('rank chatter', flush=True); time.sleep(600)"], **kw)
subprocess.Popen = sleepy
import bench
t0 = time.time()
rc = bench.main(["--gpus", "2", "--steps", "3", "--warmup", "1"])
sys.stderr.write("PROBE rc=%%d dt=%%.1f\\n" %% (rc, time.time() - t0))
sys.exit(rc)
''' % {"root": ROOT}
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["JB_BENCH_LAUNCH_TIMEOUT"] = "3"
    p = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode == 124 and p.stdout.strip() == ""
    assert "PROBE rc=124" in p.stderr and "rank chatter" in p.stderr and "killed" in p.stderr


@pytest.mark.gpu
def test_bench_gpus_2_over_rccl_on_one_gpu_succeeds_or_fails_cleanly():
    """Rehearsal of what the driver's 8-GPU tier does first, as far as a one-GPU box allows: `bench.py --gpus 2` with the DEFAULT data path
    (--collective torch: torch.distributed's gather over RCCL) and both ranks on device 0.  RCCL may accept two ranks on one
    device or refuse them (ncclInvalidUsage / "Duplicate GPU detected" - what this pool's does: torch's RCCL fails, the library's own is tried
    and fails the same way, the ranks then time their shards alone and say `degraded`).  Either way the run must end by itself well inside the
    watchdog's limit - with one result line (rc 0: gathered; rc 3: degraded) or with another rc and no line - and never hang."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["JB_BENCH_DEVICE"] = "0"
    env["JB_BENCH_LAUNCH_TIMEOUT"] = "150"
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-steady"],
                       capture_output=True, text=True, timeout=240, cwd=ROOT, env=env)
    dt = time.time() - t0
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    print("bench.py --gpus 2 over RCCL on one device: rc %d after %.0f s, %d result line(s); stderr tail: %s" % (p.returncode, dt, len(lines), p.stderr[-400:].replace("\n", " | ")))
    assert dt < 200 and p.returncode != 124, "the ranks hung until the watchdog"
    if p.returncode == 0:
        d = json.loads(lines[0])
        assert len(lines) == 1 and d["n_gpus"] == 2 and d["finite"] is True and d["degraded"] is False and d["value"] > 0
        assert d["data_path_collective"] in ("cabi", "torch") and "rows gathered through" in d["dist_notes"][-1]
    elif p.returncode == 3:
        d = json.loads(lines[0])
        assert len(lines) == 1 and d["n_gpus"] == 2 and d["degraded"] is True and d["value"] is None and d["value_independent_shards"] > 0
        assert "NO data-path collective" in d["dist_notes"][-1] and len(d["dist_notes"]) >= 3          # both RCCL paths were tried, and the line says why each failed
    else:
        assert lines == []


@pytest.mark.gpu
def test_bench_one_rank_through_the_librarys_own_collective():
    """JB_BENCH_FORCE_DIST=1: the N > 1 code path with a world of one - process group (gloo, control only), ShardedJitterbugEnv(collective=
    'cabi'): jb_comm_init with one rank, the pipelined jb_gather_rows_device on the side stream every step, the fused rollout's
    jb_gather_block_device - through real RCCL calls on a one-GPU box, and the line names the collective that ran."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update({"JB_BENCH_FORCE_DIST": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29631"})
    p = subprocess.run([sys.executable, BENCH, "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-also", "--no-host-rate", "--collective", "cabi", "--actions-from", "rank0"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["data_path_collective"] == "cabi" and d["degraded"] is False and d["finite"] is True and d["value"] > 0
    assert d["actions_from"] == "rank0" and "scatter -> step -> gather" in d["config"]["parallelism"]          # the whole round trip was inside the timed region
    assert "the library's own RCCL communicator" in d["dist_notes"][-1] and "jb_gather_rows_device" in d["config"]["parallelism"]
    assert d["rollout_fused"]["k100_sharded"]["finite"] is True and d["rollout_fused"]["k100_sharded"]["collective"] == "cabi"
