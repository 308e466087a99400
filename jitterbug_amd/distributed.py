"""Multi-GPU sharding of a Jitterbug batch: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The path shards by construction: environments never interact (reference jitterbug.py:601-925 has no cross-env
term; the reference's own parallelism is process-level env replication, benchmarks/benchmark.py:146-171).  Rank r owns
the contiguous global env range [r*N/R, (r+1)*N/R); RNG streams are keyed by the GLOBAL env index (jb_config.env_offset)
so results do not depend on R.  The only exchange per control step is the return of results to the host-facing rank:
a gather of packed rows [N_local, D+2] = obs | reward | done (fp32), and the scatter of actions the other way.
There is no all-reduce anywhere.  On 8 MI355X the gather is 7 concurrent single-hop xGMI sends into rank 0."""
import numpy as np


def shard_range(n_global, rank, world):
    """Contiguous env range of `rank`; sizes differ by at most one."""
    base, rem = divmod(int(n_global), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_rows(obs, reward, done):
    """[N, D] obs, [N] reward, [N] done  ->  [N, D+2] float32 rows (torch tensors, any device)."""
    import torch
    out = torch.empty((obs.shape[0], obs.shape[1] + 2), dtype=torch.float32, device=obs.device)
    out[:, :-2] = obs
    out[:, -2] = reward
    out[:, -1] = done.to(torch.float32)
    return out


def unpack_rows(rows):
    return rows[:, :-2], rows[:, -2], rows[:, -1] > 0.5


def gather_rows(rows, sizes, dst=0, group=None):
    """Gather per-rank row blocks (possibly of different lengths) to `dst`; returns the concatenation there, else None."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    nmax = max(sizes)
    if rows.shape[0] < nmax:                       # pad to a common length (gather needs equal shapes)
        pad = torch.zeros((nmax - rows.shape[0], rows.shape[1]), dtype=rows.dtype, device=rows.device)
        rows = torch.cat([rows, pad], 0)
    device = rows.device
    if rows.is_cuda and dist.get_backend(group) == "gloo":      # gloo has no device gather: stage through the host (tests / rehearsal)
        rows = rows.cpu()
    bufs = [torch.empty_like(rows) for _ in range(world)] if rank == dst else None
    dist.gather(rows.contiguous(), bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)], 0).to(device)


def scatter_actions(actions_global, sizes, device, src=0, group=None):
    """Rank `src` holds actions for every env [N_global]; every rank receives its slice [N_local]."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    nmax = max(sizes)
    staged = torch.device(device).type == "cuda" and dist.get_backend(group) == "gloo"
    final_device, device = device, ("cpu" if staged else device)
    out = torch.empty((nmax,), dtype=torch.float32, device=device)
    chunks = None
    if rank == src:
        a = torch.as_tensor(actions_global, dtype=torch.float32, device=device).reshape(-1)
        chunks, lo = [], 0
        for n in sizes:
            c = torch.zeros((nmax,), dtype=torch.float32, device=device)
            c[:n] = a[lo:lo + n]
            chunks.append(c)
            lo += n
    dist.scatter(out, chunks, src=src, group=group)
    return out[:sizes[rank]].to(final_device)


class ShardedJitterbugEnv:
    """One shard of a global batch per rank.  `local_env_factory(n_local, env_offset)` builds the local stepper
    (default: JitterbugVecEnv on this rank's GPU); it must offer reset()/step(actions) over numpy arrays or the
    *_device entry points (used when tensors live on the GPU)."""

    def __init__(self, n_global, task="move_from_origin", seed=0, device=None, local_env_factory=None, group=None, **env_kwargs):
        import torch
        import torch.distributed as dist
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.n_global = int(n_global)
        self.sizes = [shard_range(n_global, r, self.world)[1] - shard_range(n_global, r, self.world)[0] for r in range(self.world)]
        self.lo, self.hi = shard_range(n_global, self.rank, self.world)
        self.n_local = self.hi - self.lo
        self.device = torch.device(device if device is not None else "cpu")
        if local_env_factory is None:
            from .vec_env import JitterbugVecEnv

            def local_env_factory(n_local, env_offset):
                kw = dict(env_kwargs)
                if self.device.type == "cuda" and "stream" not in kw:       # launch on torch's stream: ordered with the collectives
                    kw["stream"] = torch.cuda.current_stream(self.device).cuda_stream
                return JitterbugVecEnv(n_local, task, seed=seed, env_offset=env_offset,
                                       device_id=self.device.index or 0, **kw)
        self.env = local_env_factory(self.n_local, self.lo)
        self.on_gpu = self.device.type == "cuda"

    def _to_tensor(self, a, dtype):
        import torch
        return torch.as_tensor(np.asarray(a), dtype=dtype, device=self.device)

    def reset(self):
        import torch
        obs = self._to_tensor(self.env.reset(), torch.float32)
        rows = pack_rows(obs, torch.zeros(self.n_local, device=self.device), torch.zeros(self.n_local, device=self.device))
        out = gather_rows(rows, self.sizes, 0, self.group)
        return None if out is None else unpack_rows(out)[0]

    def step(self, actions_global):
        """Rank 0 passes actions for all N_global envs (other ranks pass None); returns (obs, reward, done) on rank 0."""
        import torch
        a = scatter_actions(actions_global, self.sizes, self.device, 0, self.group)
        if self.on_gpu and hasattr(self.env, "step_rows_device"):
            # device path: scatter -> step kernel writes the packed rows -> gather, nothing leaves HBM on the way
            a = a.contiguous()
            rows = torch.empty((self.n_local, self.env.obs_dim + 2), dtype=torch.float32, device=self.device)
            self.env.step_rows_device(a.data_ptr(), rows.data_ptr())
            out = gather_rows(rows, self.sizes, 0, self.group)
            return None if out is None else unpack_rows(out)
        res = self.env.step(a.cpu().numpy())
        obs, rew, done = res[0], res[1], res[2]
        rows = pack_rows(self._to_tensor(obs, torch.float32), self._to_tensor(rew, torch.float32), self._to_tensor(done, torch.float32))
        out = gather_rows(rows, self.sizes, 0, self.group)
        return None if out is None else unpack_rows(out)
