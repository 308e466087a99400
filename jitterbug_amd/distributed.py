"""Multi-GPU sharding of a Jitterbug batch: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The path shards by construction: environments never interact (reference jitterbug.py:601-925 has no cross-env
term; the reference's own parallelism is process-level env replication, benchmarks/benchmark.py:146-171).  Rank r owns
the contiguous global env range [r*N/R, (r+1)*N/R); RNG streams are keyed by the GLOBAL env index (jb_config.env_offset)
so results do not depend on R.  The only exchange per control step is the return of results to the host-facing rank:
a gather of packed rows [N_local, D+2] = obs | reward | done (fp32), and the scatter of actions the other way.
There is no all-reduce anywhere.  On 8 MI355X the gather is 7 concurrent single-hop xGMI sends into rank 0.

Two data paths for the rows (ShardedJitterbugEnv(collective=...)):
  "torch"  torch.distributed's gather (RCCL through PyTorch; gloo for CPU rehearsals)
  "cabi"   the library's own RCCL binding (jb_comm_init / jb_gather_rows_device / jb_gather_block_device: grouped ncclSend / ncclRecv to
           rank 0 on a stream of the caller's choosing) - the north-star's "thin C-ABI ... RCCL-over-xGMI gather", with no torch.distributed
           call on the data path: the process group only carries the 128-byte communicator id once (any group will do; bench.py uses gloo)
           and actions that rank 0 chooses to scatter."""
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


class PendingRows:
    """Result of a pipelined step (pipeline_depth = 2): `get()` makes the CURRENT stream wait for the gather of that step and returns
    (obs, reward, done) on rank 0 (None elsewhere).  Nothing waits until the consumer asks: the step kernels launched in the meantime
    do not queue behind the gather.  The data stay valid until two more steps have been issued."""

    def __init__(self, owner, index=None, value=None):
        self._owner, self._index, self._value = owner, index, value

    def get(self):
        if self._index is not None:
            if self._owner._i > self._index + self._owner._NB:
                raise RuntimeError("PendingRows of step %d expired: its row buffers were reused (%d steps issued since; keep at most %d in flight)"
                                   % (self._index, self._owner._i - self._index - 1, self._owner._NB - 1))
            self._value = self._owner._result(self._index)
            self._index = None
        return self._value


class _EventPending:
    """the `work` handle of a gather issued through the C ABI: wait() makes the CURRENT stream wait for the event recorded behind it,
    wait_on(stream) a stream of the caller's choosing.  The event belongs to the row buffer and is re-recorded by its next gather."""

    def __init__(self, ops, event):
        self._ops, self._event = ops, event

    def wait(self):
        self._ops.wait_event(self._ops.current_stream(), self._event)

    def wait_on(self, stream):
        self._ops.wait_event(stream, self._event)


class _CudaStreamOps:
    """the stream / event vocabulary the C-ABI data path needs, on torch.cuda (torch is plumbing here: streams, events, buffers)"""

    def __init__(self, device):
        import torch
        self.torch, self.device = torch, device

    def side_stream(self):
        return self.torch.cuda.Stream(device=self.device)

    def event(self):
        with self.torch.cuda.device(self.device):
            return self.torch.cuda.Event()

    def synchronize(self, stream):
        stream.synchronize()

    def record(self, event, stream):
        event.record(stream)

    def wait_event(self, stream, event):
        stream.wait_event(event)

    def current_stream(self):
        return self.torch.cuda.current_stream(self.device)

    def raw(self, stream):
        return stream.cuda_stream


class ShardedJitterbugEnv:
    """One shard of a global batch per rank.  `local_env_factory(n_local, env_offset)` builds the local stepper
    (default: JitterbugVecEnv on this rank's GPU); it must offer reset()/step(actions) over numpy arrays or the
    *_device entry points (used when tensors live on the GPU).

    pipeline_depth = 1: step() returns this step's gathered results (scatter -> kernel -> gather, blocking in stream order).
    pipeline_depth = 2: step() returns a PendingRows handle for the PREVIOUS step's results (None on the first call; flush() returns
    the last one): `handle.get()` yields (obs, reward, done) when the consumer needs them.  The gather of step t is issued from a side stream after step t+1's kernel has been launched, so the RCCL kernel runs in the tail of step
    t+1 on SIMDs whose waves have finished instead of delaying its start (the step kernel holds one wave per SIMD, an RCCL kernel
    cannot co-reside with it), and three row buffers rotate so that no step waits for the gather issued just before it.  This is
    the path `bench.py --gpus N` times (the reference's own vectorisation, stable-baselines SubprocVecEnv, has the same
    step_async / step_wait split: benchmarks/benchmark.py:146-171)."""

    def __init__(self, n_global, task="move_from_origin", seed=0, device=None, local_env_factory=None, group=None, pipeline_depth=1, variant="auto", collective="torch",
                 stream_ops=None, **env_kwargs):
        """variant: "auto" (default) resolves the step kernel from the GLOBAL batch and the world size (jitterbug_amd.variants) and takes
        precedence over a JB_FLAG_LEAN bit in `flags`: below the threshold "auto" CLEARS that bit.  A caller who wants to force a kernel by
        flags passes variant=None (and then owns the duty of giving every shard the same one); passing both "auto" and JB_FLAG_LEAN is refused.
        collective: "torch" | "cabi" (module docstring).  stream_ops: the stream / event vocabulary of the "cabi" path (tests pass a stub)."""
        import torch
        import torch.distributed as dist
        from . import variants
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.n_global = int(n_global)
        if collective not in ("torch", "cabi"):
            raise ValueError("collective must be 'torch' or 'cabi'")
        self.collective = collective
        if variant == "auto" and (int(env_kwargs.get("flags", 0)) & 2):
            raise ValueError("variant='auto' chooses the kernel from the global batch and would clear JB_FLAG_LEAN from `flags` below its threshold: "
                             "pass variant='lean' (every shard), or variant=None to keep the flags as they are")
        # the kernel variant is resolved from the GLOBAL batch and the world size - numbers every rank holds - so that all shards of the batch
        # run the same kernel (the variants agree to rounding only); a caller that passes flags by hand (variant=None) owns that duty
        self.variant = None if variant is None else variants.resolve(variant, variants.envs_per_gpu(n_global, self.world), bool(env_kwargs.get("per_env_model", False)))
        self.sizes = [shard_range(n_global, r, self.world)[1] - shard_range(n_global, r, self.world)[0] for r in range(self.world)]
        self.lo, self.hi = shard_range(n_global, self.rank, self.world)
        self.n_local = self.hi - self.lo
        self.device = torch.device(device if device is not None else "cpu")
        if local_env_factory is None:
            from .vec_env import JitterbugVecEnv

            def local_env_factory(n_local, env_offset):
                kw = dict(env_kwargs)
                if self.variant is not None:
                    kw["variant"] = self.variant
                if self.device.type == "cuda" and "stream" not in kw:       # launch on torch's stream: ordered with the collectives
                    kw["stream"] = torch.cuda.current_stream(self.device).cuda_stream
                return JitterbugVecEnv(n_local, task, seed=seed, env_offset=env_offset,
                                       device_id=self.device.index or 0, **kw)
        self.env = local_env_factory(self.n_local, self.lo)
        self.on_gpu = self.device.type == "cuda"
        if pipeline_depth not in (1, 2):
            raise ValueError("pipeline_depth must be 1 or 2")
        self.depth = int(pipeline_depth)
        self._cabi = collective == "cabi"
        self._device_rows = (self.on_gpu or (self._cabi and stream_ops is not None)) and hasattr(self.env, "step_rows_device")
        if self._cabi:
            if not self._device_rows:
                raise ValueError("collective='cabi' needs the device path (a GPU env with step_rows_device)")
            # the communicator id: made by rank 0, handed round once over the process group (control traffic; 128 bytes)
            box = [self.env.comm_unique_id() if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            try:
                self.env.comm_init(self.world, self.rank, box[0])
                # the partition, the same list on every rank: the exchanges move blocks of the longest shard (uneven splits pad to it), and a rank
                # whose env does not hold what the list says is refused here instead of hanging the first exchange
                self.env.comm_set_shards(self.sizes)
            except Exception:
                env, self.env = self.env, None          # a half-built object leaves nothing behind: no communicator, no handle
                try:
                    env.close()
                finally:
                    raise
            self._ops = stream_ops if stream_ops is not None else _CudaStreamOps(self.device)
        # (in the row-buffer logic below "nccl" means: rows stay on the device and the gather is ordered by streams - true of both RCCL paths)
        self._nccl = self._device_rows and (self._cabi or dist.get_backend(group) == "nccl")
        self._i = 0                      # steps issued
        self._prev = None                # depth 2 on the blocking paths: the result held back one step
        if self._device_rows:
            D2 = self.env.obs_dim + 2
            nmax = max(self.sizes)
            # the step kernel writes its n_local rows into a buffer padded to the longest shard (gather needs equal shapes); the
            # buffers persist, nothing is allocated or concatenated per step
            self._NB = 3 if (self._nccl and self.depth == 2) else (2 if self.depth == 2 else 1)
            self._rows = [torch.zeros((nmax, D2), device=self.device, dtype=torch.float32) for _ in range(self._NB)]
            stage_dev = self.device if self._nccl else "cpu"
            # rank 0 receives into ONE contiguous block per buffer, [world, nmax, D+2]: with equal shards the result is a view of it
            self._blocks = [torch.empty((self.world, nmax, D2), device=stage_dev, dtype=torch.float32) for _ in range(self._NB)] if self.rank == 0 else [None] * self._NB
            self._gathered = [[blk[r] for r in range(self.world)] for blk in self._blocks] if self.rank == 0 else [None] * self._NB
            self._stage = None if self._nccl else [torch.empty((nmax, D2), dtype=torch.float32).pin_memory() for _ in range(self._NB)]
            self._pending = [None] * self._NB
            if self._cabi:
                self._side = self._ops.side_stream() if self.depth == 2 else None
                self._env_stream = self._ops.env_stream(self.env) if hasattr(self._ops, "env_stream") else (torch.cuda.ExternalStream(int(self.env.stream), device=self.device) if self.env.stream else torch.cuda.default_stream(self.device))
                self._nvtx = False
                self._step_done = [self._ops.event() for _ in range(self._NB)] if self._side is not None else None
                # one "gather done" event per row buffer and one for rollouts, re-recorded: nothing is created per step
                self._gather_done = [self._ops.event() for _ in range(self._NB)]
                self._rollout_done = self._ops.event()
                # the action half of the round trip (jb_scatter_actions_device): rank 0 stages every shard's block, every rank receives its own
                self._act_local = torch.zeros((nmax,), device=self.device, dtype=torch.float32)
                self._act_all = torch.zeros((self.world, nmax), device=self.device, dtype=torch.float32) if self.rank == 0 else None
            else:
                self._side = torch.cuda.Stream(device=self.device) if (self._nccl and self.depth == 2) else None
                # The step kernel runs on the ENV's stream (the one captured when it was built, or its own): events are recorded on, and
                # the gathers ordered against, that stream - not whatever torch stream happens to be current when step() is called.
                sp = self.env.stream
                self._env_stream = torch.cuda.ExternalStream(int(sp), device=self.device) if sp else torch.cuda.default_stream(self.device)
                try:
                    torch.cuda.nvtx.range_push("jb_sharded_env"); torch.cuda.nvtx.range_pop()
                    self._nvtx = True
                except Exception:
                    self._nvtx = False
                self._step_done = [torch.cuda.Event() for _ in range(self._NB)] if self._side is not None else None
            self._late = None            # index of the step whose rows still have to be sent (depth 2)

    def last_local_rows(self):
        """this rank's packed rows [n_local, D+2] of the last step issued (device path)"""
        return self._rows[(self._i - 1) % self._NB][:self.n_local]

    def _to_tensor(self, a, dtype):
        import torch
        return torch.as_tensor(np.asarray(a), dtype=dtype, device=self.device)

    def reset(self):
        import torch
        self.flush()
        obs = self._to_tensor(self.env.reset(), torch.float32)
        rows = pack_rows(obs, torch.zeros(self.n_local, device=self.device), torch.zeros(self.n_local, device=self.device))
        out = gather_rows(rows, self.sizes, 0, self.group)
        return None if out is None else unpack_rows(out)[0]

    # ------------------------------------------------------------------ device path
    def _send(self, j):
        """issue the gather of step j's rows (buffer j % NB)"""
        import torch
        import torch.distributed as dist
        b = j % self._NB
        if self._nccl and self._nvtx:             # a named range on the timeline (torch's nvtx shim is roctx on ROCm): where the gather sits vs jb_step
            torch.cuda.nvtx.range_push("jb_gather_rows")
            try:
                return self._send_impl(j, b)
            finally:
                torch.cuda.nvtx.range_pop()
        return self._send_impl(j, b)

    def _send_impl(self, j, b):
        import torch
        import torch.distributed as dist
        if self._cabi:
            # the library's own collective (jb_gather_rows_device: grouped ncclSend / ncclRecv to rank 0) on the side stream (depth 2: it waits
            # only for the event of the step that wrote the rows) or on the env's stream (depth 1); an event behind it is what consumers wait for
            ops = self._ops
            st = self._side if self._side is not None else self._env_stream
            if self._side is not None:
                ops.wait_event(self._side, self._step_done[b])
            self.env.gather_rows_device(self._rows[b].data_ptr(), self._blocks[b].data_ptr() if self.rank == 0 else None, ops.raw(st))
            ops.record(self._gather_done[b], st)
            self._pending[b] = _EventPending(ops, self._gather_done[b])
            return
        if self._side is not None:
            with torch.cuda.stream(self._side):
                self._side.wait_event(self._step_done[b])
                self._pending[b] = dist.gather(self._rows[b], self._gathered[b] if self.rank == 0 else None, dst=0, group=self.group, async_op=True)
        elif self._nccl:
            with torch.cuda.stream(self._env_stream):       # RCCL orders the collective behind the stream current at issue time
                self._pending[b] = dist.gather(self._rows[b], self._gathered[b] if self.rank == 0 else None, dst=0, group=self.group, async_op=True)
        else:                                            # gloo rehearsal: staged through pinned host memory, synchronous
            with torch.cuda.stream(self._env_stream):
                self._stage[b].copy_(self._rows[b], non_blocking=True)
            self._env_stream.synchronize()
            dist.gather(self._stage[b], self._gathered[b] if self.rank == 0 else None, dst=0, group=self.group)

    def _result(self, j):
        """rows of step j on rank 0 (after its gather), as (obs, reward, done) views; None elsewhere"""
        import torch
        b = j % self._NB
        if self._pending[b] is not None:
            self._pending[b].wait()                      # the current stream waits for the gather (stream order, no host sync); the handle stays:
                                                         # the ENV's stream waits for it too before the buffer is rewritten (_step_device)
        if self.rank != 0:
            return None
        if min(self.sizes) == max(self.sizes):
            out = self._blocks[b].view(-1, self._blocks[b].shape[-1])
        else:
            out = torch.cat([g[:n] for g, n in zip(self._gathered[b], self.sizes)], 0)
        return unpack_rows(out if self._nccl else out.to(self.device))

    def _step_device(self, a):
        import torch
        i = self._i
        b = i % self._NB
        if self._pending[b] is not None:                 # the buffer is rewritten only after the gather that read it:
            if self._cabi:                               # the ENV's stream (where the kernel that rewrites it runs) waits for that gather
                self._pending[b].wait_on(self._env_stream)
            else:
                with torch.cuda.stream(self._env_stream):
                    self._pending[b].wait()
            self._pending[b] = None
        if self.on_gpu:
            cur = torch.cuda.current_stream(self.device)
            if cur.cuda_stream != self._env_stream.cuda_stream:      # actions produced on another stream: the kernel waits for them.  (RAW handles:
                self._env_stream.wait_stream(cur)                    # a pool Stream and an ExternalStream never compare equal even when both wrap one hipStream)
        self.env.step_rows_device(a.data_ptr(), self._rows[b].data_ptr())
        self._i += 1
        if self.depth == 1:
            self._send(i)
            return self._result(i)
        if self._side is not None:
            if self._cabi:
                self._ops.record(self._step_done[b], self._env_stream)
            else:
                self._step_done[b].record(self._env_stream)
        prev = self._late
        self._late = i
        if prev is None:
            return None
        self._send(prev)                                 # one step late: it overlaps the kernel just launched
        return PendingRows(self, prev)

    def rollout(self, n_steps, actions_global=None, local_actions=None):
        """n_steps control steps on every shard in ONE kernel launch per rank (jb_step_many_device: every wave keeps its envs for all K
        steps), then ONE gather of the [K, N_local, D+2] row blocks to rank 0 - K times fewer, K times larger collectives than step()'s.
        actions: rank 0 passes a tape [K, N_global] (scattered step by step), or every rank its own [K, N_local] float32 device tensor as
        `local_actions`, or nothing at all: the task's heuristic policy is evaluated inside the kernel (the reference's
        benchmarks/evaluate_policy.py:29-33 loop, for the whole sharded batch).  Returns (obs [K, N, D], reward [K, N], done [K, N]) on
        rank 0, None elsewhere.  Results equal n_steps calls of step(): bit for bit on the device path."""
        import torch
        import torch.distributed as dist
        K = int(n_steps)
        self.flush()
        D2 = self.env.obs_dim + 2
        nmax = max(self.sizes)
        tape = None
        if local_actions is not None:                                  # (every rank must use the same way of passing actions)
            tape = local_actions
        else:
            ag = None
            if self.rank == 0 and actions_global is not None:
                ag = torch.as_tensor(np.asarray(actions_global), dtype=torch.float32).reshape(K, self.n_global)
            have = torch.tensor([1 if ag is not None else 0], dtype=torch.int32, device=self.device if dist.get_backend(self.group) == "nccl" else "cpu")
            dist.broadcast(have, src=0, group=self.group)              # does rank 0 bring a tape, or does the in-kernel policy act?
            if int(have.item()):
                tape = torch.stack([scatter_actions(None if ag is None else ag[k], self.sizes, self.device, 0, self.group) for k in range(K)], 0)
        rows = torch.zeros((K, nmax, D2), device=self.device, dtype=torch.float32)
        if self._device_rows:
            local = rows if self.n_local == nmax else torch.zeros((K, self.n_local, D2), device=self.device, dtype=torch.float32)
            cur = torch.cuda.current_stream(self.device) if self.on_gpu else None
            if cur is not None and cur.cuda_stream != self._env_stream.cuda_stream:
                self._env_stream.wait_stream(cur)                      # the tape / row buffers were produced on the current stream
            t = None if tape is None else tape.contiguous()
            self.env.step_many_device(K, None if t is None else t.data_ptr(), rows_ptr=local.data_ptr())
            if cur is not None and cur.cuda_stream != self._env_stream.cuda_stream:
                cur.wait_stream(self._env_stream)
            if local is not rows:
                rows[:, :self.n_local] = local
        else:                                                          # host stepper (tests: the oracle stands in for the GPU env)
            for k in range(K):
                if tape is None:
                    raise ValueError("the host stepper has no in-kernel policy: pass a tape")
                res = self.env.step(tape[k].cpu().numpy())
                rows[k, :self.n_local] = pack_rows(self._to_tensor(res[0], torch.float32), self._to_tensor(res[1], torch.float32), self._to_tensor(res[2], torch.float32))
        if self._cabi:          # ONE block gather through the library's own communicator, on the env's stream behind the launch
            blk = torch.empty((self.world, K, nmax, D2), device=rows.device, dtype=torch.float32) if self.rank == 0 else None
            self.env.gather_block_device(rows.data_ptr(), None if blk is None else blk.data_ptr(), K * nmax * D2, self._ops.raw(self._env_stream))
            self._ops.record(self._rollout_done, self._env_stream)
            self._ops.wait_event(self._ops.current_stream(), self._rollout_done)
            if self.rank != 0:
                return None
            out = torch.cat([blk[r][:, :n] for r, n in enumerate(self.sizes)], 1)          # [K, N_global, D+2]
            return out[..., :-2], out[..., -2], out[..., -1] > 0.5
        staged = rows.is_cuda and dist.get_backend(self.group) == "gloo"
        send = rows.cpu() if staged else rows
        bufs = [torch.empty_like(send) for _ in range(self.world)] if self.rank == 0 else None
        dist.gather(send.contiguous(), bufs, dst=0, group=self.group)
        if self.rank != 0:
            return None
        out = torch.cat([b[:, :n] for b, n in zip(bufs, self.sizes)], 1).to(self.device)          # [K, N_global, D+2]
        return out[..., :-2], out[..., -2], out[..., -1] > 0.5

    def _scatter_cabi(self, actions_global):
        """The action half of the round trip through the library's own communicator (jb_scatter_actions_device: grouped ncclSend x world on
        rank 0, one ncclRecv everywhere), on the env's stream - the step kernel that reads the block is queued right behind it.  Rank 0
        passes the actions of all N_global envs (numpy array or tensor, host or device); the other ranks pass None."""
        import torch
        nmax = self._act_local.shape[0]
        if self.rank == 0:
            a = torch.as_tensor(np.asarray(actions_global) if not torch.is_tensor(actions_global) else actions_global, dtype=torch.float32).reshape(-1)
            if a.shape[0] != self.n_global:
                raise ValueError("rank 0 passes one action per env of the global batch (%d), got %d" % (self.n_global, a.shape[0]))
            cur = self._ops.current_stream() if a.is_cuda else None
            if cur is not None and self._ops.raw(cur) != self._ops.raw(self._env_stream):
                self._env_stream.wait_stream(cur)              # device actions produced on the caller's stream
            with self._stream_ctx(self._env_stream):
                if min(self.sizes) == nmax:
                    self._act_all.view(-1).copy_(a, non_blocking=True)
                else:                                          # uneven shards: every block is padded to the longest
                    lo = 0
                    for r, n in enumerate(self.sizes):
                        self._act_all[r, :n].copy_(a[lo:lo + n], non_blocking=True)
                        lo += n
        self.env.scatter_actions_device(None if self._act_all is None else self._act_all.data_ptr(), self._act_local.data_ptr(), nmax, self._ops.raw(self._env_stream))
        return self._act_local[:self.n_local]

    def _stream_ctx(self, stream):
        import contextlib
        import torch
        return torch.cuda.stream(stream) if self.on_gpu else contextlib.nullcontext()

    def close(self):
        """Drains what is in flight (the late gather of depth 2), waits for the side stream and the env's stream, destroys the library's
        communicator (if this object made one) and then the env.  Safe to call twice; the object is unusable afterwards."""
        env = getattr(self, "env", None)
        if env is None:
            return
        try:
            if getattr(self, "_device_rows", False):
                try:
                    self.flush()
                finally:
                    for st in (getattr(self, "_side", None), getattr(self, "_env_stream", None)):
                        if st is not None:
                            (self._ops.synchronize(st) if self._cabi else st.synchronize())
            if getattr(self, "_cabi", False):
                env.comm_destroy()
        finally:
            self.env = None
            env.close()

    def flush(self):
        """depth 2: the results of the last step issued (None if there is none pending)"""
        if not getattr(self, "_device_rows", False):
            out, self._prev = getattr(self, "_prev", None), None
            return out
        if self.depth == 1 or self._late is None:
            return None
        prev, self._late = self._late, None
        self._send(prev)
        return PendingRows(self, prev)

    def step(self, actions_global=None, local_actions=None):
        """Rank 0 passes actions for all N_global envs (other ranks pass None) - or every rank passes its own slice as
        `local_actions` (a float32 tensor on this rank's device: actions already resident where they are consumed).
        depth 1: returns (obs, reward, done) of this step on rank 0 (None elsewhere).  depth 2: returns a PendingRows handle for the
        previous step (None on the first call)."""
        import torch
        if local_actions is not None:
            a = local_actions
        elif self._cabi:
            a = self._scatter_cabi(actions_global)
        else:
            a = scatter_actions(actions_global, self.sizes, self.device, 0, self.group)
        if self._device_rows:
            # device path: scatter -> step kernel writes the packed rows -> gather, nothing leaves HBM on the way
            return self._step_device(a.contiguous())
        res = self.env.step(a.cpu().numpy())
        obs, rew, done = res[0], res[1], res[2]
        rows = pack_rows(self._to_tensor(obs, torch.float32), self._to_tensor(rew, torch.float32), self._to_tensor(done, torch.float32))
        out = gather_rows(rows, self.sizes, 0, self.group)
        cur = None if out is None else unpack_rows(out)
        if self.depth == 1:
            return cur
        out, self._prev = self._prev, PendingRows(self, value=cur)
        return out
