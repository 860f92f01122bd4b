"""Multi-device solve behind the boundary (BASELINE north_star: "the batch shards naturally across the 8 GPUs").

Games are independent, so the only parallel pattern is the scenario split: contiguous shards of the batch, one device handle per
shard, `alg_newton_solve_async` on every handle, `alg_synchronize`, and one reduction of the integer counters.  There is no
data-path collective.  Two ways to drive it:

* one process, several devices: `ShardedGameProblem(..., devices=[0, 1, ...])` + `host.newton_solve(prob)` -- all launches are
  issued before the first synchronisation, so the devices (or the streams of one device, when a device is listed twice) overlap;
* one process per device (`torch.distributed`, RCCL): `make_shard` builds the rank's shard and `reduce_counters` is the only
  collective -- `bench.py --gpus N` is a thin caller of these two functions (tests/test_sharding_gloo.py runs them with
  world_size 2 on gloo).
"""
import numpy as np

from . import host, scenarios


class ShardedGameProblem:
    """`GameProblem` over several devices.  `x0` is (B, n); shard r owns the contiguous games shard_range(B, r, len(devices)) and
    the global scenario ids game_id0 + lo .. game_id0 + hi - 1 (all random inputs are keyed by global id, so the shard layout
    does not change them).  A per-game LQR block (arrays with a leading batch axis) is split with the batch.  All shards run the
    same kernel shape: `waves_per_game`, default = what the first shard's batch size selects automatically."""

    def __init__(self, N, dt, x0, model, opts, game_obj, game_con, devices=(0,), backend=None, game_id0=0, waves_per_game=None):
        x0 = np.ascontiguousarray(np.asarray(x0, dtype=np.float64).reshape(-1, model.n))
        self.B, self.devices = x0.shape[0], list(devices)
        if not self.devices:
            raise host.AlgamesError("ShardedGameProblem: empty device list")
        self.model, self.opts, self.game_obj, self.game_con, self.game_id0 = model, opts, game_obj, game_con, game_id0
        self.cuts = [scenarios.shard_range(self.B, r, len(self.devices)) for r in range(len(self.devices))]
        self.shards = []
        for (lo, hi), dev in zip(self.cuts, self.devices):
            if hi <= lo:
                continue
            self.shards.append(host.GameProblem(N, dt, x0[lo:hi], model, opts, _slice_obj(game_obj, lo, hi), game_con,
                                                backend=backend, device=dev, game_id0=game_id0 + lo))
        self.cuts = [c for c in self.cuts if c[1] > c[0]]
        # one kernel shape for all shards: the automatic choice depends on the batch size of a handle (team kernels for small
        # batches), and the shapes differ at rounding level -- a split must not change which arithmetic a game gets
        self.waves_per_game = int(waves_per_game) if waves_per_game is not None else int(self.shards[0].batch.get_waves_per_game())
        for s in self.shards:
            s.batch.set_waves_per_game(self.waves_per_game)
        self.stats = None

    # ---- the pieces of the GameProblem surface that make sense on a sharded batch
    def _sync_options(self):
        for s in self.shards:
            s._sync_options()

    def get_traj(self, which=0):
        return np.concatenate([s.batch.get_traj(which) for s in self.shards])

    def get_stats(self):
        return np.concatenate([s.batch.get_stats() for s in self.shards])

    def locate(self, game):
        for s, (lo, hi) in zip(self.shards, self.cuts):
            if lo <= game < hi:
                return s, game - lo
        raise IndexError(game)

    @property
    def pdtraj(self):
        parts = [s.pdtraj for s in self.shards]
        return host.PrimalDualTraj(np.concatenate([p.states for p in parts]), np.concatenate([p.controls for p in parts]),
                                   np.concatenate([p.duals for p in parts]))


def _slice_obj(obj, lo, hi):
    if obj.Qdiag.ndim != 3:
        return obj
    import copy
    o = copy.copy(obj)
    o.Qdiag, o.Rdiag = obj.Qdiag[lo:hi], obj.Rdiag[lo:hi]
    o.xf = obj.xf[lo:hi] if obj.xf.ndim == 3 else obj.xf
    o.uf = obj.uf[lo:hi] if obj.uf.ndim == 3 else obj.uf
    return o


def newton_solve_sharded(prob, init=True):
    """newton_solve!(prob) on every shard: all launches first (asynchronous, one stream per handle), then the synchronisations."""
    prob._sync_options()
    for s in prob.shards:
        s.batch.newton_solve_async(init=init, game_id0=s.game_id0)
    for s in prob.shards:
        s.batch.synchronize()
    summary = prob.get_stats()

    def history(game, **kw):
        s, g = prob.locate(game)
        return s.batch.get_history(g, **kw)
    prob.stats = host.Statistics(summary, history)
    for s, (lo, hi) in zip(prob.shards, prob.cuts):
        s.stats = host.Statistics(summary[lo:hi], s.batch.get_history)
    return None


def local_counters(prob):
    """[game-Newton-iterations, converged games, failed games] of a (sharded) problem after a solve."""
    st = prob.get_stats() if isinstance(prob, ShardedGameProblem) else prob.batch.get_stats()
    return [int(st["newton_iters"].sum()), int(st["converged"].sum()), int((st["status"] != 0).sum())]


# ---- one process per device ---------------------------------------------------------------------------------------------------
def make_shard(config, games_per_rank, rank, world, backend=None, device=0, **kw):
    """Rank `rank` of `world` owns the contiguous global scenario ids [rank*G, (rank+1)*G): all random inputs are
    keyed by global id, so the shard layout does not change them.  `config` is a scenario family of scenarios.make_problem."""
    lo, hi = scenarios.shard_range(games_per_rank * world, rank, world)
    ids = np.arange(lo, hi)
    return scenarios.make_problem(config, ids, backend=backend, device=device, **kw), ids


def reduce_counters(counts, elapsed, world, device, group=None, use_dist=None):
    """Sum of the per-rank integer counters and max of the per-rank wall time: the only collectives of a multi-process run.
    `use_dist` says whether the two all-reduces run (default: `world > 1`); `group` is the process group they run on (default: the
    caller's default group).  The function never probes `dist.is_initialized()` on its own: a library user with a process group of
    their own must not get a collective (summed over unrelated ranks, or a hang) because they reduced the counters of a one-rank job.
    `bench.py` passes `use_dist=True` for its world-of-one RCCL runs (`--force-dist`, torchrun with one rank)."""
    import torch
    tot = torch.tensor([int(c) for c in counts], dtype=torch.int64, device=device)
    tmax = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    if use_dist is None:
        use_dist = world > 1
    if use_dist:
        import torch.distributed as dist
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=group)
    return [int(v) for v in tot.tolist()], float(tmax.item())


def gather_shard_ranges(lo, hi, world, device, group=None, use_dist=None):
    """[[lo, hi), ...] of every rank's contiguous global scenario ids, in rank order (one all_gather of two int64; no collective for a
    world of one unless `use_dist`): what a caller checks to see that the shards tile the job."""
    import torch
    mine = torch.tensor([int(lo), int(hi)], dtype=torch.int64, device=device)
    if use_dist is None:
        use_dist = world > 1
    if not use_dist:
        return [[int(lo), int(hi)]]
    import torch.distributed as dist
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    return [[int(t[0]), int(t[1])] for t in out]
