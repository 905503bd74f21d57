"""Multi-GPU plumbing of the hot path: one process per GPU, torch.distributed ("nccl" = RCCL on ROCm,
"gloo" in the CPU tests).  Keyframe streams / submaps are partitioned over ranks with no data-path collective
except ONE all-gather per step of the shared-landmark blocks."""
import os


def init_from_env(backend, always=False):
    """Rendezvous from the torchrun environment (RANK / WORLD_SIZE / MASTER_*); returns (rank, world).
    `always` also creates the (single-rank) group when WORLD_SIZE is 1, to exercise the exchange path on one GPU."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if (world > 1 or always) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def partition(n_items, rank, world):
    """Contiguous share [begin, end) of n_items for this rank (weak-scaling benches give every rank its own batch;
    this is for a fixed global list of submaps)."""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def exchange_shared(shared, gathered, stream=None):
    """The one exchange step: every rank contributes `shared` ([n_shared*12] f64: position 3, g_f 3, upper H_ff 6 per
    landmark) and receives all of them rank-major in `gathered` ([world*n_shared*12]).  A single all-gather: the
    payload is tens of KB, i.e. latency-bound on xGMI, so no ring reduction; the fusion is done locally in rank order
    (ssk_lm_stitch_shared) and is therefore bit-identical on every rank.

    stream: the hipStream_t handle (int, `Context.stream()`) the pack and stitch kernels are enqueued on.  RCCL orders
    a collective against torch's CURRENT stream, and a context that owns its stream is not that one: with `stream`
    given the all-gather is issued with that stream current, i.e. behind the pack and ahead of the stitch."""
    import torch.distributed as dist
    if stream is not None and shared.is_cuda:
        import torch
        with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=shared.device)):
            dist.all_gather_into_tensor(gathered, shared)
    else:
        dist.all_gather_into_tensor(gathered, shared)
    return gathered


class SharedLandmarkAccelerator:
    """Anderson acceleration of the exchange rounds (BASELINE configs[4]).

    One round is a fixed-point map x -> G(x) on the consensus positions x of the shared landmarks: every rank solves its window with
    the shared landmarks held at x, the blocks are gathered, G(x) = x + (sum H)^-1 sum g.  Plain iteration x <- G(x) is block-coordinate
    descent on the joint problem: linear convergence (factor ~0.7 per round on the test problems, 70 rounds to 1e-6).  Anderson mixing
    over the last `m` rounds -- x_next = G(x_k) - dG gamma, gamma = argmin |f_k - dF gamma|, f = G(x) - x -- reaches the same fixed point
    (the joint optimum) in a fraction of the rounds.  Host arithmetic on 3 n_shared doubles in float64, the same on every rank (every
    rank holds the same gathered blocks), so the consensus stays bit-identical across ranks.

    usage per round:  x_next = acc.step(x, gx)   with x the positions the round started from and gx the stitched positions G(x)."""

    def __init__(self, m=5, restart_ratio=4.0):
        self.m, self.restart_ratio = int(m), float(restart_ratio)
        self.X, self.GX = [], []

    def reset(self):
        self.X, self.GX = [], []

    def step(self, x, gx):
        import numpy as np
        x = np.asarray(x, np.float64).reshape(-1)
        gx = np.asarray(gx, np.float64).reshape(-1)
        f = gx - x
        # a residual that grew by a large factor: the history no longer describes the map near here (an LM solve took another branch)
        if self.X and np.linalg.norm(f) > self.restart_ratio * np.linalg.norm(self.GX[-1] - self.X[-1]):
            self.reset()
        self.X.append(x.copy()); self.GX.append(gx.copy())
        if len(self.X) > self.m + 1:
            self.X.pop(0); self.GX.pop(0)
        if len(self.X) < 2:
            return gx.copy()
        F = [g - y for g, y in zip(self.GX, self.X)]
        dF = np.stack([F[i + 1] - F[i] for i in range(len(F) - 1)], 1)
        dG = np.stack([self.GX[i + 1] - self.GX[i] for i in range(len(F) - 1)], 1)
        gamma, *_ = np.linalg.lstsq(dF, f, rcond=1e-10)
        # an ill-conditioned history (non-finite or huge coefficients) would be written straight into the consensus positions: plain round instead
        if not np.all(np.isfinite(gamma)) or np.abs(gamma).max() > 1e3:
            self.reset()
            return gx.copy()
        return gx - dG @ gamma
