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
