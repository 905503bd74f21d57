"""(e) the one exchange step on the GPU: pack -> RCCL all-gather -> stitch, with a context that OWNS its HIP stream
(not torch's current stream).  The all-gather has to be ordered behind the pack kernel and ahead of the stitch kernel
on that stream; a single-rank "nccl" (= RCCL) group is enough to exercise the ordering and the payload layout."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu


def test_rccl_exchange_is_ordered_on_the_context_stream():
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import synth_lm as sl
    from test_lm_gpu import to_capi
    pkg = ge.load_package()
    from ssk_amd import capi
    if dist.is_initialized():
        pytest.skip("a process group is already up in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    old = {k: os.environ.get(k) for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    try:
        torch.cuda.set_device(0)
        rank, world = pkg.dist.init_from_env("nccl", always=True)
        assert (rank, world) == (0, 1)
        ctx = pkg.Context(0)                                     # owns a non-blocking stream
        assert ctx.stream() not in (0, None)
        assert ctx.stream() != torch.cuda.current_stream().cuda_stream
        S = 64
        wins, idxs, sh = sl.make_shared_submaps(n_sub=3, n_shared=S, seed0=40, per_landmark=3, n_kf=14, n_lm=220, submap_size=5, obs_per_kf=60, opt_depth=2)
        b = capi.LmBatch(ctx, wins)
        dev = torch.device("cuda:0")
        shared = torch.zeros(S * 12, dtype=torch.float64, device=dev)
        gathered = torch.zeros(S * 12, dtype=torch.float64, device=dev)
        pos = torch.from_numpy(sh["init"].reshape(-1).copy()).to(dev)
        d_idx = [torch.from_numpy(i.astype(np.int32)).to(dev) for i in idxs]
        torch.cuda.synchronize()
        prm = to_capi(sl.make_params())
        before = None
        for rep in range(3):                                     # no host synchronisation between the enqueued pieces
            b.reset()
            b.run(prm)
            if rep == 2:                                         # (a copy ordered on the CONTEXT's stream, like everything else here)
                with torch.cuda.stream(torch.cuda.ExternalStream(int(ctx.stream()), device=dev)):
                    before = pos.clone()
            b.pack_shared(rep, d_idx[rep], prm, shared)
            pkg.dist.exchange_shared(shared, gathered, stream=ctx.stream())
            ctx.stitch_shared(gathered, 1, S, pos)
            b.set_shared(rep, d_idx[rep], pos)
        ctx.synchronize()
        torch.cuda.synchronize()
        g = gathered.cpu().numpy()
        assert np.array_equal(g, shared.cpu().numpy())
        # the LAST repetition's window, after its solve: its own copy of the shared landmarks, then the update written back
        import shared_ref as sr
        exp = sr.consensus(g.reshape(1, S, 12), before.cpu().numpy().reshape(S, 3))
        assert np.allclose(pos.cpu().numpy().reshape(S, 3), exp, rtol=1e-10, atol=1e-12)
        assert np.array_equal(b.state(2)[1][idxs[2]], pos.cpu().numpy().reshape(S, 3))
        b.close()
        ctx.close()
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_c_abi_exchange_on_a_single_rank_communicator(pkg, ctx):
    """ssk_lm_exchange_shared: the all-gather through the C-ABI with an RCCL communicator created outside torch.distributed (ncclGetUniqueId +
    ncclCommInitRank, world size 1, through ctypes on the RCCL the process already holds): what the C++ facade of a multi-GPU deployment
    calls.  One rank: the gathered buffer is the rank's own blocks; the call is enqueued on the context's stream."""
    import ctypes as C
    import os
    import torch
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        S = 256
        dev = torch.device("cuda:0")
        shared = torch.arange(S * 12, dtype=torch.float64, device=dev) * 0.5
        gathered = torch.zeros(S * 12, dtype=torch.float64, device=dev)
        lib = ctx.lib
        lib.ssk_lm_exchange_shared.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        ctx._chk(lib.ssk_lm_exchange_shared(ctx.h, comm, C.c_void_p(shared.data_ptr()), C.c_void_p(gathered.data_ptr()), S))
        ctx.synchronize()
        assert torch.equal(gathered, shared)
        assert lib.ssk_lm_exchange_shared(ctx.h, None, C.c_void_p(shared.data_ptr()), C.c_void_p(gathered.data_ptr()), S) != 0      # no communicator
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
