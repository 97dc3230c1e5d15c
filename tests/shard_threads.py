"""In-process collectives for `relp_shard_run` with G engines on ONE GPU (test infrastructure).

RCCL refuses two ranks on one device, so the native multi-GPU loop (the loop inside the library, collectives
through the `relp_shard_set_collectives` hooks) is exercised for G > 1 like this: every "rank" is a Python thread
that owns one engine and calls `relp_shard_run` (ctypes releases the GIL for the duration of the call); the
hooks are ctypes callbacks that meet at a `threading.Barrier` and copy the ranks' device buffers with
`hipMemcpyDtoD` (`torch.Tensor.copy_` on views of the raw pointers).  Same call order, same buffers, same
message layout as with RCCL; only the transport differs.
"""
import ctypes as C
import threading

ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


def _view(torch, ptr, n_doubles, device):
    iface = {"shape": (n_doubles,), "typestr": "<f8", "data": (int(ptr), False), "version": 3}
    holder = type("_DevBuf", (), {"__cuda_array_interface__": iface})()
    return torch.as_tensor(holder, device=device)


class ThreadWorld:
    """Shared state of the G ranks: a barrier and the send pointers of the collective in flight."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.ptrs = [0] * world
        self.errors = []


class ThreadRank:
    def __init__(self, shared: ThreadWorld, rank: int, lib, handle, torch, device):
        self.shared, self.rank, self.lib, self.h, self.torch, self.device = shared, rank, lib, handle, torch, device
        self.calls = {"allgather": 0, "allreduce": 0}
        self._ag = ALLGATHER_FN(self._allgather)
        self._ar = ALLREDUCE_FN(self._allreduce)
        st = lib.relp_shard_set_collectives(handle, C.cast(self._ag, C.c_void_p), C.cast(self._ar, C.c_void_p), None)
        assert st == 0, lib.relp_last_error(handle).decode()

    # every rank: publish the send pointer, wait for all, gather / sum from everybody's buffer, wait again so that
    # nobody overwrites a buffer another rank is still reading
    def _exchange(self, send_ptr, work):
        sh, torch = self.shared, self.torch
        try:
            torch.cuda.synchronize()                     # the engine's kernels that produced `send`
            sh.ptrs[self.rank] = int(send_ptr)
            sh.barrier.wait(timeout=60)
            work(list(sh.ptrs))
            torch.cuda.synchronize()
            sh.barrier.wait(timeout=60)
            return 0
        except Exception as e:                           # a broken barrier on one rank breaks it for all
            import traceback
            sh.errors.append(repr(e) + " @ " + traceback.format_exc().splitlines()[-3].strip())
            sh.barrier.abort()
            return 1

    def _allgather(self, ctx, send, recv, nbytes, stream):
        self.calls["allgather"] += 1
        n = nbytes // 8

        def work(ptrs):
            out = _view(self.torch, recv, n * self.shared.world, self.device)
            for g, p in enumerate(ptrs):
                out[g * n:(g + 1) * n].copy_(_view(self.torch, p, n, self.device))
        return self._exchange(send, work)

    def _allreduce(self, ctx, buf, count, stream):
        self.calls["allreduce"] += 1
        total = {}

        def work(ptrs):
            total["v"] = self.torch.stack([_view(self.torch, p, count, self.device) for p in ptrs]).sum(dim=0)
        # two phases: everybody computes the sum from the untouched inputs, then everybody overwrites its buffer
        rc = self._exchange(buf, work)
        if rc:
            return rc

        def write(ptrs):
            _view(self.torch, buf, count, self.device).copy_(total["v"])
        return self._exchange(buf, write)


def run_ranks(world, target):
    """Run ``target(rank)`` on `world` threads; returns the list of results (exceptions are re-raised)."""
    results, errors = [None] * world, []

    def body(r):
        try:
            results[r] = target(r)
        except BaseException as e:  # noqa: BLE001
            errors.append((r, e))
    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    if errors:
        raise errors[0][1]
    assert all(not t.is_alive() for t in threads), "a rank did not finish"
    return results
