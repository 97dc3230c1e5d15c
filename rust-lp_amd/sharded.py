"""Multi-GPU pivot loop, one process per GPU, collectives through ``torch.distributed``
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Tableau engine (what bench.py --gpus N runs): the stored tableau columns are split over the ranks; b, the basis and
the pending-update block W are replicated.  Per pivot ONE all-gather: every rank sends [key, j, d_j, its tableau
column of j (m), the block minima of the ratio test]; every rank picks the same winner, runs the ratio test on the
winner's column and updates its own columns.  Both phases of any `MatrixData`: at the end of phase 1 the ranks pivot
the basic artificial variables out through the same exchange (the library calls back through the collective hooks),
redundant rows are removed on every rank.

Revised engine (explicit B^-1): structural columns of A and rows of B^-1 sharded; LPs with a full slack basis.
Per pivot, with G ranks and m rows (all messages are device buffers, no host sync):
  1. local PRICE over the owned columns            -> candidate [key, j, d_j, a_j (m)]
     all-gather of the G candidates                   (8 * (m + 3) B per rank; the tableau engine appends
                                                       its m/256 block minima of the ratio test)
     every rank picks the same winner: min key, then min j   (pivot_rule.rs:118 first-wins)
  2. local FTRAN slice alpha[rows of this rank]     -> all-gather of the G slices (8 * m / G B per rank)
     every rank runs the same ratio test on the full alpha (b, basis are replicated)
  3. the owner of pivot row r writes rho = row_r(B^-1) / alpha_r, everyone else zeros;
     a SUM all-reduce is the broadcast               (8 * m B)
  4. every rank updates its rows of B^-1 and its replicas of b, -pi, -obj, basis.
RCCL has no MINLOC, so step 1 gathers (key, j) pairs instead of emulating it with two
all-reduces.  The messages are <= 80 KB at m = 10,000: latency-bound on xGMI, not bandwidth-bound.

Two drivers of the same steps: `ShardedPivotLoop` issues the collectives through ``torch.distributed``
from Python (any backend; the gloo tests run it); `NativeShardedLoop` hands the loop to the library
(`relp_shard_run`), which calls RCCL itself between its kernels -- no interpreter in the pivot.
"""
from __future__ import annotations

# Import order: a process that mixes this package with PyTorch must import torch BEFORE the first engine is
# created (PyTorch ships its own HIP runtime; if librelp_engine.so has already brought in the system one, torch
# finds "No HIP GPUs").  bench.py and the tests do.

import ctypes as C
import sys
from typing import Tuple

from . import engine as _engine


class HipShardOps:
    """The shard entry points of the C ABI (include/relp_engine.h, `relp_shard_*`)."""

    def __init__(self, tableau: "_engine.Tableau"):
        self.t = tableau
        self.lib = _engine.load_library()
        self.h = tableau.handle
        self.refresh_lengths()
        self.update_block = self.lib.relp_update_block(self.h)
        self.tableau = int(tableau.config.engine) == _engine.ENGINE_TABLEAU

    def refresh_lengths(self):
        lo, hi, rlo, rhi, stride = (C.c_int32() for _ in range(5))
        self.lib.relp_shard_ranges(self.h, C.byref(lo), C.byref(hi), C.byref(rlo), C.byref(rhi), C.byref(stride))
        self.row_stride = stride.value
        self.candidate_len = self.lib.relp_shard_candidate_len(self.h)
        self.rho_len = self.lib.relp_shard_rho_len(self.h)

    def _ck(self, st):
        if st != 0:
            raise _engine.RelpError(f"shard call failed ({st}): {self.lib.relp_last_error(self.h).decode()}")

    def set_stream(self, stream_ptr: int):
        self.t.set_stream(stream_ptr)

    def price(self, cand):
        self._ck(self.lib.relp_shard_price(self.h, cand.data_ptr()))

    def select_column(self, cands, count: int):
        self._ck(self.lib.relp_shard_select_column(self.h, cands.data_ptr(), count))

    def ftran(self, alpha_slice):
        self._ck(self.lib.relp_shard_ftran(self.h, alpha_slice.data_ptr()))

    def ratio(self, slices, count: int, rho):
        self._ck(self.lib.relp_shard_ratio(self.h, slices.data_ptr(), count, rho.data_ptr()))

    def update(self, rho):
        self._ck(self.lib.relp_shard_update(self.h, rho.data_ptr()))

    def pivot(self):
        """Tableau engine: everything after the candidate exchange (no further collective)."""
        self._ck(self.lib.relp_shard_pivot(self.h))

    def poll(self) -> Tuple[int, int]:
        oc, it = C.c_int32(), C.c_int64()
        self._ck(self.lib.relp_poll(self.h, C.byref(oc), C.byref(it)))
        return oc.value, it.value

    def flush_begin(self, torch, device):
        """Snapshot of S' B0inv (own rows, zeros elsewhere) as a tensor view to all-reduce, or None."""
        ptr, ln = C.c_void_p(), C.c_int64()
        self._ck(self.lib.relp_shard_flush_begin(self.h, C.byref(ptr), C.byref(ln)))
        if ln.value == 0:
            return None
        if getattr(self, "_snap", None) is None or self._snap_ptr != ptr.value:
            # wrap the engine-owned buffer without copying (device memory plumbing only)
            iface = {"shape": (ln.value,), "typestr": "<f8", "data": (ptr.value, False), "version": 3}
            holder = type("_DevBuf", (), {"__cuda_array_interface__": iface})()
            self._snap = torch.as_tensor(holder, device=device)
            self._snap_ptr = ptr.value
        return self._snap

    def flush_end(self):
        self._ck(self.lib.relp_shard_flush_end(self.h))


class _NullContext:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class ShardedPivotLoop:
    """phase_one::primal / phase_two::primal (phase_one.rs:125, phase_two.rs:22) across ranks."""

    def __init__(self, tableau_or_ops, dist, device, poll_interval: int = 64):
        import torch
        self.torch = torch
        self.dist = dist
        self.ops = tableau_or_ops if hasattr(tableau_or_ops, "price") else HipShardOps(tableau_or_ops)
        self.world = dist.get_world_size()
        self.device = device
        self.poll_interval = max(1, poll_interval)
        o = self.ops
        self._alloc_buffers()
        self._block = int(getattr(o, "update_block", 0))
        self._since_flush = 0
        self.stream = None
        if device.type == "cuda":
            # collectives are ordered against the current stream: put the kernels on it too
            self.stream = torch.cuda.Stream(device=device)
            o.set_stream(self.stream.cuda_stream)
        if isinstance(o, HipShardOps):
            self._install_hooks()

    def _install_hooks(self):
        """The library's own collective steps (the removal of basic artificial variables at the end of phase 1,
        `relp_shard_run`) call back into ``torch.distributed`` through `relp_shard_set_collectives`."""
        torch, dist, world, device = self.torch, self.dist, self.world, self.device
        on_device = dist.get_backend() == "nccl"

        def view(ptr, n):
            iface = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 3}
            return torch.as_tensor(type("_DevBuf", (), {"__cuda_array_interface__": iface})(), device=device)

        def on_stream():
            return torch.cuda.stream(self.stream) if self.stream is not None else _NullContext()

        self.hook_calls = 0
        self.hook_error = None

        def allgather(ctx, send, recv, nbytes, stream):
            try:
                self.hook_calls += 1
                n = nbytes // 8
                with on_stream():
                    src, dst = view(send, n), view(recv, n * world)
                    if on_device:
                        dist.all_gather_into_tensor(dst, src)
                    else:                                           # gloo: staged through host memory
                        host = torch.empty(n * world, dtype=torch.float64)
                        dist.all_gather_into_tensor(host, src.cpu())
                        dst.copy_(host)
                return 0
            except Exception as e:                                  # noqa: BLE001  (reported through the status code)
                self.hook_error = repr(e)
                print(f"[rust_lp_amd.sharded] all-gather hook failed: {e!r}", file=sys.stderr)
                return 1

        def allreduce(ctx, buf, count, stream):
            try:
                with on_stream():
                    t = view(buf, count)
                    if on_device:
                        dist.all_reduce(t, op=dist.ReduceOp.SUM)
                    else:
                        host = t.cpu()
                        dist.all_reduce(host, op=dist.ReduceOp.SUM)
                        t.copy_(host)
                return 0
            except Exception as e:                                  # noqa: BLE001
                self.hook_error = repr(e)
                print(f"[rust_lp_amd.sharded] all-reduce hook failed: {e!r}", file=sys.stderr)
                return 1

        self._hook_allgather = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)(allgather)
        self._hook_allreduce = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)(allreduce)
        o = self.ops
        o._ck(o.lib.relp_shard_set_collectives(o.h, C.cast(self._hook_allgather, C.c_void_p),
                                               C.cast(self._hook_allreduce, C.c_void_p), None))

    def _alloc_buffers(self):
        """Message buffers; their lengths follow the number of rows (redundant rows are removed at the phase switch)."""
        o, torch, device = self.ops, self.torch, self.device
        if hasattr(o, "refresh_lengths"):
            o.refresh_lengths()
        f64 = torch.float64
        self.cand = torch.zeros(o.candidate_len, dtype=f64, device=device)
        self.cands = torch.zeros(o.candidate_len * self.world, dtype=f64, device=device)
        self.slice = torch.zeros(o.row_stride, dtype=f64, device=device)
        self.slices = torch.zeros(o.row_stride * self.world, dtype=f64, device=device)
        self.rho = torch.zeros(o.rho_len, dtype=f64, device=device)

    def _iteration(self):
        o, d = self.ops, self.dist
        o.price(self.cand)
        d.all_gather_into_tensor(self.cands, self.cand)
        o.select_column(self.cands, self.world)
        if getattr(o, "tableau", False):
            # dense tableau: the candidate message already carried the entering tableau column; the
            # rest of the pivot (and the flush) is local -- ONE collective per pivot
            o.pivot()
            return
        o.ftran(self.slice)
        d.all_gather_into_tensor(self.slices, self.slice)
        o.ratio(self.slices, self.world, self.rho)
        d.all_reduce(self.rho, op=d.ReduceOp.SUM)
        o.update(self.rho)
        self._since_flush += 1
        if self._block > 0 and self._since_flush >= self._block:
            self._flush()

    def _flush(self):
        """Deferred update: fold the last K pivots into the owned rows of B0inv.  The rows S' B0inv
        live on different ranks, so their snapshot is completed by one SUM all-reduce (8 K m bytes,
        once per K pivots)."""
        snap = self.ops.flush_begin(self.torch, self.device)
        if snap is not None:
            self.dist.all_reduce(snap, op=self.dist.ReduceOp.SUM)
            self.ops.flush_end()
        self._since_flush = 0

    def flush(self):
        """Fold the pending pivots into the stored representation now (every rank must call): the tableau engine's
        flush is local to the owned columns, the revised engine's needs the all-reduce of `_flush`."""
        if getattr(self.ops, "tableau", False):
            if hasattr(self.ops, "lib"):                       # (the numpy stand-ins of the CPU tests keep no block)
                self.ops._ck(self.ops.lib.relp_flush(self.ops.h))
        elif self._block > 0 and self._since_flush > 0:
            if self.stream is not None:
                with self.torch.cuda.stream(self.stream):
                    self._flush()
            else:
                self._flush()

    def _enqueue(self, count: int):
        if self.stream is not None:
            with self.torch.cuda.stream(self.stream):
                for _ in range(count):
                    self._iteration()
        else:
            for _ in range(count):
                self._iteration()

    def run(self, max_iters: int) -> Tuple[int, int]:
        """Up to ``max_iters`` basis changes; returns (iterations done, outcome).  Every rank takes
        the same decisions from the same gathered data, so the outcome is identical on all ranks."""
        oc, start = self.ops.poll()
        if oc != _engine.RUNNING:
            return 0, oc
        left = max_iters
        while left > 0:
            chunk = min(left, self.poll_interval)
            self._enqueue(chunk)
            left -= chunk
            oc, it = self.ops.poll()
            if oc != _engine.RUNNING:
                if oc == _engine.PHASE_ONE_DONE:
                    self._alloc_buffers()
                return it - start, oc
        oc, it = self.ops.poll()
        if oc == _engine.PHASE_ONE_DONE:
            self._alloc_buffers()
        return it - start, oc

    def finish_phase_one(self) -> int:
        """With a full slack basis phase 1 has no candidate: one PRICE proves it and switches."""
        return self.run(1)[1]

    def solve_relaxation(self, max_iters: int = 1 << 40) -> Tuple[int, int]:
        """Both phases (two_phase/mod.rs:30-76): returns (pivots, outcome).  The tableau engine runs phase 1 on any
        `MatrixData`; the revised engine needs a full slack basis."""
        done, oc = self.run(max_iters)
        if oc == _engine.PHASE_ONE_DONE:
            more, oc = self.run(max_iters - done)
            done += more
        return done, oc


class NativeShardedLoop:
    """The same loop inside the library: `relp_shard_run` enqueues kernels and RCCL collectives on the engine's
    stream (include/relp_engine.h, "native multi-GPU loop").  ``torch.distributed`` only carries the 128-byte
    RCCL unique id from rank 0 to the other ranks and the agreement that every rank attached."""

    ID_BYTES = 128

    def __init__(self, tableau: "_engine.Tableau", dist, device):
        import torch
        self.t = tableau
        self.lib = _engine.load_library()
        self.h = tableau.handle
        self.dist = dist
        on_device = dist.get_backend() == "nccl"
        buf_dev = device if on_device else "cpu"
        ident = (C.c_uint8 * self.ID_BYTES)()
        # every rank must be able to load RCCL before anybody enters ncclCommInitRank (a rank that cannot would
        # leave the others waiting there): probe locally, agree, and only then make the id that is used
        loadable = 1 if self.lib.relp_rccl_unique_id(ident) == 0 else 0
        agreed = torch.tensor([loadable], dtype=torch.int32, device=buf_dev)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        if int(agreed.item()) == 0:
            raise _engine.RelpError("no RCCL library could be loaded on at least one rank")
        status = 0
        if dist.get_rank() == 0:
            status = self.lib.relp_rccl_unique_id(ident)
        msg = torch.tensor([status & 0xFF] + list(ident), dtype=torch.uint8, device=buf_dev)
        dist.broadcast(msg, 0)
        msg = msg.cpu()
        if int(msg[0]) != 0:
            raise _engine.RelpError("relp_rccl_unique_id failed on rank 0: no RCCL library")
        for k in range(self.ID_BYTES):
            ident[k] = int(msg[1 + k])
        st = self.lib.relp_rccl_attach(self.h, ident)
        ok = torch.tensor([1 if st == 0 else 0], dtype=torch.int32, device=buf_dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            detail = self.lib.relp_last_error(self.h).decode() if st != 0 else "another rank failed"
            raise _engine.RelpError(f"relp_rccl_attach failed ({st}): {detail}")

    def run(self, max_iters: int) -> Tuple[int, int]:
        done, oc = C.c_int64(), C.c_int32()
        st = self.lib.relp_shard_run(self.h, max_iters, C.byref(done), C.byref(oc))
        if st != 0:
            raise _engine.RelpError(f"relp_shard_run failed ({st}): {self.lib.relp_last_error(self.h).decode()}")
        return done.value, oc.value

    def finish_phase_one(self) -> int:
        return self.run(1)[1]

    def flush(self):
        """relp_flush on every rank: local for the tableau engine, completed by the library's own all-reduce
        (the RCCL hooks are attached) for the revised engine."""
        st = self.lib.relp_flush(self.h)
        if st != 0:
            raise _engine.RelpError(f"relp_flush failed ({st}): {self.lib.relp_last_error(self.h).decode()}")
