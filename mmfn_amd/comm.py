"""RCCL gradient exchange through the C ABI of libmmfn_comm.so (include/mmfn_comm.h).

Default transport of `parallel.DataParallel` is torch.distributed (backend "nccl" = RCCL on ROCm), which owns the
communicator and its streams.  This module is the alternative SURVEY.md section 8(b) asks for: a communicator created
through the C ABI (`mmfn_comm_*`) and collectives enqueued on a HIP stream of OUR choosing (`mmfn_allreduce_sum_f32`).
Collectives on a caller-owned stream can be captured into a hipGraph, so a data-parallel step on this transport is one graph.

The rendezvous id (128 bytes from rank 0) travels over the torch.distributed process group the launcher already set up
(any backend: it is a one-off object broadcast), RCCL itself is driven only through the C ABI.
"""
import ctypes
import gc
import os
import weakref

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmmfn_comm.so")
HEADER_PATH = os.path.join(_HERE, "..", "include", "mmfn_comm.h")
ID_BYTES = 128
_lib = None


class MMFNCommError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MMFNCommError("libmmfn_comm.so not found at %s - run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        h = ctypes.CDLL(LIB_PATH)
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        h.mmfn_comm_abi_version.restype = i32
        h.mmfn_comm_unique_id.argtypes = [vp]
        h.mmfn_comm_init.argtypes = [ctypes.POINTER(vp), vp, i32, i32]
        h.mmfn_comm_destroy.argtypes = [vp]
        h.mmfn_comm_ranks.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32)]
        h.mmfn_allreduce_sum_f32.argtypes = [vp, vp, i64, vp]
        h.mmfn_allreduce_sum_bf16.argtypes = [vp, vp, i64, vp]
        h.mmfn_broadcast_bytes.argtypes = [vp, vp, i64, i32, vp]
        for name in ("mmfn_comm_unique_id", "mmfn_comm_init", "mmfn_comm_destroy", "mmfn_comm_ranks", "mmfn_allreduce_sum_f32",
                     "mmfn_allreduce_sum_bf16", "mmfn_broadcast_bytes"):
            getattr(h, name).restype = i32
        _lib = h
    return _lib


def _check(rc, what):
    if rc != 0:
        raise MMFNCommError("%s failed with code %d (positive = ncclResult_t)" % (what, rc))


class RcclComm(object):
    """One RCCL communicator per process (= per GPU), created through the C ABI."""

    def __init__(self, rank, world, unique_id=None, dist=None):
        """unique_id: the 128 rendezvous bytes of rank 0 (bytes object).  When absent they are created on rank 0 and
        broadcast over `dist` (a torch.distributed-like module with an initialised default group); world == 1 needs neither."""
        L = lib()
        if unique_id is None:
            buf = (ctypes.c_char * ID_BYTES)()
            if rank == 0:
                _check(L.mmfn_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)), "mmfn_comm_unique_id")
            payload = [bytes(buf.raw)]
            if world > 1:
                if dist is None:
                    raise ValueError("world > 1 needs the rendezvous id or a process group to broadcast it over")
                dist.broadcast_object_list(payload, src=0)
            unique_id = payload[0]
        assert len(unique_id) == ID_BYTES
        self.rank, self.world = rank, world
        self._holders = weakref.WeakSet()   # objects owning hipGraphs that captured collectives of this communicator (retain())
        self._comm = ctypes.c_void_p()
        idbuf = ctypes.create_string_buffer(unique_id, ID_BYTES)
        _check(L.mmfn_comm_init(ctypes.byref(self._comm), ctypes.cast(idbuf, ctypes.c_void_p), world, rank), "mmfn_comm_init")

    def all_reduce_sum_(self, t, stream=None):
        """In-place sum over ranks of a contiguous fp32 or bf16 device tensor, enqueued on `stream` (default: the current one)."""
        assert t.dtype in (torch.float32, torch.bfloat16) and t.is_contiguous() and t.is_cuda
        st = (stream or torch.cuda.current_stream()).cuda_stream
        if t.dtype == torch.bfloat16:
            _check(lib().mmfn_allreduce_sum_bf16(self._comm, t.data_ptr(), t.numel(), st), "mmfn_allreduce_sum_bf16")
        else:
            _check(lib().mmfn_allreduce_sum_f32(self._comm, t.data_ptr(), t.numel(), st), "mmfn_allreduce_sum_f32")

    def broadcast_(self, t, root=0, stream=None):
        assert t.is_contiguous() and t.is_cuda
        st = (stream or torch.cuda.current_stream()).cuda_stream
        _check(lib().mmfn_broadcast_bytes(self._comm, t.data_ptr(), t.numel() * t.element_size(), root, st), "mmfn_broadcast_bytes")

    def ranks(self):
        n, r = ctypes.c_int(), ctypes.c_int()
        _check(lib().mmfn_comm_ranks(self._comm, ctypes.byref(n), ctypes.byref(r)), "mmfn_comm_ranks")
        return n.value, r.value

    def retain(self, holder):
        """Register an object that owns hipGraphs with captured collectives of this communicator (graphs.Recorder; weakly held)."""
        self._holders.add(holder)

    def destroy_unused(self):
        """Destroy a communicator NOTHING was ever captured or launched on (an init that came up after its time limit,
        open_transport): only the C-ABI call - no garbage collection and no drain of retired hipGraphs, which belong to the main
        thread (it may be capturing or training on torch.distributed by now; this runs on the abandoned helper thread)."""
        if self._comm:
            lib().mmfn_comm_destroy(self._comm)
            self._comm = ctypes.c_void_p()

    def destroy(self):
        """Destroy the communicator.  A hipGraph that captured its collectives must be gone first: RCCL hooks the graph's
        destruction and reaches into the communicator from there - freeing the communicator under a live (or merely retired,
        graphs._graveyard) capture corrupted the heap and crashed a LATER replay (round-4 full-suite segfault).  So the retired
        captures are destroyed here, and a live one is an error instead of a crash."""
        if not self._comm:
            return
        if torch.cuda.is_available() and torch.cuda.is_initialized() and not torch.cuda.is_current_stream_capturing():
            from . import graphs
            gc.collect()
            graphs.drain_graveyard()
        if len(self._holders):
            raise MMFNCommError("%d captured step(s) still hold collectives of this communicator: drop them (GraphedStep / the trainer's "
                                "captures) before RcclComm.destroy()" % len(self._holders))
        lib().mmfn_comm_destroy(self._comm)
        self._comm = ctypes.c_void_p()


def all_ranks_agree(dist, dev, ok):
    """Logical AND of `ok` over the ranks (through the launcher's process group)."""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def _real_communicator(rank, world, unique_id, dev):
    """ncclCommInitRank through the C ABI + the self-test: sum of (rank + 1) over a 1 MiB bucket on a side stream."""
    torch.cuda.set_device(dev)
    c = RcclComm(rank, world, unique_id=unique_id)
    try:
        x = torch.full((1 << 18,), float(rank + 1), dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        c.all_reduce_sum_(x, stream=side)
        side.synchronize()
        expect = world * (world + 1) / 2.0
        if float(x.min().item()) != expect or float(x.max().item()) != expect or c.ranks() != (world, rank):
            raise RuntimeError("self-test all-reduce returned %r..%r, expected %r" % (float(x.min()), float(x.max()), expect))
    except BaseException:
        c.destroy()
        raise
    return c


def _local_rendezvous_id(rank):
    """This rank's part of the rendezvous that needs no other rank: the library loads, rank 0 creates the 128 id bytes."""
    L = lib()
    buf = (ctypes.c_char * ID_BYTES)()
    if rank == 0:
        _check(L.mmfn_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)), "mmfn_comm_unique_id")
    return bytes(buf.raw)


def open_transport(rank, world, dist, dev, required=False, timeout_s=120.0, make_id=None, make_comm=None):
    """RCCL communicator through the C ABI + a self-test all-reduce; every rank must pass, else all of them use
    torch.distributed.  Returns (RcclComm or None, note); required=True raises MMFNCommError instead of falling back.

    Collective-safe by construction - every collective on the launcher's process group `dist` is issued from THIS thread, in
    the same order on every rank, whatever fails where:
      1. local readiness (library loads, rank 0 has its id) -> all_ranks_agree; a rank without the library never leaves the
         others blocked in the id broadcast,
      2. the id broadcast (only when 1. passed everywhere),
      3. ncclCommInitRank + self-test in a helper thread that touches RCCL only, never `dist`, bounded by `timeout_s`,
      4. all_ranks_agree on the outcome.
    An init that is still running at the time limit (asymmetric failure: one rank's init errors at once, the others wait for it
    in RCCL's bootstrap) is ABANDONED: this rank votes no, and if the helper ever returns, it destroys its communicator itself.

    make_id(rank) -> 128 bytes and make_comm(rank, world, id, dev) -> communicator (with .destroy()) replace the C-ABI calls in
    the CPU tests of this protocol."""
    import sys
    import threading
    make_id = make_id or _local_rendezvous_id
    make_comm = make_comm or _real_communicator
    # under the nccl backend broadcast_object_list / all_reduce stage through torch.cuda.current_device(): a caller that never set
    # its device (trainer.fit -> connect) would have every rank use cuda:0 ("Duplicate GPU detected" or a hang)
    if getattr(dev, "type", None) == "cuda" and torch.cuda.is_available():
        torch.cuda.set_device(dev)

    def give_up(why):
        if required:
            raise MMFNCommError("C-ABI transport required but unavailable on rank %d: %s" % (rank, why))
        if rank == 0:
            sys.stderr.write("C-ABI RCCL transport not used (%s); falling back to torch.distributed\n" % why)
        return None, "fallback: " + why

    # 1. what this rank can establish alone
    local_err, my_id = None, None
    try:
        my_id = make_id(rank)
        assert len(my_id) == ID_BYTES
    except BaseException as exc:   # noqa: BLE001 - a missing library, an RCCL error code: all mean "use torch.distributed"
        local_err = "%s: %s" % (type(exc).__name__, exc)
    if not all_ranks_agree(dist, dev, local_err is None):
        return give_up(local_err or "another rank cannot load libmmfn_comm.so / create the rendezvous id")
    # 2. rank 0's id to everybody (main thread, default group)
    payload = [my_id]
    if world > 1:
        dist.broadcast_object_list(payload, src=0)
    unique_id = payload[0]

    # 3. communicator + self-test, timed
    box, lock = {}, threading.Lock()

    def attempt():
        try:
            c = make_comm(rank, world, unique_id, dev)
        except BaseException as exc:   # noqa: BLE001
            with lock:
                box["error"] = "%s: %s" % (type(exc).__name__, exc)
            return
        with lock:
            if box.get("abandoned"):   # the time limit passed and this rank already voted no: nobody will use this communicator
                late = c
            else:
                box["comm"], late = c, None
        if late is not None:
            try:
                getattr(late, "destroy_unused", late.destroy)()   # (never used: no captured graph can refer to it)
            except BaseException:   # noqa: BLE001
                pass

    th = threading.Thread(target=attempt, daemon=True, name="mmfn-comm-init")
    th.start()
    th.join(timeout_s)
    with lock:
        if "comm" not in box and "error" not in box:
            box["abandoned"] = True
            box["error"] = "communicator did not come up within %.0f s" % timeout_s
        handle = box.get("comm")
    # 4. the verdict
    if not all_ranks_agree(dist, dev, handle is not None):
        if handle is not None:   # came up here, not elsewhere
            try:
                handle.destroy()
            except BaseException:   # noqa: BLE001
                pass
        return give_up(box.get("error", "another rank failed"))
    return handle, None
