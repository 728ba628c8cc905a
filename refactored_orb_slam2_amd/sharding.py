"""Batched-sequence mode: independent frames are sharded over ranks (one process per GPU) and the per-frame
keypoint records are gathered with one collective per tensor (RCCL on GPUs, gloo in the CPU tests): `mode="all"` is an
all_gather (every rank ends up with every frame's records), `mode="root"` a gather to rank 0 (the consumer of
the records is one process; N - 1 sends of one rank's records instead of N x (N - 1)).

The reference processes frames strictly sequentially in one process (Source/Examples/Stereo/stereo_kitti.cc:
36-150); extraction carries no state between frames, so contiguous chunks of the frame range are independent
units.  Contiguous rather than round-robin keeps consecutive-frame matching rank-local.
"""
from __future__ import annotations


def shard_range(n_frames: int, rank: int, world: int) -> tuple[int, int]:
    """[begin, end) of the frames rank `rank` owns: contiguous chunks, sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world) or n_frames < 0:
        raise ValueError("bad shard arguments")
    base, rem = divmod(n_frames, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_range_c(n_frames: int, rank: int, world: int) -> tuple[int, int]:
    """The same split through the C ABI (orbfe_shard_range): what a C++ host uses."""
    import ctypes as C
    from . import _lib
    b, e = C.c_int(0), C.c_int(0)
    _lib.check(_lib.lib().orbfe_shard_range(n_frames, rank, world, C.byref(b), C.byref(e)), "orbfe_shard_range")
    return b.value, e.value


class RcclGather:
    """ctypes mirror of the C ABI's record gather (include/orbfe.h: orbfe_gather_*): RCCL called by liborbfe itself, no
    torch.distributed in the data path.  `unique_id()` on rank 0, hand the 128 bytes to the other ranks (any side channel:
    here usually a torch.distributed broadcast of the control-plane group), then RcclGather(id, rank, world, device)."""
    ALL, ROOT = 0, 1

    def __init__(self, uid: bytes, rank: int, world: int, device: int = -1):
        import ctypes as C
        from . import _lib
        self._L = _lib.lib()
        self._h = C.c_void_p(None)
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        _lib.check(self._L.orbfe_gather_create(buf, rank, world, device, C.byref(self._h)), "orbfe_gather_create")
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import _lib
        buf = (C.c_uint8 * 128)()
        _lib.check(_lib.lib().orbfe_gather_unique_id(buf), "orbfe_gather_unique_id")
        return bytes(buf)

    def gather(self, n, kps, desc, mode: str = "all", stream=None):
        """n: int32 (F,), kps: uint8 (F, cap, 28), desc: uint8 (F, cap, 32) device tensors.  Returns the gathered tensors
        (world * F leading dimension) on the ranks that receive, None elsewhere.  Asynchronous on `stream`."""
        import torch
        from . import _lib
        if mode not in ("all", "root"):
            raise ValueError("mode must be 'all' or 'root'")
        recv = mode == "all" or self.rank == 0
        out = [torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in (n, kps, desc)] if recv else None
        ptrs = [_lib.ptr(t) for t in out] if recv else [None, None, None]
        # Ordering against torch: the collective runs on `stream`, or -- stream None -- on the handle's own non-blocking stream, which
        # nothing orders behind the torch streams that produced n / kps / desc.  An explicit stream is the caller's to order (the
        # tensors are marked as used on it, so the caching allocator does not hand them out while the collective is in flight); with
        # the private stream the producers are waited for here and inputs and outputs are kept alive until sync().
        if stream is None:
            torch.cuda.current_stream(n.device).synchronize()
            self._inflight = (n, kps, desc, out)
        else:
            for t in (n, kps, desc) + tuple(out or ()):
                t.record_stream(stream)
        _lib.check(self._L.orbfe_gather_records(self._h, _lib.ptr(n), _lib.ptr(kps), _lib.ptr(desc), n.shape[0], kps.shape[1],
                                                self.ALL if mode == "all" else self.ROOT, ptrs[0], ptrs[1], ptrs[2],
                                                _lib.stream_handle(stream)), "orbfe_gather_records")
        return tuple(out) if recv else None

    def sync(self):
        """Waits for the collectives issued on the handle's own stream (gather(..., stream=None)); call it before reading them."""
        from . import _lib
        _lib.check(self._L.orbfe_gather_sync(self._h), "orbfe_gather_sync")
        self._inflight = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.orbfe_gather_destroy(self._h)
            self._h.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def padded_chunk(n_frames: int, world: int) -> int:
    """Frames per rank after padding the last chunks, so every rank contributes equally sized records."""
    return (n_frames + world - 1) // world


def record_bytes(n, kps, desc) -> int:
    """Bytes of one rank's padded records {n[F]; kps[F, cap, 28]; desc[F, cap, 32]}."""
    return sum(int(t.numel()) * t.element_size() for t in (n, kps, desc))


def gather_traffic(n, kps, desc, world: int, mode: str = "all") -> dict:
    """Bytes a rank sends / receives per gather (payload, not the ring's forwarding): the figures bench.py prints."""
    b = record_bytes(n, kps, desc)
    if mode == "root":
        return {"mode": "root", "record_bytes_per_rank": b, "sent_per_rank": b, "received_rank0": (world - 1) * b, "received_other_ranks": 0}
    return {"mode": "all", "record_bytes_per_rank": b, "sent_per_rank": b, "received_per_rank": (world - 1) * b}


class AsyncGather:
    """Overlaps the record gather of step k with the compute of step k+1: the step's outputs are snapshotted
    (device-to-device copy on the compute stream), the collectives run asynchronously on the process group's
    stream, and the previous gather is waited for only when its buffers are about to be reused.
    mode "all": all_gather_into_tensor; mode "root": gather to rank 0 (the other ranks hold no output)."""

    def __init__(self, n, kps, desc, group=None, mode: str = "all"):
        import torch
        import torch.distributed as dist
        if mode not in ("all", "root"):
            raise ValueError("mode must be 'all' or 'root'")
        self.group = group
        self.mode = mode
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # RCCL moves device tensors; the gloo rehearsal path (CPU tests, several ranks on one GPU -- RCCL refuses two ranks
        # per device) stages the records through host memory
        self.on_host = dist.get_backend(group) != "nccl" and n.is_cuda
        dev = "cpu" if self.on_host else n.device
        self.snap = [torch.empty_like(t, device=dev) for t in (n, kps, desc)]
        self.out = None
        if mode == "all" or self.rank == 0:
            self.out = [torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
                        for t in (n, kps, desc)]
        self.pending = []

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []

    def launch(self, n, kps, desc):
        import torch
        import torch.distributed as dist
        self.wait()  # the snapshot / output buffers are free again
        for s, t in zip(self.snap, (n, kps, desc)):
            s.copy_(t, non_blocking=not self.on_host)
        if self.on_host:
            torch.cuda.current_stream().synchronize()
        if self.mode == "all":
            self.pending = [dist.all_gather_into_tensor(o, s, group=self.group, async_op=True)
                            for o, s in zip(self.out, self.snap)]
        else:
            dst = dist.get_global_rank(self.group, 0) if self.group is not None else 0
            self.pending = [dist.gather(s, list(o.chunk(self.world)) if self.rank == 0 else None, dst=dst, group=self.group, async_op=True)
                            for o, s in zip(self.out or [None] * 3, self.snap)]

    def result(self):
        """(n_all, kps_all, desc_all) in rank order; None on the ranks that hold no output (mode "root", rank > 0)."""
        self.wait()
        return tuple(self.out) if self.out is not None else None


class CabiAsyncGather:
    """AsyncGather's interface over the C ABI's gather (RcclGather: RCCL called by liborbfe itself -- what a C++ host of the batched
    mode uses): the step's records are snapshotted on the compute stream, the collective runs on a side stream behind it and overlaps
    the next step; launch() / wait() / result() as AsyncGather.  The communicator's unique id travels over the torch.distributed
    control-plane group when world > 1 (any side channel would do); world 1 is a valid communicator (self-check on one GPU)."""

    def __init__(self, n, kps, desc, rank: int, world: int, device: int, mode: str = "all", group=None):
        import torch
        import torch.distributed as dist
        if mode not in ("all", "root"):
            raise ValueError("mode must be 'all' or 'root'")
        uid = [RcclGather.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(uid, src=0, group=group)
        self.g = RcclGather(uid[0], rank, world, device)
        self.mode, self.rank, self.world = mode, rank, world
        self.snap = [torch.empty_like(t) for t in (n, kps, desc)]
        self.stream = torch.cuda.Stream(n.device)
        self.ev = torch.cuda.Event()
        self.out = None
        self.pending = False

    def wait(self):
        """Orders the current stream behind the collective in flight (its snapshot / output buffers are free again after it)."""
        import torch
        if self.pending:
            torch.cuda.current_stream().wait_event(self.ev)
            self.pending = False

    def launch(self, n, kps, desc):
        import torch
        cur = torch.cuda.current_stream()
        self.wait()
        for s, t in zip(self.snap, (n, kps, desc)):
            s.copy_(t, non_blocking=True)
        self.stream.wait_stream(cur)
        self.out = self.g.gather(self.snap[0], self.snap[1], self.snap[2], mode=self.mode, stream=self.stream)
        self.ev.record(self.stream)
        self.pending = True

    def result(self):
        self.ev.synchronize()
        self.pending = False
        return self.out

    def close(self):
        self.g.close()


def gather_records(n, kps, desc, group=None, mode: str = "all"):
    """Gather of fixed-size padded per-frame records {n[f]; kps[f, cap, 28]; desc[f, cap, 32]}.

    n: int32 (F,), kps: uint8 (F, cap, 28), desc: uint8 (F, cap, 32) -- F identical on every rank.
    Returns (n_all (world*F,), kps_all (world*F, cap, 28), desc_all (world*F, cap, 32)) in rank order, i.e.
    in global frame order for contiguous shards; with mode "root" only rank 0 gets them, the others None.
    With world size 1 (or no process group) it is the identity.
    """
    import torch
    import torch.distributed as dist

    if mode not in ("all", "root"):
        raise ValueError("mode must be 'all' or 'root'")
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return n, kps, desc
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    on_host = dist.get_backend(group) != "nccl" and n.is_cuda   # gloo rehearsal: records go through host memory
    out = []
    for t in (n, kps, desc):
        t = t.contiguous().cpu() if on_host else t.contiguous()
        if mode == "all":
            g = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(g, t, group=group)  # concatenation along dim 0, rank order
        else:
            g = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) if rank == 0 else None
            dst = dist.get_global_rank(group, 0) if group is not None else 0
            dist.gather(t, list(g.chunk(world)) if rank == 0 else None, dst=dst, group=group)
        out.append(g)
    return tuple(out) if (mode == "all" or rank == 0) else None


def unpack_records(n_all, kps_all, desc_all, n_frames: int, world: int = 1):
    """Drops the padding frames and returns per-frame (keypoints, descriptors) numpy arrays in global frame order.

    The gathered arrays hold `world` chunks of equal length (padded_chunk(n_frames, world) frames each, rank order); rank r's
    chunk starts at row r * chunk and holds its shard_range(n_frames, r, world) frames followed by padding, so with
    n_frames % world != 0 the rows behind the first short chunk are NOT at their global frame number."""
    import numpy as np
    from ._lib import KP_DTYPE

    n_np = n_all.cpu().numpy()
    k_np = kps_all.cpu().numpy()
    d_np = desc_all.cpu().numpy()
    if world < 1 or n_np.shape[0] % world:
        raise ValueError("the gathered arrays are not `world` equal chunks")
    chunk = n_np.shape[0] // world
    res = []
    for r in range(world):
        b, e = shard_range(n_frames, r, world)
        if e - b > chunk:
            raise ValueError("chunk shorter than the rank's shard")
        for f in range(b, e):
            row = r * chunk + (f - b)
            c = int(n_np[row])
            res.append((k_np[row, :c].copy().view(KP_DTYPE).reshape(-1), d_np[row, :c].copy()))
    return res
