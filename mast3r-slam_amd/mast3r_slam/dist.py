"""Multi-GPU sharding of keyframe pairs / graph edges: one process per GPU, RCCL over xGMI.

The reference has no multi-device code; this is new design.  Pairs (and backend
edges) are independent units, so the data path needs NO collective: rank r simply owns pairs
[r*P, (r+1)*P).  The only exchange is the result all-gather, issued as ONE collective per step
on a single packed byte buffer (pointmaps + confidences + indices + validity + poses): xGMI is
point-to-point (7 links per GPU), so one large all-gather keeps every link busy, whereas many
small ones would each pay the launch + ring latency.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> range:
    """Static block partition of `total` units; the first (total % world) ranks get one extra."""
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return range(lo, lo + q + (1 if rank < r else 0))


def _layout(meta):
    """Byte offset and length of every field of a packed buffer (fields start on 16-byte boundaries) + the total."""
    spans, off = [], 0
    for shape, dtype in meta:
        nbytes = torch.empty(0, dtype=dtype).element_size()
        for s in shape:
            nbytes *= int(s)
        spans.append((off, nbytes))
        off += nbytes + ((-nbytes) % 16)
    return spans, off


def pack(tensors):
    """Flatten a tuple of tensors into one uint8 buffer (+ the metadata to undo it)."""
    meta = [(tuple(t.shape), t.dtype) for t in tensors]
    spans, total = _layout(meta)
    buf = torch.zeros(total, dtype=torch.uint8, device=tensors[0].device)
    for (off, nbytes), t in zip(spans, tensors):
        buf[off:off + nbytes].view(t.dtype).view(t.shape).copy_(t)
    return buf, meta


def unpack(buf, meta, world, fold: bool = True):
    """buf uint8 [world, nbytes] -> tuple of tensors.
    fold=False: STRIDED VIEWS of `buf`, shape [world, *shape] - no byte moves (a field of rank r sits at
    buf[r, off:off+n]; 16-byte field offsets keep every dtype aligned).
    fold=True: the rank axis folded into dim 0, [world * shape[0], ...] - still a view at world == 1; for
    world > 1 the ranks' rows are not adjacent in the packed buffer, so this form copies (cold paths only:
    the per-step exchange uses PackedGather, whose views never copy)."""
    spans, _ = _layout(meta)
    out = []
    for (off, nbytes), (shape, dtype) in zip(spans, meta):
        v = buf[:, off:off + nbytes].view(dtype).view((world,) + tuple(shape))
        if fold:
            v = v.reshape((world * shape[0],) + tuple(shape[1:])) if len(shape) else v.reshape(world)
        out.append(v)
    return tuple(out)


class PackedGather:
    """The per-step result exchange with every buffer allocated ONCE: a packed send buffer (16-byte aligned fields),
    a [world, nbytes] receive buffer and per-field strided views of both.  post() snapshots the step's result tensors
    into the send buffer (one copy kernel per field on the caller's stream; dtype conversions such as int64 -> int32
    indices happen inside that copy) and issues ONE asynchronous all-gather; wait() makes the caller's stream wait
    for it and returns the receive VIEWS, shape [world, *field shape] - nothing is re-allocated or copied per step.
    At most one gather is in flight; post() waits for the previous one first (stream-ordered, no host sync), which
    also orders the new collective behind every read of the previous step's views on the caller's stream."""

    def __init__(self, like, dtypes=None, group=None):
        self.group, self.world = group, dist.get_world_size(group)
        dtypes = list(dtypes) if dtypes is not None else [t.dtype for t in like]
        self.meta = [(tuple(t.shape), dt) for t, dt in zip(like, dtypes)]
        spans, total = _layout(self.meta)
        self.spans = spans
        dev = like[0].device
        self.nccl = dist.get_backend(group) == "nccl"
        self.device = dev
        self.send = torch.zeros(total, dtype=torch.uint8, device=dev)
        self.send_views = [self.send[off:off + n].view(dt).view(shape) for (off, n), (shape, dt) in zip(spans, self.meta)]
        if self.nccl:
            self.recv = torch.empty((self.world, total), dtype=torch.uint8, device=dev)
        else:                                                   # gloo (CPU tests; ranks sharing one GPU in tests): through the host
            self.recv = torch.empty((self.world, total), dtype=torch.uint8)
            self.hsend = torch.empty(total, dtype=torch.uint8)
            self.parts = list(self.recv.unbind(0))              # row views of the receive buffer: gathered in place
            self.dev_recv = torch.empty((self.world, total), dtype=torch.uint8, device=dev) if dev.type != "cpu" else None
        self.views = unpack(self.recv if self.nccl or self.dev_recv is None else self.dev_recv, self.meta, self.world, fold=False)
        self.nbytes = total
        self.work = None

    def _snapshot(self, tensors):
        """tensors -> the packed send buffer.  On the ROCm device: ONE launch of the library's pack kernel (m3_pack_fields: 16 B
        per lane, int64 -> int32 narrowing for the index field); seven device-to-device copies through the runtime took
        0.6 ms of copy-kernel time per 8-pair step.  Elsewhere (CPU tensors: the gloo tests) plain copy_."""
        if self.device.type != "cuda":
            for v, t in zip(self.send_views, tensors):
                v.copy_(t)
            return
        import ctypes as C
        from . import _ffi
        segs = (_ffi.PackSeg * len(tensors))()
        keep = []
        for i, ((off, nbytes), (shape, dt), t) in enumerate(zip(self.spans, self.meta, tensors)):
            if tuple(t.shape) != shape or not t.is_cuda:
                raise ValueError(f"field {i}: expected a device tensor of shape {shape}, got {tuple(t.shape)} on {t.device}")
            t = t.contiguous()
            narrow = t.dtype == torch.int64 and dt == torch.int32
            if not narrow and t.dtype != dt:
                t = t.to(dt)                                    # rare: any other conversion goes through torch
            keep.append(t)
            segs[i].src, segs[i].dst_off, segs[i].nbytes, segs[i].mode = t.data_ptr(), off, nbytes, 1 if narrow else 0
        _ffi.call("m3_pack_fields", self.send.data_ptr(), C.addressof(segs), len(tensors), _ffi.stream_ptr())

    def post(self, tensors):
        self.wait()
        self._snapshot(tensors)
        if self.nccl:
            self.work = dist.all_gather_into_tensor(self.recv.view(-1), self.send, group=self.group, async_op=True)
        else:
            self.hsend.copy_(self.send)
            self.work = dist.all_gather(self.parts, self.hsend, group=self.group, async_op=True)
        return self

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
            if not self.nccl and self.dev_recv is not None:
                self.dev_recv.copy_(self.recv)
        return self.views


class GatherHandle:
    """An all-gather in flight (all_gather_results(..., async_op=True)).  wait() makes the CURRENT stream
    wait for the collective (no host sync) and returns the gathered tuple."""

    def __init__(self, work, out, parts, meta, world, device=None):
        self._work, self._out, self._parts, self._meta, self._world, self._device = work, out, parts, meta, world, device

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        out = self._out if self._parts is None else torch.stack(self._parts)
        if self._device is not None:
            out = out.to(self._device)
        return unpack(out, self._meta, self._world)


def all_gather_results(tensors, group=None, async_op: bool = False):
    """All ranks end up with every rank's per-pair results concatenated along dim 0 (rank order).
    One collective for the whole tuple.  Every rank must pass identically shaped tensors.
    async_op=True returns a GatherHandle: the collective runs on the communicator's stream while the
    caller's stream goes on with the next batch (the packed send buffer is a private copy, so the inputs
    may be overwritten immediately)."""
    world = dist.get_world_size(group)
    buf, meta = pack(tensors)
    if dist.get_backend(group) == "nccl":
        out = torch.empty((world, buf.numel()), dtype=torch.uint8, device=buf.device)
        work = dist.all_gather_into_tensor(out.view(-1), buf, group=group, async_op=async_op)
        handle = GatherHandle(work if async_op else None, out, None, meta, world)
    else:                                                       # gloo: CPU tests, and ranks sharing one GPU in tests (gloo
        dev = buf.device                                        # has no device all_gather: the packed buffer goes through the host)
        hbuf = buf.cpu() if buf.is_cuda else buf
        parts = [torch.empty_like(hbuf) for _ in range(world)]
        work = dist.all_gather(parts, hbuf, group=group, async_op=async_op)
        handle = GatherHandle(work if async_op else None, None, parts, meta, world, device=dev if buf.is_cuda else None)
    return handle if async_op else handle.wait()


def all_gather_rows(local: torch.Tensor, total: int, group=None, sizes=None) -> torch.Tensor:
    """Ragged all-gather along dim 0: rank r passes its sizes[r] rows, every rank gets all `total` rows in rank
    order.  sizes=None: the shard_range() partition of `total`.  One collective (shards are padded to the largest
    shard; the padding is dropped on arrival)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if sizes is None:
        sizes = [len(shard_range(total, r, world)) for r in range(world)]
    sizes = [int(s) for s in sizes]
    if len(sizes) != world or sum(sizes) != total:
        raise ValueError(f"sizes {sizes} do not describe {total} rows over {world} ranks")
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank} owns {sizes[rank]} rows, got {local.shape[0]}")
    cap = max(sizes) if sizes else 0
    buf = local.new_zeros((cap,) + tuple(local.shape[1:]))
    buf[: sizes[rank]] = local
    gathered = all_gather_results((buf[None],), group)[0]              # [world, cap, ...]
    return torch.cat([gathered[r, : sizes[r]] for r in range(world)])


def sharded_edge_blocks(block_fn, num_edges: int, group=None) -> torch.Tensor:
    """Backend GN blocks with the graph edges split over the ranks: rank r evaluates
    block_fn(edge_range) -> [len(range), 36] float64 for its own edges only (the heavy, per-point part:
    10.8 MB of reads per edge), then the 36-double blocks are all-gathered (288 B per edge) so that every
    rank can assemble and solve the same dense system.  Edges are independent: no other collective."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = shard_range(num_edges, rank, world)
    return all_gather_rows(block_fn(mine), num_edges, group)
