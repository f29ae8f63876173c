"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm) for the single
exchange step of the path — the all-reduce of the flat fp32 gradient bucket — and tile sharding for inference."""
import numpy as np


class _DevicePointer:
    """Zero-copy view of library-owned HBM for torch (CUDA array interface v2)."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (ptr, False), "version": 2, "strides": None}


def device_tensor(ptr, count, typestr="<f4", device=None):
    import torch
    return torch.as_tensor(_DevicePointer(ptr, count, typestr), device=device or torch.device("cuda", torch.cuda.current_device()))


def grad_bucket_tensor(trainer):
    """The trainer's flat gradient bucket (n_params fp32 + 1 trailing slot = the loss) as a torch tensor."""
    ptr, n = trainer.grad_buffer()
    return device_tensor(ptr, n)


def handle_stream(handle):
    """The handle's HIP stream as a torch stream: torch work issued under `with torch.cuda.stream(handle_stream(h))` is
    ordered with the handle's own kernels (torch.distributed orders a collective against the CURRENT stream: the
    all-reduce then starts after backward and the update starts after the all-reduce)."""
    import torch
    s = torch.cuda.ExternalStream(handle.stream_ptr(), device=torch.device("cuda", torch.cuda.current_device()))
    s._anh_handle = handle   # the wrapper does not own the stream (torch: "the user keeps it alive"): it keeps the OWNER alive instead
    return s


class EarlyReduce:
    """The all-reduce of the gradient bucket split in two so that most of it overlaps the end of backward: the filter gradients
    come out last layer first, so the bucket's tail [first, n + 1) — every layer but the first two, the head, the loss slot — is
    final while the last backward-data convs and the first layers' filter gradients still run (anh_trainer_early_grads).  That
    part is reduced on a side stream gated by the library's event; the short head [0, first) follows on the trainer's stream."""

    def __init__(self, trainer, bucket):
        import torch
        self.first = trainer.early_grads() if bucket.is_cuda else bucket.numel()
        self.split = 0 < self.first < bucket.numel() - 1
        self.side = torch.cuda.Stream(device=bucket.device) if self.split else None

    def run(self, trainer, bucket, main_stream, group=None, events=None):
        """events (a list, optional): ("tail" | "head" | "whole", start, stop) torch events around each collective are appended —
        bench.py's `exchange` object; an event pair costs stream time, so only on sampled, untimed steps."""
        import torch
        import torch.distributed as dist
        if not self.split:
            with torch.cuda.stream(main_stream):
                _timed_all_reduce(dist, torch, bucket, group, events, "whole")
            return
        trainer.wait_early_grads(self.side.cuda_stream)
        with torch.cuda.stream(self.side):
            _timed_all_reduce(dist, torch, bucket[self.first:], group, events, "tail")
        with torch.cuda.stream(main_stream):
            _timed_all_reduce(dist, torch, bucket[:self.first], group, events, "head")
        main_stream.wait_stream(self.side)      # the update reads the whole bucket


def _timed_all_reduce(dist, torch, tensor, group, events, name):
    if events is None:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()          # on the current stream (the caller's `with torch.cuda.stream(...)`)
    dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
    b.record()
    events.append((name, a, b))


def data_parallel_step(trainer, bucket, d_images, d_labels, d_weights, n, h, w, world_size, group=None, force_collective=False, stream=None, early=None, events=None):
    """One optimiser step of a data-parallel job.  The loss scale uses the GLOBAL batch (n * world_size), so the
    all-reduce is a plain SUM and every rank then applies the identical update.  The collective is
    enqueued on the trainer's own stream (`stream` = handle_stream(trainer), made once by the caller): backward ->
    all-reduce -> SGD is one stream-ordered chain with no host synchronisation.  With `early` (an EarlyReduce made once by the
    caller) the bulk of the bucket is reduced while backward is still running."""
    trainer.forward_backward_device(d_images, d_labels, d_weights, n, h, w, n * world_size)
    if world_size > 1 or force_collective:
        import torch
        import torch.distributed as dist
        if bucket.is_cuda:
            main = stream if stream is not None else handle_stream(trainer)
            if early is not None:
                early.run(trainer, bucket, main, group, events=events)
            else:
                with torch.cuda.stream(main):
                    _timed_all_reduce(dist, torch, bucket, group, events, "whole")
        else:   # CPU rehearsal of the same logic over gloo (tests/test_dist_gloo.py)
            dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
    trainer.apply_update(1.0)


def shard_tiles(tiles, rank, world_size):
    """Tile sharding for multi-GPU inference: contiguous row-major chunks, so a rank's tiles are spatial neighbours."""
    n = len(tiles)
    lo = (n * rank) // world_size
    hi = (n * (rank + 1)) // world_size
    return tiles[lo:hi]


def tile_owner(n_tiles, world_size):
    """rank of every tile under shard_tiles' contiguous chunks"""
    owner = [0] * n_tiles
    for r in range(world_size):
        for i in range((n_tiles * r) // world_size, (n_tiles * (r + 1)) // world_size):
            owner[i] = r
    return owner


def cross_rank_overlaps(tiles, world_size, width, height):
    """The rectangles (left, top, right, bottom; inclusive, clipped to the image) in which tiles of DIFFERENT ranks overlap:
    the only places where a rank's blended planes hold partial sums (annonet_infer.cpp:116-164 adds the ramps of every tile
    covering a pixel).  The same sorted list on every rank.  Overlaps are a few tens of pixels wide (overlap = receptive
    field), so this is a few percent of the image."""
    owner = tile_owner(len(tiles), world_size)
    rects = set()
    for i, (fi, _) in enumerate(tiles):
        for j in range(i + 1, len(tiles)):
            if owner[i] == owner[j]:
                continue
            fj = tiles[j][0]
            l, t = max(fi[0], fj[0], 0), max(fi[1], fj[1], 0)
            r, b = min(fi[2], fj[2], width - 1), min(fi[3], fj[3], height - 1)
            if l <= r and t <= b:
                rects.add((l, t, r, b))
    return sorted(rects)


class OverlapExchange:
    """Gather / scatter of the cross-rank overlap pixels of blended planes [K, H, W] (torch, on the GPU): ONE index tensor
    of the unique pixel positions, built once per (tile list, world size), so an exchange is one gather, one all-reduce of
    a [K, n] buffer and one scatter — not a launch per rectangle."""

    def __init__(self, tiles, world_size, width, height, device):
        import torch
        self.rects = cross_rank_overlaps(tiles, world_size, width, height) if world_size > 1 else []
        if self.rects:
            parts = [(np.arange(t, b + 1, dtype=np.int64)[:, None] * width + np.arange(l, r + 1, dtype=np.int64)[None, :]).ravel() for (l, t, r, b) in self.rects]
            self.index = torch.from_numpy(np.unique(np.concatenate(parts))).to(device)
        else:
            self.index = None

    def pixels(self):
        return 0 if self.index is None else int(self.index.numel())

    def pack(self, blended):
        return blended.reshape(blended.shape[0], -1).index_select(1, self.index)

    def unpack(self, blended, packed):
        blended.view(blended.shape[0], -1).index_copy_(1, self.index, packed)   # unique positions: deterministic

    def run(self, blended, group=None):
        if self.index is None:
            return
        import torch.distributed as dist
        packed = self.pack(blended)
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        self.unpack(blended, packed)


class LabelGather:
    """ONE label map on one rank from the ranks' shares of a sharded annonet_infer().  After the overlap exchange a rank's labels
    are right exactly where ITS tiles cover the image (elsewhere its planes hold nothing); two ranks that share a pixel hold the
    same sums there, hence the same label.  Every rank codes its covered pixels as label + 1 (0 = "not mine", 255 = the all-NaN
    label 65535) and sends the ROWS its tiles touch — one byte per pixel, point to point — to `dst`, which merges each band with an
    elementwise maximum.  `dst` receives H x W bytes (plus the overlap rows) over its direct xGMI links to the other ranks; a
    reduce(MAX) of full-size maps would move H x W bytes per rank around the ring."""

    def __init__(self, tiles, world_size, rank, width, height, device, classes):
        import torch
        if classes > 253:
            raise ValueError("LabelGather codes labels in one byte: at most 253 classes")
        self.rank, self.world, self.width, self.height = rank, world_size, width, height
        self.rows = []                                    # [row0, row1) of every rank's tiles (the same list on every rank)
        for r in range(world_size):
            mine = shard_tiles(tiles, r, world_size)
            if mine:
                self.rows.append((max(0, min(t[0][1] for t in mine)), min(height, max(t[0][3] for t in mine) + 1)))
            else:
                self.rows.append((0, 0))
        row0, row1 = self.rows[rank]
        mask = np.zeros((max(row1 - row0, 0), width), dtype=bool)
        for (full, _) in shard_tiles(tiles, rank, world_size):
            l, t, r, b = max(full[0], 0), max(full[1], 0), min(full[2], width - 1), min(full[3], height - 1)
            if l <= r and t <= b:
                mask[t - row0:b + 1 - row0, l:r + 1] = True
        self.mask = torch.from_numpy(mask).to(device)     # which pixels of this rank's row band its tiles cover

    def code(self, labels):
        """this rank's row band of `labels` ([H, W] int16) as bytes: label + 1 where its tiles cover the pixel, else 0"""
        import torch
        row0, row1 = self.rows[self.rank]
        wide = labels[row0:row1].to(torch.int32) & 0xFFFF
        coded = torch.where(wide == 65535, torch.full_like(wide, 255), wide + 1)
        return torch.where(self.mask, coded, torch.zeros_like(coded)).to(torch.uint8).contiguous()

    @staticmethod
    def decode(coded):
        import torch
        wide = coded.to(torch.int32)
        return torch.where(wide == 255, torch.full_like(wide, 65535), wide - 1).to(torch.int16)   # (u16 bit pattern in an int16 tensor)

    def merge(self, bands):
        """bands[r] = rank r's coded row band (or None for a rank without tiles) -> the coded [H, W] map"""
        import torch
        first = next(b for b in bands if b is not None)
        whole = torch.zeros((self.height, self.width), dtype=torch.uint8, device=first.device)
        for (row0, row1), band in zip(self.rows, bands):
            if band is not None and row1 > row0:
                torch.maximum(whole[row0:row1], band, out=whole[row0:row1])
        return whole

    def run(self, labels, group=None, dst=0):
        """labels: this rank's [H, W] int16 map -> the assembled map on rank `dst` (None elsewhere)"""
        import torch
        import torch.distributed as dist
        mine = self.code(labels)
        # gloo moves host memory only in its point-to-point calls: a rehearsal job with GPU tensors (bench.py --backend gloo) stages the
        # bands through the host; over RCCL the bands travel GPU to GPU
        via_host = mine.is_cuda and dist.get_backend(group) == "gloo"
        if self.rank != dst:
            if mine.numel():
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, mine.cpu() if via_host else mine, dst, group)]):
                    w.wait()
            return None
        bands, ops = [], []
        for r, (row0, row1) in enumerate(self.rows):
            if r == dst:
                bands.append(mine if mine.numel() else None)
            elif row1 > row0:
                buf = torch.empty((row1 - row0, self.width), dtype=torch.uint8, device="cpu" if via_host else mine.device)
                bands.append(buf)
                ops.append(dist.P2POp(dist.irecv, buf, r, group))
            else:
                bands.append(None)
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if via_host:
            bands = [b.to(mine.device) if b is not None and not b.is_cuda else b for b in bands]
        return self.decode(self.merge(bands))


def sharded_infer(net, image, labels, blended, tiles, rank, world_size, exchange, tiling_parameters=None, gains=None, group=None, stream=None):
    """annonet_infer() with the tile list sharded over the ranks of a job (image, labels [H, W] int16 and blended [K, H, W]
    float32 are torch tensors on this rank's GPU; `exchange` = OverlapExchange(tiles, world_size, W, H, device)).
      1. this rank's tiles are blended into its own planes;
      2. the ONE exchange step of the path: the plane sums at the pixels where tiles of different ranks overlap are
         all-reduced (nothing is exchanged at world size 1);
      3. find_label over the rows this rank's tiles cover.
    Returns that row range (row0, row1), the rank's share of the label map: every pixel of it that one of this rank's tiles
    covers now carries the complete sum, hence the same label on every rank that covers it."""
    import torch
    from . import netpimpl as nn
    height, width = int(blended.shape[1]), int(blended.shape[2])
    mine = shard_tiles(tiles, rank, world_size)
    nn.annonet_infer_device(net, image.data_ptr(), height, width, 0, blended.data_ptr(), gains=gains, tiling_parameters=tiling_parameters, tiles=mine)
    if exchange.index is not None:
        with torch.cuda.stream(stream if stream is not None else handle_stream(net)):   # gather / all-reduce / scatter on the net's own stream
            exchange.run(blended, group)
    if not mine:
        return 0, 0
    row0 = max(0, min(t[0][1] for t in mine))
    row1 = min(height, max(t[0][3] for t in mine) + 1)
    nn.argmax_device(net, blended.data_ptr(), height, width, row0, row1, labels.data_ptr(), gains=gains)
    return row0, row1


def reduce_shard_planes(blended_np_list):
    """Sum per-rank blended planes in rank order (fixed order => reproducible overlap sums)."""
    out = np.zeros_like(blended_np_list[0])
    for b in blended_np_list:
        out = out + b
    return out


def tile_cells(tiles, width, height):
    """One rectangle (left, top, right, bottom; inclusive) per tile such that the rectangles PARTITION the image and each lies inside its
    tile's full rectangle: per axis the boundary between two neighbouring tiles runs through the middle of their overlap.  After the
    overlap exchange both owners of an overlap pixel hold the same sums, hence the same label, so any partition assembles the map of the
    single-process annonet_infer(); this one gives every tile a compact block.  Needs the grid the tiler builds (make_tiles)."""
    def axis(spans, size):
        spans = sorted(set(spans))
        bounds = [0]
        for (s0, e0), (s1, e1) in zip(spans, spans[1:]):
            if not (s0 < s1 and e0 < e1 and s1 <= e0 + 1):
                raise ValueError("tile_cells: the tiles do not form a grid")
            bounds.append((e0 + s1 + 1) // 2)
        bounds.append(size)
        return {span: (bounds[i], bounds[i + 1] - 1) for i, span in enumerate(spans)}
    xs = axis([(t[0][0], t[0][2]) for t in tiles], width)
    ys = axis([(t[0][1], t[0][3]) for t in tiles], height)
    cells = []
    for full, _ in tiles:
        (l, r), (t, b) = xs[(full[0], full[2])], ys[(full[1], full[3])]
        cells.append((max(l, 0), max(t, 0), min(r, width - 1), min(b, height - 1)))
    if sum((c[2] - c[0] + 1) * (c[3] - c[1] + 1) for c in cells) != width * height:
        raise ValueError("tile_cells: the tiles do not form a grid")
    return cells


_SEGMENTS_CREATED_HERE = set()   # names of the shared-memory segments this process created (HostLabelMap)


class HostLabelMap:
    """ONE [H, W] u16 label map in HOST memory shared by every rank of a sharded annonet_infer() (POSIX shared memory; the reference's
    writer threads consume host label maps, annonet_infer_main.cpp:403-419).  Every rank copies the cells of ITS tiles (tile_cells)
    from its device-resident map straight into the shared map, over its own PCIe link: rank 0 copies only its share, nothing is
    gathered in HBM first (LabelGather stays for hosts that need the map on a device).  The mapping is page-locked per rank
    (anh_host_register), so the copies are asynchronous on the rank's stream; `deliver` returns at once, the caller's stream / event
    says when the rows have landed.  The creator (rank `owner`) passes name=None and publishes `.name`; the others attach by name."""

    def __init__(self, tiles, world_size, rank, width, height, name=None, pin=True):
        from multiprocessing import shared_memory
        import ctypes as C
        self.width, self.height, self.rank = width, height, rank
        nbytes = width * height * 2
        self.creator = name is None
        if self.creator:
            self.shm = shared_memory.SharedMemory(create=True, size=nbytes)
            _SEGMENTS_CREATED_HERE.add(self.shm.name)
        else:
            # An ATTACHING rank must not own the segment's lifetime: before Python 3.13 SharedMemory(name=...) registers it with this
            # process's resource tracker, which unlinks it (and warns of a leak) when whichever rank exits first — the creator unlinks.
            try:
                self.shm = shared_memory.SharedMemory(name=name, track=False)          # Python >= 3.13
            except TypeError:
                self.shm = shared_memory.SharedMemory(name=name)
                if name not in _SEGMENTS_CREATED_HERE:     # (several ranks rehearsed in ONE process share the creator's registration)
                    from multiprocessing import resource_tracker
                    resource_tracker.unregister(self.shm._name, "shared_memory")
        self.name = self.shm.name
        self.array = np.ndarray((height, width), dtype=np.uint16, buffer=self.shm.buf)
        self._cbuf = C.c_char.from_buffer(self.shm.buf)      # (an export of the mapping: dropped in close() before the mapping is)
        self.ptr = C.addressof(self._cbuf)
        owner = tile_owner(len(tiles), world_size)
        cells = tile_cells(tiles, width, height)
        mine = [c for c, o in zip(cells, owner) if o == rank]
        # cells of one tile row that reach from the left to the right border merge into one block of whole rows (a contiguous copy)
        rows = {}
        for c in mine:
            rows.setdefault((c[1], c[3]), []).append(c)
        self.rects = []
        for (t, b), cs in sorted(rows.items()):
            cs.sort()
            if cs[0][0] == 0 and cs[-1][2] == width - 1 and all(a[2] + 1 == b2[0] for a, b2 in zip(cs, cs[1:])):
                self.rects.append((0, t, width - 1, b))
            else:
                self.rects.extend(cs)
        self.bytes = sum((r[2] - r[0] + 1) * (r[3] - r[1] + 1) for r in self.rects) * 2
        self.pinned = False
        if pin:
            from . import _lib
            self.pinned = _lib.lib().anh_host_register(self.ptr, nbytes) == 0

    def deliver(self, d_labels_ptr, stream_ptr):
        """enqueue the copies of this rank's rectangles on `stream_ptr` (a HIP stream handle); d_labels_ptr: the rank's [H, W] u16 map in HBM"""
        from . import _lib
        L = _lib.lib()
        for (l, t, r, b) in self.rects:
            _lib.check(L.anh_labels_rect_to_host(d_labels_ptr, self.ptr, self.width, self.height, l, t, r, b, stream_ptr))

    def deliver_from_host(self, labels):
        """the same rectangles from a host array (CPU rehearsal of the ownership logic: tests/test_dist_gloo.py)"""
        src = np.asarray(labels).view(np.uint16).reshape(self.height, self.width)
        for (l, t, r, b) in self.rects:
            self.array[t:b + 1, l:r + 1] = src[t:b + 1, l:r + 1]

    def close(self):
        if self.shm is None:
            return
        if self.pinned:
            from . import _lib
            _lib.lib().anh_host_unregister(self.ptr)
            self.pinned = False
        self.array = None       # the numpy view and the ctypes object are this object's only exports of the mapping; a caller that
        self._cbuf = None       # still holds a view of `.array` keeps it mapped, and close() then says so (BufferError) instead of hiding it
        self.shm.close()
        if self.creator:
            try:
                self.shm.unlink()
            except FileNotFoundError:
                pass
        self.shm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
