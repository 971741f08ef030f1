"""Image-tile sharding across the GPUs of one node.

Rays are independent, so the frame is cut into bands of tile rows, one band per
rank (one process per GPU).  Nothing is exchanged while tracing; the only
collective is ONE gather of the RGB bands to rank 0 per frame (RCCL over xGMI
when the backend is "nccl": point-to-point fan-in, each peer on its own link).
Bands are contiguous in memory because a band is a run of image rows.
"""
import numpy as np


def band_rows(rank, world, nby):
    """Tile rows [begin, end) of `rank`; sizes differ by at most one row."""
    return (rank * nby) // world, ((rank + 1) * nby) // world


def equal_bounds(world, nby):
    """Band boundaries [b_0 = 0, ..., b_world = nby] of band_rows."""
    return [band_rows(r, world, nby)[0] for r in range(world)] + [nby]


def balanced_bounds(bounds, times, fixed=0.0):
    """New band boundaries from the frame times the ranks measured on the current ones (SURVEY.md 8e shards by
    tile rows; which rows is free).  A band's time is taken as `fixed` (what every rank pays whatever its band:
    the replicated light and uniform grid builds) + a cost spread evenly over its tile rows; the new boundaries
    cut the accumulated cost into equal parts.  Every band keeps at least one row; the result is a pure function
    of its arguments, so ranks that all-gather `times` agree on it without a broadcast."""
    world = len(times)
    assert len(bounds) == world + 1 and bounds[0] == 0
    nby = bounds[-1]
    dens = []
    for r in range(world):
        rows = bounds[r + 1] - bounds[r]
        dens += [max(times[r] - fixed, 1e-9) / rows] * rows
    cum = np.concatenate([[0.0], np.cumsum(dens)])
    out = [0]
    for r in range(1, world):
        target = cum[-1] * r / world
        b = int(np.searchsorted(cum, target, side="left"))
        # the nearer of the two rows around the target
        if b > 0 and abs(cum[b - 1] - target) <= abs(cum[min(b, nby)] - target):
            b -= 1
        b = max(b, out[-1] + 1)
        b = min(b, nby - (world - r))
        out.append(b)
    return out + [nby]


def weak_scaling_resolution(world, base=(1920, 1080)):
    """Same view, more pixels: area scales with `world`, aspect stays 16:9, multiples of 8.
    1 -> 1920x1080, 2 -> 2720x1528, 4 -> 3840x2160 (BASELINE config 4), 8 -> 5432x3056."""
    s = float(world) ** 0.5
    w = int(round(base[0] * s / 8.0)) * 8
    h = int(round(base[1] * s / 8.0)) * 8
    return w, h


class BandGather:
    """Gathers every rank's RGB band into rank 0's full image with one collective."""

    def __init__(self, dist, torch, device, width, nby, rank, world, host_staging=False, bounds=None):
        self.dist, self.torch, self.rank, self.world = dist, torch, rank, world
        self.image_device = device
        if host_staging:  # backends without device-tensor gather (gloo): stage the bands through the host
            device = torch.device("cpu")
        self.width, self.nby = width, nby
        bounds = bounds if bounds is not None else equal_bounds(world, nby)
        self.bands = [(bounds[r], bounds[r + 1]) for r in range(world)]
        self.max_bytes = max(e - b for b, e in self.bands) * 8 * width * 3
        self.send = torch.zeros(self.max_bytes, dtype=torch.uint8, device=device)
        self.recv = ([torch.zeros(self.max_bytes, dtype=torch.uint8, device=device) for _ in range(world)]
                     if rank == 0 else None)

    def band_bytes(self, r):
        b, e = self.bands[r]
        return (e - b) * 8 * self.width * 3

    def gather(self, image, wait=False):
        """image: uint8 [3*W*H] on the device.  Starts this frame's gather: the collective runs beside the next
        frame (it only reads the staging copy of the band made here), and the received bands are placed into
        rank 0's image by the next call or by finish().  The renderer of rank 0 only ever writes its own band, so
        the late placement cannot collide with the next frame.
        Returns the image only when it is COMPLETE (world == 1, or wait=True, which finishes the collective before
        returning: what a caller that writes or shows every frame needs); otherwise None - until finish() rank
        0's image holds its own band of this frame beside the other ranks' bands of the previous one."""
        if self.world == 1:
            return image
        self._complete()
        b, _ = self.bands[self.rank]
        off = b * 8 * self.width * 3
        n = self.band_bytes(self.rank)
        self.send[:n].copy_(image[off:off + n])  # (device -> host when staging)
        self._work = self.dist.gather(self.send, self.recv, dst=0, async_op=True)
        self._image = image
        if wait:
            self._complete()
            return image
        return None

    def finish(self):
        """Waits for the gather in flight; rank 0's image then holds every band of the last gathered frame."""
        self._complete()

    def _complete(self):
        work, image = getattr(self, "_work", None), getattr(self, "_image", None)
        if work is None:
            return
        work.wait()  # device backends: the current stream waits, the host does not
        self._work = None
        if self.rank == 0:
            for r in range(1, self.world):
                rb, _ = self.bands[r]
                roff, rn = rb * 8 * self.width * 3, self.band_bytes(r)
                image[roff:roff + rn].copy_(self.recv[r][:rn])


def face_window(rank, world, num_faces):
    """Triangles [begin, end) whose light / uniform grid references `rank` builds."""
    return (rank * num_faces) // world, ((rank + 1) * num_faces) // world


class GridShards:
    """Exchange of the ranks' shards of a grid build: every rank ends up with every rank's (keys, values, span).

    A shard is a complete set of grid arrays for the rank's window of the triangle list; the windows are disjoint
    and ascending with the rank, so merging is concatenation per cell in rank order (ugrt_grid_merge_shards on the
    device, merge_shards_numpy below as the same arithmetic on the host).  Three collectives per grid: the counts,
    then keys and values padded to the largest shard, and the per-cell spans."""

    def __init__(self, dist, torch, device, rank, world, host_staging=False):
        self.dist, self.torch, self.rank, self.world = dist, torch, rank, world
        self.device = torch.device("cpu") if host_staging else device
        self.out_device = device

    def exchange(self, keys, vals, span, count):
        """keys/vals: int32 tensors with >= count entries, span: int32 [C].  Returns lists over the ranks of
        (keys, vals, span) tensors on the device plus the list of counts."""
        t, dist = self.torch, self.dist
        if self.world == 1:
            return [keys[:count]], [vals[:count]], [span], [count]
        cnt = t.tensor([count], dtype=t.int64, device=self.device)
        counts = [t.zeros(1, dtype=t.int64, device=self.device) for _ in range(self.world)]
        dist.all_gather(counts, cnt)
        counts = [int(c.item()) for c in counts]
        cap = max(max(counts), 1)
        send = t.zeros(2 * cap, dtype=t.int32, device=self.device)
        send[:count].copy_(keys[:count])
        send[cap:cap + count].copy_(vals[:count])
        recv = [t.zeros(2 * cap, dtype=t.int32, device=self.device) for _ in range(self.world)]
        dist.all_gather(recv, send)
        sp = span.to(self.device).contiguous()
        spans = [t.zeros_like(sp) for _ in range(self.world)]
        dist.all_gather(spans, sp)
        dev = self.out_device
        ks = [recv[r][:counts[r]].to(dev) for r in range(self.world)]
        vs = [recv[r][cap:cap + counts[r]].to(dev) for r in range(self.world)]
        return ks, vs, [x.to(dev) for x in spans], counts


def merge_shards_numpy(keys, vals, spans):
    """The merge of ugrt_grid_merge_shards on the host: (keys, vals, span, offset) of the full build."""
    C = len(spans[0])
    total = np.zeros(C, np.int64)
    before = []
    for sp in spans:
        before.append(total.copy())
        total += sp.astype(np.int64)
    offset = np.concatenate([[0], np.cumsum(total)[:-1]])
    R = int(total.sum())
    okeys, ovals = np.zeros(R, np.uint32), np.zeros(R, np.uint32)
    for r, (k, v, sp) in enumerate(zip(keys, vals, spans)):
        k = np.asarray(k).astype(np.int64)
        start = np.concatenate([[0], np.cumsum(sp.astype(np.int64))[:-1]])  # run starts inside the shard
        pos = offset[k] + before[r][k] + (np.arange(len(k)) - start[k])
        okeys[pos] = k
        ovals[pos] = np.asarray(v).astype(np.uint32)
    return okeys, ovals, total.astype(np.uint32), offset.astype(np.uint32)
