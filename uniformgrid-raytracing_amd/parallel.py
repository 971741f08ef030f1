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


def weak_scaling_resolution(world, base=(1920, 1080)):
    """Same view, more pixels: area scales with `world`, aspect stays 16:9, multiples of 8.
    1 -> 1920x1080, 2 -> 2720x1528, 4 -> 3840x2160 (BASELINE config 4), 8 -> 5432x3056."""
    s = float(world) ** 0.5
    w = int(round(base[0] * s / 8.0)) * 8
    h = int(round(base[1] * s / 8.0)) * 8
    return w, h


class BandGather:
    """Gathers every rank's RGB band into rank 0's full image with one collective."""

    def __init__(self, dist, torch, device, width, nby, rank, world, host_staging=False):
        self.dist, self.torch, self.rank, self.world = dist, torch, rank, world
        self.image_device = device
        if host_staging:  # backends without device-tensor gather (gloo): stage the bands through the host
            device = torch.device("cpu")
        self.width, self.nby = width, nby
        self.bands = [band_rows(r, world, nby) for r in range(world)]
        self.max_bytes = max(e - b for b, e in self.bands) * 8 * width * 3
        self.send = torch.zeros(self.max_bytes, dtype=torch.uint8, device=device)
        self.recv = ([torch.zeros(self.max_bytes, dtype=torch.uint8, device=device) for _ in range(world)]
                     if rank == 0 else None)

    def band_bytes(self, r):
        b, e = self.bands[r]
        return (e - b) * 8 * self.width * 3

    def gather(self, image):
        """image: uint8 [3*W*H] on the device.  Starts this frame's gather and returns: the collective runs
        beside the next frame (it only reads the staging copy of the band made here), and the received bands
        are placed into rank 0's image by the next call or by finish().  The renderer of rank 0 only ever
        writes its own band, so the late placement cannot collide with the next frame."""
        if self.world == 1:
            return image
        self._complete()
        b, _ = self.bands[self.rank]
        off = b * 8 * self.width * 3
        n = self.band_bytes(self.rank)
        self.send[:n].copy_(image[off:off + n])  # (device -> host when staging)
        self._work = self.dist.gather(self.send, self.recv, dst=0, async_op=True)
        self._image = image
        return image

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
