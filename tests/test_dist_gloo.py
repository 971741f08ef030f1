"""N > 1 path on CPU: world_size 2, gloo.  Each rank renders ITS band of tile rows (here with the CPU oracle as
the renderer, since no GPU exists in this container) and the product's BandGather -- the code bench.py runs over
RCCL -- assembles the frame on rank 0, which must equal the single-process frame."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    import oracle_lib as O
    import ugrt
    from ugrt import parallel

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        O.set_threads(2)
        s = ugrt.scenes.hall(scale=0.05)
        setup = ugrt.FrameSetup.from_scene(s)
        nby = H // 8
        rows = parallel.band_rows(rank, world, nby)
        fr = O.frame(s, setup, W, H, rows=rows, light_grid=(32, 32), all_chunks=True)
        image = torch.from_numpy(fr["image"].copy())
        g = parallel.BandGather(dist, torch, torch.device("cpu"), W, nby, rank, world)
        g.gather(image)
        g.gather(image)  # a second frame: the first one's bands are placed, the buffers are reused
        g.finish()
        dist.barrier()
        if rank == 0:
            np.save(out_path, image.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("H", [128, 136])  # 16 tile rows (even split) and 17 (uneven bands)
def test_two_rank_band_gather(tmp_path, ugrt, O, H):
    import torch.multiprocessing as mp

    W = 128
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(2, _free_port(), W, H, out), nprocs=2, join=True)
    got = np.load(out)
    s = ugrt.scenes.hall(scale=0.05)
    want = O.frame(s, ugrt.FrameSetup.from_scene(s), W, H, light_grid=(32, 32), all_chunks=True)["image"]
    np.testing.assert_array_equal(got, want)
    assert want.max() > 0


def test_band_rows_partition(ugrt):
    from ugrt import parallel

    for nby in (135, 270, 382, 17):
        for world in (1, 2, 4, 8):
            bands = [parallel.band_rows(r, world, nby) for r in range(world)]
            assert bands[0][0] == 0 and bands[-1][1] == nby
            assert all(bands[i][1] == bands[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in bands]
            assert max(sizes) - min(sizes) <= 1
    assert parallel.weak_scaling_resolution(1) == (1920, 1080)
    assert parallel.weak_scaling_resolution(4) == (3840, 2160)
    for w in (2, 8):
        W, H = parallel.weak_scaling_resolution(w)
        assert W % 8 == 0 and H % 8 == 0 and abs(W * H / (1920 * 1080.0) - w) < 0.02 * w
