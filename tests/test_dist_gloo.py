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


def _worker_balanced(rank, world, port, W, H, out_path):
    """The band balancing of bench.py: the ranks all-gather what their bands cost, derive the same new boundaries
    from it (no broadcast), render the UNEQUAL bands and gather them."""
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
        bounds = parallel.equal_bounds(world, nby)
        times = [None] * world
        dist.all_gather_object(times, [3.0, 1.0][rank])  # rank 0's band costs three times rank 1's
        bounds = parallel.balanced_bounds(bounds, times)
        every = [None] * world
        dist.all_gather_object(every, bounds)
        assert every[0] == every[1] and bounds[1] < nby // 2, every
        rows = (bounds[rank], bounds[rank + 1])
        fr = O.frame(s, setup, W, H, rows=rows, light_grid=(32, 32), all_chunks=True)
        image = torch.from_numpy(fr["image"].copy())
        g = parallel.BandGather(dist, torch, torch.device("cpu"), W, nby, rank, world, bounds=bounds)
        g.gather(image)
        g.finish()
        dist.barrier()
        if rank == 0:
            np.save(out_path, image.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_balanced_bands(tmp_path, ugrt, O):
    import torch.multiprocessing as mp

    W, H = 128, 136
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker_balanced, args=(2, _free_port(), W, H, out), nprocs=2, join=True)
    got = np.load(out)
    s = ugrt.scenes.hall(scale=0.05)
    want = O.frame(s, ugrt.FrameSetup.from_scene(s), W, H, light_grid=(32, 32), all_chunks=True)["image"]
    np.testing.assert_array_equal(got, want)


def test_balanced_bounds_is_a_partition_and_evens_out_a_cost_profile(ugrt):
    from ugrt import parallel

    rng = np.random.default_rng(5)
    for world, nby in ((2, 17), (4, 270), (8, 382), (8, 9)):
        cost = rng.uniform(0.2, 1.0, nby) + 3.0 * np.exp(-((np.arange(nby) - 0.3 * nby) / (0.08 * nby)) ** 2)
        fixed = 0.3 * cost.sum() / world  # what every rank pays whatever its band
        bounds = parallel.equal_bounds(world, nby)
        spread = []
        for _ in range(6):
            times = [fixed + cost[bounds[r]:bounds[r + 1]].sum() for r in range(world)]
            spread.append(max(times) / (sum(times) / world))
            bounds = parallel.balanced_bounds(bounds, times)
            assert bounds[0] == 0 and bounds[-1] == nby and all(b > a for a, b in zip(bounds, bounds[1:])), bounds
        assert spread[-1] <= spread[0] + 1e-9
        if nby >= 32 * world:
            assert spread[-1] < 1.08, spread  # (the model ignores the fixed part: it under-corrects and converges)
    assert parallel.balanced_bounds([0, 1, 2, 3], [1.0, 5.0, 1.0]) == [0, 1, 2, 3]  # one row each: nothing to move


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


def _shard_worker(rank, world, port, out_path):
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
        F = s["num_faces"]
        f0, f1 = parallel.face_window(rank, world, F)
        lcam = O.cam_from(setup.light_camera, setup.fovy, 1.0)
        v3 = np.asarray(s["verts"], np.float32).reshape(-1, 3)
        out = {}
        for name in ("light", "uniform"):
            # this rank's shard: the oracle's builder on a scene in which every triangle outside the window is
            # given no references (what ugrt_ctx_set_face_window does in the count kernels)
            if name == "light":
                g = O.grid_spherical(lcam.cc, s["faces"], s["verts"], 32, 32, window=(f0, f1))
            else:
                g = O.grid_uniform(s["faces"], s["verts"], v3.min(0), v3.max(0), (16, 16, 8), window=(f0, f1))
            sh = parallel.GridShards(dist, torch, torch.device("cpu"), rank, world)
            k = torch.from_numpy(g["keys"].view(np.int32).copy())
            v = torch.from_numpy(g["vals"].view(np.int32).copy())
            sp = torch.from_numpy(g["span"].view(np.int32).copy())
            ks, vs, sps, counts = sh.exchange(k, v, sp, g["R"])
            assert counts[rank] == g["R"]
            mk, mv, msp, moff = parallel.merge_shards_numpy([x.numpy().view(np.uint32) for x in ks],
                                                            [x.numpy().view(np.uint32) for x in vs],
                                                            [x.numpy().view(np.uint32) for x in sps])
            out[name] = (mk, mv, msp, moff)
        dist.barrier()
        np.savez(out_path % rank, **{"%s_%d" % (n, i): a for n, t in out.items() for i, a in enumerate(t)})
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_grid_build(tmp_path, ugrt, O):
    """the light grid and the uniform grid are built in two shards of the triangle list, exchanged
    over the process group and merged; on BOTH ranks the result is the single-rank build, element for element."""
    import torch.multiprocessing as mp

    out = str(tmp_path / "grids_%d.npz")
    mp.spawn(_shard_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    s = ugrt.scenes.hall(scale=0.05)
    setup = ugrt.FrameSetup.from_scene(s)
    lcam = O.cam_from(setup.light_camera, setup.fovy, 1.0)
    v3 = np.asarray(s["verts"], np.float32).reshape(-1, 3)
    want = {"light": O.grid_spherical(lcam.cc, s["faces"], s["verts"], 32, 32),
            "uniform": O.grid_uniform(s["faces"], s["verts"], v3.min(0), v3.max(0), (16, 16, 8))}
    for rank in (0, 1):
        got = np.load(out % rank)
        for name, g in want.items():
            np.testing.assert_array_equal(got[name + "_0"], g["keys"])
            np.testing.assert_array_equal(got[name + "_1"], g["vals"])
            np.testing.assert_array_equal(got[name + "_2"], g["span"])
            np.testing.assert_array_equal(got[name + "_3"], g["offset"])
            assert g["R"] > 0
