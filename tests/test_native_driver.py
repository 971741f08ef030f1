"""The reference-side binding, compiled: integration/ugrt_shim.h (the class stubs INTEGRATION.md shows) and
integration/display_main.cpp (display() of main.cu:59-302 in C++ over the C-ABI, one- and two-stream)."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INTEG = os.path.join(ROOT, "integration")
BIN = os.path.join(INTEG, "display_main")


def _build():
    subprocess.run(["make", "-C", INTEG], check=True, capture_output=True)
    assert os.path.exists(BIN)


def test_shim_compiles_against_the_header():
    """hipcc builds the driver (ugrt.h + the shim + the frame loop) and links it to libugrt.so; no GPU needed."""
    _build()
    out = subprocess.run(["ldd", BIN], capture_output=True, text=True).stdout
    assert "libugrt.so" in out and "oracle" not in out


def test_integration_md_quotes_the_shim_verbatim():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    shim = open(os.path.join(INTEG, "ugrt_shim.h")).read()
    blocks = re.findall(r"```cpp\n(.*?)```", md, re.S)
    assert any(b == shim for b in blocks), "INTEGRATION.md must contain integration/ugrt_shim.h as a cpp block"


def _read_ppm(path):
    tok = open(path).read().split()
    assert tok[0] == "P3" and tok[3] == "255"
    w, h = int(tok[1]), int(tok[2])
    return np.array(tok[4:], dtype=np.int64).astype(np.uint8).reshape(h * w * 3), w, h


@pytest.mark.gpu
@pytest.mark.parametrize("streams,reflect,all_chunks,ranks", [(1, 0, 0, 0), (2, 0, 1, 0), (1, 1, 1, 0), (2, 1, 1, 0),
                                                              (1, 0, 0, 1), (2, 1, 1, 1)])
def test_cpp_display_writes_the_oracle_image(tmp_path, ugrt, O, streams, reflect, all_chunks, ranks):
    """A C++ host (no Python, no torch in the process) loads the OBJ + material file, runs display() through the
    shim classes and writes the PPM; the pixels are the oracle's frame.  ranks = 1: the frame's RGB goes through the
    RCCL band gather (ncclSend / ncclRecv on a communicator of one rank; N > 1 needs N GPUs)."""
    _build()
    d = str(tmp_path)
    s = ugrt.scenes.hall(d, scale=0.1) if not reflect else ugrt.scenes.crash(d, scale=0.02)
    W, H = 256, 256
    cam, lcam = s["cameras"]["ref"], s["light_camera"]
    flat = lambda c: " ".join("%.9g" % v for v in (list(c["eye"]) + list(c["look"]) + list(c["up"]) + [c["near"], c["far"]]))
    params = os.path.join(d, "params.txt")
    with open(params, "w") as f:
        f.write("obj %s\nmat %s\nsize %d %d\ncamera %s\nlight_camera %s\nshading_light %s\nstreams %d\nreflect %d\n"
                "frames 2\nflags %d\n%s" % (s["obj"], s["mat"], W, H, flat(cam), flat(lcam),
                                          " ".join("%.9g" % v for v in s["shading_light"]), streams, reflect,
                                          ugrt.FLAG_SHADOW_ALL_CHUNKS if all_chunks else 0,
                                          "ranks %d\n" % ranks if ranks else ""))
    out = os.path.join(d, "out.ppm")
    # the mtllib is opened relative to the cwd (obj_parser.cpp:417)
    p = subprocess.run([BIN, params, out], cwd=d, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    got, w, h = _read_ppm(out)
    assert (w, h) == (W, H)
    m = ugrt.Model()
    m.some_material(s["mat"])
    cwd = os.getcwd()
    os.chdir(d)
    try:
        m.load_model(s["obj"])
    finally:
        os.chdir(cwd)
    sc = dict(s, verts=m.h_vertexlist.reshape(-1, 3), faces=m.h_facelist.reshape(-1, 3), matidx=m.h_materiallist_index,
              mat_list=m.h_materiallist.reshape(-1, 6), reflect=m.h_reflectlist)
    setup = ugrt.FrameSetup(cam, lcam, s["shading_light"])
    # frame 2 of the loop: spot shading without the bounce (main.cu:205-219, Q19); the bounce shades as frame 1
    want = O.frame(sc, setup, W, H, light_grid=(128, 128), all_chunks=bool(all_chunks), reflect=bool(reflect),
                   uniform_dims=(128, 128, 64), frame_cnt=2)
    np.testing.assert_array_equal(got, want["image"])
    assert want["image"].max() > 0 and want["is_shadowed"].sum() > 0


def _group_is_empty(pgid):
    try:
        os.killpg(pgid, 0)
    except ProcessLookupError:
        return True
    return False


def _run_failing_ranks(tmp_path, ugrt, ranks, extra=""):
    import signal
    import time

    _build()
    d = str(tmp_path)
    s = ugrt.scenes.hall(d, scale=0.05)
    cam, lcam = s["cameras"]["ref"], s["light_camera"]
    flat = lambda c: " ".join("%.9g" % v for v in (list(c["eye"]) + list(c["look"]) + list(c["up"]) + [c["near"], c["far"]]))
    params = os.path.join(d, "params.txt")
    with open(params, "w") as f:
        f.write("obj %s\nmat %s\nsize 128 128\ncamera %s\nlight_camera %s\nshading_light 1 2 3\nranks %d\n%s"
                % (s["obj"], s["mat"], flat(cam), flat(lcam), ranks, extra))
    t0 = time.time()
    p = subprocess.Popen([BIN, params, os.path.join(d, "out.ppm")], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, start_new_session=True)  # its own process group: nothing of it may be left behind
    try:
        out, err = p.communicate(timeout=60)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        p.communicate()
        raise AssertionError("display_main with a failing rank hung (no exit within 60 s)")
    took = time.time() - t0
    for _ in range(100):  # (killed children are reaped by init a moment later)
        if _group_is_empty(p.pid):
            break
        time.sleep(0.05)
    left = not _group_is_empty(p.pid)
    if left:
        os.killpg(p.pid, signal.SIGKILL)
    return p.returncode, took, left, out + err


def test_cpp_display_rank_failure_is_an_exit_code_not_a_hang(tmp_path, ugrt):
    """`ranks N` with one rank more than the box has GPUs (here, without a GPU: every rank): that rank's hipSetDevice
    fails and it exits; rank 0 (blocked in ncclCommInitRank, or failing itself) must end the job with a non-zero exit
    code within seconds and leave no process behind (watchdog threads + the shared failure flag)."""
    import torch

    ranks = torch.cuda.device_count() + 1  # (counting devices does not initialise the GPU)
    rc, took, left, log = _run_failing_ranks(tmp_path, ugrt, max(2, ranks))
    assert rc != 0, log
    assert took < 30.0, "took %.1f s: %s" % (took, log)
    assert not left, "processes of the job outlived it"


@pytest.mark.gpu
@pytest.mark.parametrize("extra", ["", "rendezvous_timeout 2\n"])
def test_cpp_display_failing_rank_on_the_gpu_box(tmp_path, ugrt, extra):
    """The same on the GPU box (`ranks` = its GPUs + 1: the last rank cannot get a device while rank 0 waits in
    ncclCommInitRank), with the default start-up deadline and with one of 2 s: whichever notices first (the reaped
    child or the deadline), the job ends non-zero, at once, and leaves no process."""
    import torch

    rc, took, left, log = _run_failing_ranks(tmp_path, ugrt, torch.cuda.device_count() + 1, extra)
    assert rc != 0 and took < 30.0 and not left, (rc, took, left, log)


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts the two ranks itself (it never touches
    the GPU), relays rank 0's line and their exit code.  Rehearsal on one GPU (gloo, host staging), verified against
    the single-context frame."""
    import json
    import sys

    env = dict(os.environ, UGRT_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--cpu-seconds", "0",
           "--repeats", "0", "--scale", "0.05", "--width", "640", "--height", "360", "--balance-rounds", "1"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["verified_against_single_context_frame"] is True


@pytest.mark.gpu
def test_bench_fails_when_the_timed_frames_do_not_verify():
    """bench.py compares what its timed renderers hold with one sequential context; a difference (forced here by
    flipping one pixel before the comparison) must end the run with a non-zero exit code, `value` null and an `error`
    field in the line - never a headline number with a false flag beside it."""
    import json
    import sys

    env = dict(os.environ, UGRT_BENCH_FORCE_MISMATCH="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--repeats", "0",
           "--scale", "0.05", "--width", "640", "--height", "360"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0, p.stdout[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["verified_against_single_context_frame"] is False and line["value"] is None and "error" in line
    env.pop("UGRT_BENCH_FORCE_MISMATCH")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["verified_against_single_context_frame"] is True and line["value"] > 0 and "error" not in line


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--shard-builds"], ["--config3"]])
def test_bench_two_ranks_on_one_gpu_gather_the_right_image(extra):
    """The N > 1 path of bench.py end to end with the real kernels: two ranks (both on this one GPU, gloo with host
    staging: a rehearsal, not RCCL) render their bands with several frames in flight and builds that never wait, rank 0
    gathers them, and the gathered image equals the same frame rendered whole by one context."""
    import json
    import sys

    env = dict(os.environ, UGRT_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29600 + (os.getpid() % 300)
    scale = ["--scale", "0.05"] if "--config3" not in extra else ["--scale", "0.02"]
    size = [] if "--config3" in extra else ["--width", "640", "--height", "360"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4",
           "--warmup", "2", "--cpu-seconds", "0", "--repeats", "0", "--verify"] + scale + size + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["verified_against_single_context_frame"] is True
    if "--shard-builds" not in extra:  # the bands were timed and moved before the measurement: still a partition
        b = line["band_bounds_tile_rows"]
        assert b[0] == 0 and len(b) == 3 and b[0] < b[1] < b[2] and line["band_balance_rounds"], line
