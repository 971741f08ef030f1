"""display() of the reference (main.cu:59-302) over the C-ABI.

One :class:`Renderer` owns what the reference's ``DecisionData`` (decision_data.h:13-40),
``Model`` device lists (scene.h:24-28) and the image buffer own, as torch tensors, and
sequences one frame exactly in the reference's order:

  updateLightPosition -> camera -> fillCoordinatesData -> build_frustum_grid ->
  FrustumTracer::trace -> per light { light camera -> fillCoordinatesData ->
  getEffectiveRayGridMapping -> build_secondary_frustum_grid -> processData ->
  check_for_shadows } -> simpleShade | spotlight_shade -> add_shadows

(getRayGridMapping + the host max loop, main.cu:172-187, only produce values that
are overwritten with M_PI, so they are dropped: xM = yM = (float)M_PI.)
With ``reflect=True`` the shading step becomes: secondary rays -> uniform grid
build -> 3D-DDA -> shade_reflect (not in the reference).
"""
import numpy as np

from . import GRID_PERSPECTIVE, GRID_SPHERICAL, GRID_UNIFORM
from .host import Camera

PI_F = float(np.float32(np.pi))


class FrameSetup:
    def __init__(self, camera, light_camera, shading_light, fovy=45.0):
        self.camera, self.light_camera, self.shading_light, self.fovy = camera, light_camera, shading_light, fovy

    @staticmethod
    def from_scene(info, cam="ref"):
        cams = info["cameras"]
        return FrameSetup(cams[cam] if cam in cams else next(iter(cams.values())), info["light_camera"],
                          info["shading_light"])


def make_camera(params, fovy, aspect):
    c = Camera(fovy, aspect)
    c.setCameraCenter(*params["eye"])
    c.setCameraLookAt(*params["look"])
    c.setCameraUp(*params["up"])
    c.setNearFar(params["near"], params["far"])
    return c.adjustCameraAndPosition()


class Renderer:
    def __init__(self, ctx, verts, faces, matidx, mat_list, reflect=None, reflect_eps=1e-3, overlap=False,
                 shards=None, helper_thread=True, aux_stream=None, batch_builds=False):
        """overlap=True: the light grid and the uniform grid (which do not depend on the camera pass) are built
        by a second context on a second HIP stream while the main stream builds the perspective grid and
        traces the primary rays; streams are joined with events before the grids are consumed.  Same results.
        A grid build blocks its caller once (the read-back of total_refs), so the second context is driven by
        a helper thread: both streams then really run side by side."""
        t = ctx.torch
        self.ctx = ctx
        self.aux = None
        self._worker = None
        # parallel.GridShards: the light grid and the uniform grid are built in shards of the triangle list, one
        # per rank, exchanged and merged; one-stream frames only
        self.shards = shards
        assert not (overlap and shards is not None), "sharded builds run in the one-stream frame"
        # (two-stream frame from one host thread: the light and the uniform build may share their sorts' launches --
        # three launches less per frame, measured 2 % SLOWER with four frames in flight and equal with one:
        # profiles/r04_batched_builds.txt -- so it is off unless asked for)
        self.batch_builds = batch_builds
        if overlap:
            from .device import Context

            self.main_stream = t.cuda.current_stream(ctx.device)
            # (aux_stream: a stream the caller made; which streams end up on the same hardware queue depends on
            # the order in which they were created)
            self.aux_stream = aux_stream if aux_stream is not None else t.cuda.Stream(ctx.device)
            with t.cuda.stream(self.aux_stream):
                self.aux = Context(ctx.width, ctx.height, device=ctx.device_index, light_grid=ctx.light_grid,
                                   rows=ctx.rows, flags=int(ctx.cfg.flags),
                                   uniform_dims=tuple(ctx.cfg.uniform_dims[k] for k in range(3)),
                                   slabs=int(ctx.cfg.slabs))
            self._inline = not helper_thread
            if self._inline:
                # builds that never wait for the device (option async_build): one host thread keeps both streams fed
                ctx.set_option("async_build", 1)
                self.aux.set_option("async_build", 1)
            # the bounce runs beside the ray sort and the shadow pass and has slack: four persistent waves per CU leave
            # the registers and LDS of every CU to the main stream's workgroups (with the whole chip taken by
            # the bounce's waves, a sort pass of the main stream waited 0.2 ms for room)
            self.aux.set_option("dda_blocks", 1024)
            import queue
            import threading

            if self._inline:
                return self._finish_init(ctx, verts, faces, matidx, mat_list, reflect, reflect_eps)
            self._jobs, self._done = queue.Queue(), queue.Queue()

            def loop():
                while True:
                    job = self._jobs.get()
                    if job is None:
                        return
                    try:
                        job()
                        self._done.put(None)
                    except BaseException as e:  # handed to the frame loop
                        self._done.put(e)

            self._worker = threading.Thread(target=loop, name="ugrt-aux", daemon=True)
            self._worker.start()
        self._finish_init(ctx, verts, faces, matidx, mat_list, reflect, reflect_eps)

    def _finish_init(self, ctx, verts, faces, matidx, mat_list, reflect, reflect_eps):
        t = ctx.torch
        self.F = int(len(faces))
        self.num_materials = int(len(mat_list) // 6 if np.ndim(mat_list) == 1 else len(mat_list))
        self.d_verts = ctx.upload(np.asarray(verts, np.float32).reshape(-1))
        self.d_faces = ctx.upload(np.asarray(faces, np.int32).reshape(-1))
        self.d_matidx = ctx.upload(np.asarray(matidx, np.int32).reshape(-1))
        self.d_matlist = ctx.upload(np.asarray(mat_list, np.float32).reshape(-1))
        refl = np.zeros(self.num_materials, np.float32) if reflect is None else np.asarray(reflect, np.float32)
        self.d_reflect = ctx.upload(refl)
        v = np.asarray(verts, np.float32).reshape(-1, 3)
        self.bbmin, self.bbmax = v.min(0), v.max(0)
        N = ctx.width * ctx.height
        self.N = N
        # DecisionData, decision_data.h:64-78
        self.normal = ctx.empty(3 * N, t.float32)
        self.t = ctx.empty(N, t.float32)
        self.dir = ctx.empty(3 * N, t.float32)
        self.is_shadowed = ctx.empty(N, t.int32)
        self.intersect_id = ctx.empty(N, t.int32)
        self._d_map = ctx.empty(2 * ctx.npix, t.int32)
        self._prefix = ctx.empty(ctx.prefix_capacity(), t.int32)
        self.image = t.zeros(3 * N, dtype=t.uint8, device=ctx.device)
        self.cam_pos = ctx.empty(3, t.float32)
        # pinned staging for d_cam_position: a pageable-memory copy would make the host wait for the whole
        # previous frame before it may enqueue the next one.  Two buffers alternate and an event behind each copy
        # is waited for before the buffer is rewritten, so no other call has to synchronise for this to be safe
        self._cam_pos_host = [t.empty(3, dtype=t.float32).pin_memory() for _ in range(2)]
        self._cam_pos_done = [None, None]  # event behind the copy out of each staging buffer
        self._cam_pos_turn = 0
        self.rays = self.active = self.hit_t = self.hit_id = None
        self.reflect_eps = float(reflect_eps)
        self._num_chunks = 0
        self.orig = None
        self.aspect = float(np.float32(ctx.width) / np.float32(ctx.height))

    def _upload_cam_pos(self, worldori):
        """main.cu:128 d_cam_position <- worldori, without a host wait in the steady state."""
        t = self.ctx.torch
        k = self._cam_pos_turn
        self._cam_pos_turn = 1 - k
        if self._cam_pos_done[k] is not None:
            self._cam_pos_done[k].synchronize()  # the copy that read this buffer two frames ago
        self._cam_pos_host[k].copy_(t.from_numpy(worldori[:3].copy()))
        self.cam_pos.copy_(self._cam_pos_host[k], non_blocking=True)
        ev = t.cuda.Event()
        ev.record(t.cuda.current_stream(self.ctx.device))
        self._cam_pos_done[k] = ev

    @property
    def num_chunks(self):
        """h_numCudaBlocks of the last frame (fetched from the device on first use: the frame loop itself
        never waits for it)."""
        if self._num_chunks == 0xFFFFFFFF:
            self._num_chunks = self.ctx.sort_rays_chunks()
        return self._num_chunks

    # The ray map sorted by light cell and the chunk starts (processData's outputs).  With FLAG_SHADOW_ALL_CHUNKS a frame
    # does not need them (ugrt_sort_rays, deferred form): they are produced when they are looked at.
    @property
    def d_map(self):
        self.num_chunks
        return self._d_map

    @property
    def prefix(self):
        self.num_chunks
        return self._prefix

    # Model::init_orig_list, scene.h:336
    def init_orig_list(self, size, offset):
        self.orig = self.d_verts[3 * offset:3 * (offset + size)].clone()
        self.orig_size, self.orig_offset = size, offset

    # Model::rotate_bunny, scene.h:122
    def rotate_bunny(self, rot):
        self.ctx.animate(self.d_verts, self.orig, self.orig_size, self.orig_offset, rot)
        if self.aux is not None:
            self.aux.geometry_changed()  # the second context did not see the call

    def close(self):
        """Stops the helper thread of an overlapped renderer."""
        if self._worker is not None:
            self._jobs.put(None)
            self._worker.join()
            self._worker = None

    def _ensure_reflect_buffers(self):
        if self.rays is None:
            t = self.ctx.torch
            self.rays = self.ctx.empty(6 * self.N, t.float32)
            self.active = self.ctx.empty(self.N, t.int32)
            self.hit_t = self.ctx.empty(self.N, t.float32)
            self.hit_id = self.ctx.empty(self.N, t.int32)

    def display(self, setup, frame_cnt=1, shadows=True, reflect=False, shade=True):
        if self.aux is not None and shade:
            if getattr(self, "_inline", False):
                return self._display_two_streams_inline(setup, frame_cnt, shadows, reflect)
            return self._display_overlapped(setup, frame_cnt, shadows, reflect)
        ctx = self.ctx
        t = ctx.torch
        # updateLightPosition, per_frame_funcs.h:6
        ctx.set_light_position(setup.shading_light)
        cam = make_camera(setup.camera, setup.fovy, self.aspect)
        # main.cu:128 d_cam_position <- worldori ; fillCoordinatesData
        self._upload_cam_pos(cam.worldori)
        ctx.upload_camera(cam.camcoords)
        # build_frustum_grid
        ctx.grid_build_perspective(self.d_faces, self.d_verts, self.F)
        value, span, offset, _ = ctx.grid_ptrs(GRID_PERSPECTIVE)
        # FrustumTracer::trace
        ctx.trace_primary(value, span, offset, self.normal, self.t, self.dir, self.is_shadowed, self.intersect_id,
                          self.d_verts, self.d_faces)
        # dd_camcoords is the light's from here on, shadows or not (main.cu:158-170: the shading kernels read it)
        lcam = make_camera(setup.light_camera, setup.fovy, self.aspect)
        ctx.upload_camera(lcam.camcoords)
        if shadows:
            ctx.map_rays_to_light(self.t, self.dir, self._d_map, self.cam_pos, PI_F, PI_F)
            if self.shards is not None:
                self._sharded(GRID_SPHERICAL,
                              lambda: ctx.grid_build_spherical(self.d_faces, self.d_verts, self.F, PI_F, PI_F))
            else:
                ctx.grid_build_spherical(self.d_faces, self.d_verts, self.F, PI_F, PI_F)
            lvalue, lspan, loffset, _ = ctx.grid_ptrs(GRID_SPHERICAL)
            self._num_chunks = ctx.sort_rays(self._d_map, self._prefix, deferred=True)
            ctx.trace_shadow(lvalue, self.d_verts, self.d_faces, lspan, loffset, self.t, self.dir, self.is_shadowed,
                             self._d_map, self._prefix, self.cam_pos, self._num_chunks)
        if not shade:
            return
        if reflect:
            self._ensure_reflect_buffers()
            ctx.reflect_rays(self.cam_pos, self.t, self.dir, self.intersect_id, self.d_matidx, self.d_reflect,
                             self.num_materials, self.d_verts, self.d_faces, self.reflect_eps, self.rays,
                             self.active)
            if self.shards is not None:
                self._sharded(GRID_UNIFORM, lambda: ctx.grid_build_uniform(self.d_faces, self.d_verts, self.F,
                                                                            self.bbmin, self.bbmax))
            else:
                ctx.grid_build_uniform(self.d_faces, self.d_verts, self.F, self.bbmin, self.bbmax)
            uvalue, uspan, uoffset, _ = ctx.grid_ptrs(GRID_UNIFORM)
            ctx.trace_dda(uvalue, uspan, uoffset, self.d_verts, self.d_faces, self.rays, self.active, self.hit_t,
                          self.hit_id)
            ctx.shade_reflect(self.image, self.normal, self.t, self.dir, self.intersect_id, self.cam_pos,
                              self.d_matidx, self.d_matlist, self.d_reflect, self.num_materials, self.d_verts,
                              self.d_faces, self.rays, self.active, self.hit_t, self.hit_id)
        elif frame_cnt < 2:
            ctx.shade_simple(self.image, self.normal, self.t, self.dir, self.intersect_id, self.cam_pos,
                             self.d_matidx, self.d_matlist, self.num_materials)
        else:
            ctx.shade_spotlight(self.image, self.normal, self.t, self.dir, self.intersect_id, self.cam_pos,
                                self.d_matidx, self.d_matlist, self.num_materials)
        if shadows:
            ctx.shade_add_shadows(self.image, self.is_shadowed)

    def _display_overlapped(self, setup, frame_cnt, shadows, reflect):
        """display() on two streams.  Side stream (second context, driven by the helper thread): light grid,
        uniform grid, then - once the primary hits exist - secondary rays and the 3D-DDA.  Main stream: screen
        grid, primary rays, ray mapping and sort, shadow rays (after the light grid), shading (after the DDA).
        The shadow pass and the bounce only depend on the primary hits, not on each other."""
        import threading

        ctx, aux, t = self.ctx, self.aux, self.ctx.torch
        main, side = self.main_stream, self.aux_stream
        lcam = make_camera(setup.light_camera, setup.fovy, self.aspect)
        if reflect:
            self._ensure_reflect_buffers()
        ev_primary, ev_light_grid = t.cuda.Event(), t.cuda.Event()
        primary_recorded, light_grid_recorded = threading.Event(), threading.Event()
        # side stream: starts once the geometry of this frame is final on the main stream
        side.wait_stream(main)

        status = {"primary_failed": False, "light_grid_failed": False}

        def side_job():
            try:
                if shadows:
                    aux.upload_camera(lcam.camcoords)
                    aux.grid_build_spherical(self.d_faces, self.d_verts, self.F, PI_F, PI_F)
                    ev_light_grid.record(side)
            except BaseException:
                status["light_grid_failed"] = True  # the main thread must not wait for an event never recorded
                raise
            finally:
                light_grid_recorded.set()
            if reflect:
                aux.grid_build_uniform(self.d_faces, self.d_verts, self.F, self.bbmin, self.bbmax)
                primary_recorded.wait()
                if status["primary_failed"]:
                    return
                side.wait_event(ev_primary)
                aux.reflect_rays(self.cam_pos, self.t, self.dir, self.intersect_id, self.d_matidx, self.d_reflect,
                                 self.num_materials, self.d_verts, self.d_faces, self.reflect_eps, self.rays,
                                 self.active)
                uvalue, uspan, uoffset, _ = aux.grid_ptrs(GRID_UNIFORM)
                aux.trace_dda(uvalue, uspan, uoffset, self.d_verts, self.d_faces, self.rays, self.active,
                              self.hit_t, self.hit_id)

        self._jobs.put(side_job)
        failed = None
        try:
            try:
                # main stream: the camera pass
                ctx.set_light_position(setup.shading_light)
                cam = make_camera(setup.camera, setup.fovy, self.aspect)
                self._upload_cam_pos(cam.worldori)
                ctx.upload_camera(cam.camcoords)
                ctx.grid_build_perspective(self.d_faces, self.d_verts, self.F)
                value, span, offset, _ = ctx.grid_ptrs(GRID_PERSPECTIVE)
                ctx.trace_primary(value, span, offset, self.normal, self.t, self.dir, self.is_shadowed,
                                  self.intersect_id, self.d_verts, self.d_faces)
                ev_primary.record(main)
            except BaseException as e:
                status["primary_failed"] = True  # the side job skips what depends on the primary hits
                raise e
            finally:
                primary_recorded.set()
            ctx.upload_camera(lcam.camcoords)  # dd_camcoords is the light's from here on (main.cu:170)
            if shadows:
                ctx.map_rays_to_light(self.t, self.dir, self._d_map, self.cam_pos, PI_F, PI_F)
                self._num_chunks = ctx.sort_rays(self._d_map, self._prefix, deferred=True)
                light_grid_recorded.wait()
                if status["light_grid_failed"]:
                    raise RuntimeError("the light grid build on the side stream failed")
                main.wait_event(ev_light_grid)
                lvalue, lspan, loffset, _ = aux.grid_ptrs(GRID_SPHERICAL)
                ctx.trace_shadow(lvalue, self.d_verts, self.d_faces, lspan, loffset, self.t, self.dir,
                                 self.is_shadowed, self._d_map, self._prefix, self.cam_pos, self._num_chunks)
        except BaseException as e:
            failed = e
        finally:
            # exactly one result per submitted job is consumed, whatever happened above: a result left in the
            # queue would make the NEXT frame join the side stream before its own work was enqueued
            err = self._done.get()
        main.wait_stream(side)
        if failed is not None:
            raise failed
        if err is not None:
            raise err
        if reflect:
            ctx.shade_reflect(self.image, self.normal, self.t, self.dir, self.intersect_id, self.cam_pos,
                              self.d_matidx, self.d_matlist, self.d_reflect, self.num_materials, self.d_verts,
                              self.d_faces, self.rays, self.active, self.hit_t, self.hit_id)
        elif frame_cnt < 2:
            ctx.shade_simple(self.image, self.normal, self.t, self.dir, self.intersect_id, self.cam_pos,
                             self.d_matidx, self.d_matlist, self.num_materials)
        else:
            ctx.shade_spotlight(self.image, self.normal, self.t, self.dir, self.intersect_id, self.cam_pos,
                                self.d_matidx, self.d_matlist, self.num_materials)
        if shadows:
            ctx.shade_add_shadows(self.image, self.is_shadowed)

    def _display_two_streams_inline(self, setup, frame_cnt, shadows, reflect):
        """The two-stream frame from ONE host thread: with option async_build no call waits for the device, so the
        side stream's work is simply enqueued first (light grid, uniform grid), then the camera pass on the main
        stream, then what depends on the primary hits on either stream; events join them as in _display_overlapped."""
        ctx, aux, t = self.ctx, self.aux, self.ctx.torch
        main, side = self.main_stream, self.aux_stream
        lcam = make_camera(setup.light_camera, setup.fovy, self.aspect)
        if reflect:
            self._ensure_reflect_buffers()
        ev_primary, ev_light_grid = t.cuda.Event(), t.cuda.Event()
        side.wait_stream(main)  # the geometry of this frame is final on the main stream
        # the light grid and the uniform grid depend on the geometry only: their reference lists are sorted in shared
        # launches (one histogram kernel and one kernel per pass level for both: ugrt_grid_build_batch_begin / _end)
        batch = shadows and reflect and self.batch_builds
        if batch:
            aux.grid_build_batch_begin()
        if shadows:
            aux.upload_camera(lcam.camcoords)
            aux.grid_build_spherical(self.d_faces, self.d_verts, self.F, PI_F, PI_F)
            if not batch:
                ev_light_grid.record(side)
        if reflect:
            aux.grid_build_uniform(self.d_faces, self.d_verts, self.F, self.bbmin, self.bbmax)
        if batch:
            aux.grid_build_batch_end()
            ev_light_grid.record(side)
        ctx.set_light_position(setup.shading_light)
        cam = make_camera(setup.camera, setup.fovy, self.aspect)
        self._upload_cam_pos(cam.worldori)
        ctx.upload_camera(cam.camcoords)
        ctx.grid_build_perspective(self.d_faces, self.d_verts, self.F)
        value, span, offset, _ = ctx.grid_ptrs(GRID_PERSPECTIVE)
        ctx.trace_primary(value, span, offset, self.normal, self.t, self.dir, self.is_shadowed, self.intersect_id,
                          self.d_verts, self.d_faces)
        ev_primary.record(main)
        if reflect:
            side.wait_event(ev_primary)
            aux.reflect_rays(self.cam_pos, self.t, self.dir, self.intersect_id, self.d_matidx, self.d_reflect,
                             self.num_materials, self.d_verts, self.d_faces, self.reflect_eps, self.rays, self.active)
            uvalue, uspan, uoffset, _ = aux.grid_ptrs(GRID_UNIFORM)
            aux.trace_dda(uvalue, uspan, uoffset, self.d_verts, self.d_faces, self.rays, self.active, self.hit_t,
                          self.hit_id)
        ctx.upload_camera(lcam.camcoords)  # dd_camcoords is the light's from here on (main.cu:170)
        if shadows:
            ctx.map_rays_to_light(self.t, self.dir, self._d_map, self.cam_pos, PI_F, PI_F)
            self._num_chunks = ctx.sort_rays(self._d_map, self._prefix, deferred=True)
            main.wait_event(ev_light_grid)
            lvalue, lspan, loffset, _ = aux.grid_ptrs(GRID_SPHERICAL)
            ctx.trace_shadow(lvalue, self.d_verts, self.d_faces, lspan, loffset, self.t, self.dir, self.is_shadowed,
                             self._d_map, self._prefix, self.cam_pos, self._num_chunks)
        main.wait_stream(side)
        if reflect:
            ctx.shade_reflect(self.image, self.normal, self.t, self.dir, self.intersect_id, self.cam_pos,
                              self.d_matidx, self.d_matlist, self.d_reflect, self.num_materials, self.d_verts,
                              self.d_faces, self.rays, self.active, self.hit_t, self.hit_id)
        elif frame_cnt < 2:
            ctx.shade_simple(self.image, self.normal, self.t, self.dir, self.intersect_id, self.cam_pos,
                             self.d_matidx, self.d_matlist, self.num_materials)
        else:
            ctx.shade_spotlight(self.image, self.normal, self.t, self.dir, self.intersect_id, self.cam_pos,
                                self.d_matidx, self.d_matlist, self.num_materials)
        if shadows:
            ctx.shade_add_shadows(self.image, self.is_shadowed)

    def _sharded(self, which, build):
        """Runs `build` on this rank's window of the triangle list, exchanges the shards, merges them."""
        from .parallel import face_window

        ctx, sh = self.ctx, self.shards
        ctx.set_face_window(*face_window(sh.rank, sh.world, self.F))
        try:
            build()
            value, key, span, offset, gi = ctx.grid_arrays(which)
            ks, vs, sps, counts = sh.exchange(key, value, span, gi.total_refs)
            if sh.world == 1:  # the parts must not be the context's own arrays
                ks, vs, sps = [ks[0].clone()], [vs[0].clone()], [sps[0].clone()]
            ctx.grid_merge_shards(which, ks, vs, sps, counts)
            self._shard_parts = (ks, vs, sps)  # alive until the merge has run
        finally:
            ctx.set_face_window(0, -1)  # whatever happened: later builds bin every triangle again

    def synchronize(self):
        """Both contexts are synchronised before anything is raised: an overflow reported by one must not leave the
        other's report pending (it would be raised again after the frames had been repeated)."""
        err = None
        for c in (self.ctx, self.aux):
            if c is None:
                continue
            try:
                c.synchronize()
            except Exception as e:  # UGRT_EOVERFLOW of either context: one report for the pair
                err = err or e
        if err is not None:
            raise err

    def band_image(self):
        """uint8 view [rows*8, W, 3] of this context's band."""
        ctx = self.ctx
        return self.image[3 * ctx.p0:3 * (ctx.p0 + ctx.npix)].view(-1, ctx.width, 3)


class BandedRenderer:
    """ONE frame at a time, cut into bands of tile rows that run on HIP streams of their own (SURVEY 8(e)'s sharding,
    applied to the streams of one GPU instead of to GPUs).

    A frame alone cannot fill the chip: its main stream is a chain of ~45 dependent launches, most of them short and
    latency-bound (the sorts' passes, scans, run and item kernels), and only the three tracers are wide.  With B band
    contexts the chains of the bands run beside each other; the light grid, the uniform grid and the bounce stay whole
    on ONE side context (they depend on the geometry, not on the band).  Every band context writes its rows of the SAME
    per-pixel arrays (pixel ids are global: a band context writes its band), so the frame's buffers are those of a
    single-context frame, bit for bit (tests/test_gpu_parity.py::test_banded_frame_equals_the_single_context_frame).

        side:    light grid, uniform grid .................. (all primaries) secondary rays, 3D-DDA
        band i:  screen grid (rows of the band), primary ... ray map + sort, (grids) shadow pass, (DDA) shading

    No call waits for the device (option async_build on every context); one host thread enqueues the stages band by
    band."""

    def __init__(self, Context, W, H, verts, faces, matidx, mat_list, reflect=None, bands=2, device=0, light_grid=(128, 128),
                 uniform_dims=(128, 128, 64), flags=0):
        from . import parallel

        assert bands >= 1
        self.bands = bands
        nby = H // 8
        import torch as t

        self.torch = t
        dev = t.device("cuda", device)
        self.main_stream = t.cuda.current_stream(dev)
        self.side_stream = t.cuda.Stream(dev)
        with t.cuda.stream(self.side_stream):
            self.aux = Context(W, H, device=device, light_grid=light_grid, flags=flags, uniform_dims=uniform_dims)
        self.streams, self.rs = [], []
        bounds = parallel.equal_bounds(bands, nby)
        for b in range(bands):
            st = self.main_stream if b == 0 else t.cuda.Stream(self.aux.device)
            with t.cuda.stream(st):
                cx = Context(W, H, device=device, light_grid=light_grid, rows=(bounds[b], bounds[b + 1]), flags=flags,
                             uniform_dims=uniform_dims)
                r = Renderer(cx, verts, faces, matidx, mat_list, reflect)
            self.streams.append(st)
            self.rs.append(r)
        r0 = self.rs[0]
        r0._ensure_reflect_buffers()
        shared = ("normal", "t", "dir", "is_shadowed", "intersect_id", "image", "rays", "active", "hit_t", "hit_id", "d_verts",
                  "d_faces", "d_matidx", "d_matlist", "d_reflect")
        for r in self.rs[1:]:
            for name in shared:
                setattr(r, name, getattr(r0, name))
        for name in shared:
            setattr(self, name, getattr(r0, name))
        self.F, self.num_materials, self.bbmin, self.bbmax, self.aspect = r0.F, r0.num_materials, r0.bbmin, r0.bbmax, r0.aspect
        for c in [self.aux] + [r.ctx for r in self.rs]:
            c.set_option("async_build", 1)
        # (as in the two-stream frame: the bounce's persistent waves leave room for the bands' short kernels)
        self.aux.set_option("dda_blocks", 1024)

    def contexts(self):
        return [self.aux] + [r.ctx for r in self.rs]

    def display(self, setup, frame_cnt=1, shadows=True, reflect=True):
        t, aux, main, side = self.torch, self.aux, self.main_stream, self.side_stream
        r0 = self.rs[0]
        cam = make_camera(setup.camera, setup.fovy, self.aspect)
        lcam = make_camera(setup.light_camera, setup.fovy, self.aspect)
        ev_grids, ev_dda = t.cuda.Event(), t.cuda.Event()
        ev_prim = [t.cuda.Event() for _ in self.rs]
        for st in self.streams[1:] + [side]:
            st.wait_stream(main)  # the geometry of this frame (and the last frame's readers) are behind the main stream
        with t.cuda.stream(side):
            if shadows:
                aux.upload_camera(lcam.camcoords)
                aux.grid_build_spherical(self.d_faces, self.d_verts, self.F, PI_F, PI_F)
            if reflect:
                aux.grid_build_uniform(self.d_faces, self.d_verts, self.F, self.bbmin, self.bbmax)
            ev_grids.record(side)
        for r, st, ev in zip(self.rs, self.streams, ev_prim):
            with t.cuda.stream(st):
                ctx = r.ctx
                ctx.set_light_position(setup.shading_light)
                r._upload_cam_pos(cam.worldori)
                ctx.upload_camera(cam.camcoords)
                ctx.grid_build_perspective(self.d_faces, self.d_verts, self.F)
                value, span, offset, _ = ctx.grid_ptrs(GRID_PERSPECTIVE)
                ctx.trace_primary(value, span, offset, self.normal, self.t, self.dir, self.is_shadowed, self.intersect_id,
                                  self.d_verts, self.d_faces)
                ev.record(st)
        if reflect:
            with t.cuda.stream(side):
                for ev in ev_prim:
                    side.wait_event(ev)
                aux.reflect_rays(r0.cam_pos, self.t, self.dir, self.intersect_id, self.d_matidx, self.d_reflect,
                                 self.num_materials, self.d_verts, self.d_faces, r0.reflect_eps, self.rays, self.active)
                uvalue, uspan, uoffset, _ = aux.grid_ptrs(GRID_UNIFORM)
                aux.trace_dda(uvalue, uspan, uoffset, self.d_verts, self.d_faces, self.rays, self.active, self.hit_t, self.hit_id)
                ev_dda.record(side)
        for r, st in zip(self.rs, self.streams):
            with t.cuda.stream(st):
                ctx = r.ctx
                ctx.upload_camera(lcam.camcoords)  # dd_camcoords is the light's from here on (main.cu:170)
                if shadows:
                    ctx.map_rays_to_light(self.t, self.dir, r._d_map, r.cam_pos, PI_F, PI_F)
                    r._num_chunks = ctx.sort_rays(r._d_map, r._prefix, deferred=True)
                    st.wait_event(ev_grids)
                    lvalue, lspan, loffset, _ = aux.grid_ptrs(GRID_SPHERICAL)
                    ctx.trace_shadow(lvalue, self.d_verts, self.d_faces, lspan, loffset, self.t, self.dir, self.is_shadowed,
                                     r._d_map, r._prefix, r.cam_pos, r._num_chunks)
        for r, st in zip(self.rs, self.streams):
            with t.cuda.stream(st):
                ctx = r.ctx
                if reflect:
                    st.wait_event(ev_dda)
                    ctx.shade_reflect(self.image, self.normal, self.t, self.dir, self.intersect_id, r.cam_pos, self.d_matidx,
                                      self.d_matlist, self.d_reflect, self.num_materials, self.d_verts, self.d_faces, self.rays,
                                      self.active, self.hit_t, self.hit_id)
                elif frame_cnt < 2:
                    ctx.shade_simple(self.image, self.normal, self.t, self.dir, self.intersect_id, r.cam_pos, self.d_matidx,
                                     self.d_matlist, self.num_materials)
                else:
                    ctx.shade_spotlight(self.image, self.normal, self.t, self.dir, self.intersect_id, r.cam_pos, self.d_matidx,
                                        self.d_matlist, self.num_materials)
                if shadows:
                    ctx.shade_add_shadows(self.image, self.is_shadowed)
        for st in self.streams[1:] + [side]:
            main.wait_stream(st)

    def synchronize(self):
        err = None
        for c in self.contexts():
            try:
                c.synchronize()
            except Exception as e:  # UGRT_EOVERFLOW of any context: one report for the frame
                err = err or e
        if err is not None:
            raise err
