"""Device context: one per GPU / per image band.  Thin wrappers over the C-ABI.

Every tensor argument must be a contiguous torch tensor on the context's device;
only its ``data_ptr()`` crosses the boundary.
"""
import ctypes as C

import numpy as np

from . import lib, check, Config, GridInfo, SlabInfo, STAGES, _f3, _P


def _ptr(t):
    """torch tensor -> device pointer; raw pointers (int / c_void_p) pass through."""
    if t is None:
        return None
    if isinstance(t, (int, C.c_void_p)):
        return t
    assert t.is_contiguous(), "ugrt: tensors must be contiguous"
    return C.c_void_p(t.data_ptr())


class Context:
    def __init__(self, width, height, device=0, light_grid=(128, 128), rows=None, flags=0,
                 uniform_dims=(64, 64, 32), slabs=1):
        import torch  # device memory, streams

        self.torch = torch
        cfg = Config()
        cfg.width, cfg.height, cfg.tile, cfg.slabs = width, height, 8, slabs
        cfg.light_nbx, cfg.light_nby = light_grid
        nby = height // 8
        cfg.row_begin, cfg.row_end = rows if rows is not None else (0, nby)
        cfg.flags = flags
        for k in range(3):
            cfg.uniform_dims[k] = uniform_dims[k]
        self.cfg = cfg
        self.width, self.height = width, height
        self.nbx, self.nby = width // 8, nby
        self.rows = (cfg.row_begin, cfg.row_end)
        self.p0 = cfg.row_begin * 8 * width
        self.npix = (cfg.row_end - cfg.row_begin) * 8 * width
        self.light_grid = tuple(light_grid)
        self.device_index = device
        self._h = _P()
        check(lib.ugrt_ctx_create(C.byref(self._h), device, C.byref(cfg)))
        self.device = torch.device("cuda", device)
        self.use_current_stream()

    def __del__(self):
        if getattr(self, "_h", None):
            lib.ugrt_ctx_destroy(self._h)
            self._h = None

    # -- plumbing ----------------------------------------------------------
    def use_current_stream(self):
        s = self.torch.cuda.current_stream(self.device)
        check(lib.ugrt_ctx_set_stream(self._h, C.c_void_p(s.cuda_stream)))

    def synchronize(self):
        check(lib.ugrt_ctx_synchronize(self._h))

    def empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=dtype, device=self.device)

    def upload(self, arr):
        return self.torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)

    def wrap_u32(self, ptr, n):
        """int32 tensor view of n uint32 values that the context owns (grid arrays)."""
        if n == 0:
            return self.torch.empty(0, dtype=self.torch.int32, device=self.device)
        key = (int(ptr), int(n))
        cache = self.__dict__.setdefault("_wrap_cache", {})
        if key in cache:
            return cache[key]
        if len(cache) > 64:
            cache.clear()
        cache[key] = self._wrap_u32_uncached(ptr, n)
        return cache[key]

    def _wrap_u32_uncached(self, ptr, n):
        iface = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 2}
        holder = type("_P", (), {"__cuda_array_interface__": iface})()
        return self.torch.as_tensor(holder, device=self.device)

    # -- camera / light ----------------------------------------------------
    def upload_camera(self, camcoords):
        check(lib.ugrt_upload_camera(self._h, _f3(camcoords)))

    def set_light_position(self, pos):
        check(lib.ugrt_set_light_position(self._h, _f3(pos)))

    # -- grids -------------------------------------------------------------
    def grid_build_perspective(self, d_faces, d_verts, num_faces):
        check(lib.ugrt_grid_build_perspective(self._h, _ptr(d_faces), _ptr(d_verts), num_faces))

    def grid_build_spherical(self, d_faces, d_verts, num_faces, xM, yM):
        check(lib.ugrt_grid_build_spherical(self._h, _ptr(d_faces), _ptr(d_verts), num_faces, xM, yM))

    def grid_build_uniform(self, d_faces, d_verts, num_faces, bbmin, bbmax):
        check(lib.ugrt_grid_build_uniform(self._h, _ptr(d_faces), _ptr(d_verts), num_faces, _f3(bbmin), _f3(bbmax)))

    def grid_build_batch_begin(self):
        """The grid builds that follow (at most two, asynchronous form) stop in front of their sorts ..."""
        check(lib.ugrt_grid_build_batch_begin(self._h))

    def grid_build_batch_end(self):
        """... and are sorted in shared launches and completed here; the grids must not be used before."""
        check(lib.ugrt_grid_build_batch_end(self._h))

    def grid_info(self, which):
        gi = GridInfo()
        check(lib.ugrt_grid_get_info(self._h, which, C.byref(gi)))
        return gi

    def set_face_window(self, begin, end):
        """Triangles [begin, end) the light / uniform builds bin (end < 0: to the last one; 0, -1 = all; begin == end:
        none): this rank's shard of the build."""
        check(lib.ugrt_ctx_set_face_window(self._h, int(begin), int(end)))

    def grid_merge_shards(self, which, keys, vals, spans, counts):
        """keys/vals/spans: per rank, device tensors (or pointers) of that rank's shard; counts: refs per rank."""
        n = len(counts)
        arr = lambda xs: (_P * n)(*[_ptr(x) for x in xs])
        check(lib.ugrt_grid_merge_shards(self._h, which, n, arr(keys), arr(vals), arr(spans),
                                         (C.c_uint * n)(*[int(c) for c in counts])))

    def grid_slabs(self, which):
        si = SlabInfo()
        check(lib.ugrt_grid_get_slabs(self._h, which, C.byref(si)))
        return si

    def grid_ptrs(self, which):
        """(value, span, offset) as raw device pointers + the GridInfo: what a frame loop passes on."""
        gi = self.grid_info(which)
        return gi.d_triangle_value_list, gi.d_span, gi.d_offset, gi

    def grid_arrays(self, which):
        """(value, key, span, offset) as int32 torch views + the GridInfo (tests, analysis)."""
        gi = self.grid_info(which)
        return (self.wrap_u32(gi.d_triangle_value_list, gi.total_refs),
                self.wrap_u32(gi.d_triangle_key_list, gi.total_refs),
                self.wrap_u32(gi.d_span, gi.num_cells), self.wrap_u32(gi.d_offset, gi.num_cells), gi)

    def set_option(self, key, value):
        """Launch-shape options (no effect on results), e.g. "dda_rays_per_wave"."""
        check(lib.ugrt_ctx_set_option(self._h, key.encode(), int(value)))

    def get_state(self, key):
        """Counters and findings of the context: "radix_launches", "sort_rank_atomic", "recip_mismatches" (runs the
        exhaustive check of the tracers' reciprocal on the device)."""
        v = C.c_longlong(0)
        check(lib.ugrt_ctx_get_state(self._h, key.encode(), C.byref(v)))
        return int(v.value)

    def geometry_changed(self):
        """With FLAG_STATIC_GEOMETRY: the vertex or face array was rewritten outside ugrt_animate."""
        check(lib.ugrt_geometry_changed(self._h))

    def sort_pairs(self, keys_in, keys_out, values_in, values_out, key_bits, library=False):
        """cudppSort on (uint key, uint value) pairs: stable, on key bits [0, key_bits)."""
        check(lib.ugrt_sort_pairs(self._h, _ptr(keys_in), _ptr(keys_out), _ptr(values_in), _ptr(values_out),
                                  keys_in.numel(), key_bits, 1 if library else 0))

    # -- tracing -----------------------------------------------------------
    def trace_primary(self, value, span, offset, normal, t, ray_dir, shadowed, ids, verts, faces):
        check(lib.ugrt_trace_primary(self._h, _ptr(value), _ptr(span), _ptr(offset), _ptr(normal), _ptr(t),
                                     _ptr(ray_dir), _ptr(shadowed), _ptr(ids), _ptr(verts), _ptr(faces)))

    def map_rays_to_light(self, t, ray_dir, d_map, cam_pos, xM, yM):
        check(lib.ugrt_map_rays_to_light(self._h, _ptr(t), _ptr(ray_dir), _ptr(d_map), _ptr(cam_pos), xM, yM))

    def prefix_capacity(self):
        return self.npix // 64 + self.light_grid[0] * self.light_grid[1] + 2

    def sort_rays(self, d_map, d_prefix, deferred=False):
        """processData.  deferred=True: does not wait for the device; returns CHUNKS_ON_DEVICE, which
        trace_shadow accepts, and sort_rays_chunks() fetches the number later."""
        if deferred:
            check(lib.ugrt_sort_rays(self._h, _ptr(d_map), _ptr(d_prefix), d_prefix.numel(), None))
            return 0xFFFFFFFF
        n = C.c_uint()
        check(lib.ugrt_sort_rays(self._h, _ptr(d_map), _ptr(d_prefix), d_prefix.numel(), C.byref(n)))
        return n.value

    def sort_rays_chunks(self):
        n = C.c_uint()
        check(lib.ugrt_sort_rays_chunks(self._h, C.byref(n)))
        return n.value

    def trace_shadow(self, value, verts, faces, span, offset, t, ray_dir, is_shadowed, d_map, d_prefix, cam_pos,
                     num_chunks):
        check(lib.ugrt_trace_shadow(self._h, _ptr(value), _ptr(verts), _ptr(faces), _ptr(span), _ptr(offset),
                                    _ptr(t), _ptr(ray_dir), _ptr(is_shadowed), _ptr(d_map), _ptr(d_prefix),
                                    _ptr(cam_pos), num_chunks))

    # -- shading -----------------------------------------------------------
    def shade_simple(self, img, normal, t, ray_dir, ids, cam_pos, mat_idx, mat_list, num_materials):
        check(lib.ugrt_shade_simple(self._h, _ptr(img), _ptr(normal), _ptr(t), _ptr(ray_dir), _ptr(ids),
                                    _ptr(cam_pos), _ptr(mat_idx), _ptr(mat_list), num_materials))

    def shade_spotlight(self, img, normal, t, ray_dir, ids, cam_pos, mat_idx, mat_list, num_materials, dump=None):
        check(lib.ugrt_shade_spotlight(self._h, _ptr(img), _ptr(normal), _ptr(t), _ptr(ray_dir), _ptr(ids),
                                       _ptr(cam_pos), _ptr(mat_idx), _ptr(mat_list), num_materials, _ptr(dump)))

    def shade_add_shadows(self, img, is_shadowed):
        check(lib.ugrt_shade_add_shadows(self._h, _ptr(img), _ptr(is_shadowed)))

    def shade_perlin(self, img, t, ray_dir, cam_pos, ids):
        check(lib.ugrt_shade_perlin(self._h, _ptr(img), _ptr(t), _ptr(ray_dir), _ptr(cam_pos), _ptr(ids)))

    # -- reflection bounce ---------------------------------------------------
    def reflect_rays(self, cam_pos, t, ray_dir, ids, mat_idx, reflect, num_materials, verts, faces, eps, rays,
                     active):
        check(lib.ugrt_reflect_rays(self._h, _ptr(cam_pos), _ptr(t), _ptr(ray_dir), _ptr(ids), _ptr(mat_idx),
                                    _ptr(reflect), num_materials, _ptr(verts), _ptr(faces), eps, _ptr(rays),
                                    _ptr(active)))

    def trace_dda(self, value, span, offset, verts, faces, rays, active, hit_t, hit_id):
        check(lib.ugrt_trace_dda(self._h, _ptr(value), _ptr(span), _ptr(offset), _ptr(verts), _ptr(faces),
                                 _ptr(rays), _ptr(active), _ptr(hit_t), _ptr(hit_id)))

    def shade_reflect(self, img, normal, t, ray_dir, ids, cam_pos, mat_idx, mat_list, reflect, num_materials,
                      verts, faces, rays, active, hit_t, hit_id):
        check(lib.ugrt_shade_reflect(self._h, _ptr(img), _ptr(normal), _ptr(t), _ptr(ray_dir), _ptr(ids),
                                     _ptr(cam_pos), _ptr(mat_idx), _ptr(mat_list), _ptr(reflect), num_materials,
                                     _ptr(verts), _ptr(faces), _ptr(rays), _ptr(active), _ptr(hit_t),
                                     _ptr(hit_id)))

    # -- animation -----------------------------------------------------------
    def animate(self, verts, orig, size, offset, rot):
        check(lib.ugrt_animate(self._h, _ptr(verts), _ptr(orig), size, offset, rot))

    # -- profiling -----------------------------------------------------------
    def prof_enable(self, on=True, stages=None):
        """stages: iterable of stage names to time (default: all)."""
        if on and stages is not None:
            mask = 0
            for name in stages:
                mask |= 1 << (STAGES.index(name) + 1)
            check(lib.ugrt_prof_enable(self._h, mask))
        else:
            check(lib.ugrt_prof_enable(self._h, 1 if on else 0))

    def prof_reset(self):
        check(lib.ugrt_prof_reset(self._h))

    def prof_get(self):
        """{stage name: (total ms, launches)} since the last reset."""
        out = {}
        for i, name in enumerate(STAGES):
            ms, n = C.c_double(), C.c_int()
            check(lib.ugrt_prof_get(self._h, i, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def stats(self):
        a = (C.c_ulonglong * 8)()
        check(lib.ugrt_stats_get(self._h, a))
        return list(a)

    PRIMARY_STATS = ("items", "batches", "batches_with_tile_survivors", "references", "tile_survivors",
                     "quadrant_survivors", "jobs", "flushes", "rounds", "rounds_to_division", "rounds_to_v",
                     "rounds_to_t", "lane_tests", "hits", "jobs_behind_the_flush_s_final_hits", "jobs_dropped_by_the_depth_bound")

    def stats_primary(self):
        """Work counters of the primary tracer's last counting launch (a FLAG_COUNT_WORK context)."""
        a = (C.c_ulonglong * 16)()
        check(lib.ugrt_stats_primary(self._h, a, 16))
        return dict(zip(self.PRIMARY_STATS, list(a)))

    WALK_STATS = ("windows", "jobs", "job_rays", "cull_batches", "cull_tests", "rounds", "round_pairs", "empty_windows")

    def stats_dda_split(self):
        """Split walks of the last trace_dda (option dda_split): segments the cut groups were listed as, jobs of the launch
        before, cut groups with rays that were walked again in one piece, those rays."""
        a = (C.c_uint * 4)()
        check(lib.ugrt_stats_dda_split(self._h, a))
        return dict(zip(("segments", "jobs_of_the_launch_before", "groups_walked_again", "rays_walked_again"), list(a)))

    def stats_dda(self, kernel=0):
        """Work sharing of the window kernel's last counting launch (a FLAG_COUNT_WORK context, "dda_kernel" 0)."""
        a = (C.c_ulonglong * 46)()
        check(lib.ugrt_stats_dda(self._h, a, 46))
        d = dict(zip(self.WALK_STATS, list(a)[:8]))
        d["waves_by_log2_cycles_over_4096"] = list(a)[8:24]
        d["cycles_sum"], d["cycles_max"] = a[24], a[25]
        phases = ("plan_bitmap", "job_list_headers", "operand_arrival", "box", "cull", "exact_rounds", "settle", "rest")
        d["phase_cycles"] = dict(zip(phases, list(a)[26:34]))
        d["waves_of_2e19_cycles_or_more"] = dict(zip(phases + ("waves", "rounds", "jobs", "windows"), list(a)[34:46]))
        return d
