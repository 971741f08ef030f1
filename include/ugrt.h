/*
 * ugrt.h -- C-ABI of the MI355X-native grid ray tracer (libugrt.so).
 *
 * Drop-in boundary for the render hot path of sushruta/uniformgrid-raytracing.
 * The reference has no FFI; its operator API is the set of C++ methods that
 * display() calls (main.cu:59-302).  Every entry point below names the
 * reference method it replaces (file:line under /root/reference) and keeps the
 * reference's argument ORDER, so a shim class with the reference's method names
 * is a one-liner.
 *
 * Conventions
 *  - plain C, pointers + sizes only; "d_" = device pointer (HIP), everything
 *    else is host memory.  The caller owns what it passes in; the context owns
 *    what it hands out (grid arrays stay valid until the next build of the
 *    same grid or ugrt_ctx_destroy).
 *  - every function returns 0 on success, a UGRT_E* code otherwise, and
 *    ugrt_last_error() describes the failure.  (The reference aborts the
 *    process instead: cutilSafeCall / exit(-1), frustum_grid.h:127-131.)
 *  - all device work is enqueued on the context's stream (ugrt_ctx_set_stream);
 *    functions that must return a host value (the grid builders' total_refs,
 *    ugrt_sort_rays' chunk count) synchronise that stream.
 *  - there is NO CPU fallback: a device entry point fails with UGRT_ENODEV
 *    when no HIP device is usable.
 *  - matrices are OpenGL column-major, indices int32/uint32, geometry fp32,
 *    image uint8 RGB, pixel id = row*width + col with row 0 at the BOTTOM of
 *    the view (trace_kernel.cu:91).
 */
#ifndef UGRT_H
#define UGRT_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UGRT_VERSION 105

enum {
	UGRT_OK = 0,
	UGRT_EINVAL = 1, /* bad argument / unsupported configuration */
	UGRT_ENODEV = 2, /* no usable HIP device */
	UGRT_EHIP = 3,   /* a HIP runtime call failed */
	UGRT_EIO = 4,    /* file could not be opened / parsed */
	UGRT_ENOMEM = 5,
	UGRT_EOVERFLOW = 6 /* option "async_build": a call's data-dependent size exceeded the room estimated from the call
			      before; returned by ugrt_ctx_synchronize, the frames since the last one are to be repeated */
};

/* flags in ugrt_config.flags */
enum {
	/* trace every shadow chunk instead of reproducing the reference's launch
	 * (block b handles chunk b-1, at most nbx*nby blocks: light_kernel.cu:76-85,
	 * per_frame_funcs.h:144) */
	UGRT_FLAG_SHADOW_ALL_CHUNKS = 1u,
	/* tracers also count their work (DDA: candidates tested, cells visited, active rays) for
	 * ugrt_stats_get; a counting context is for measurement set-up, never for timing */
	UGRT_FLAG_COUNT_WORK = 2u,
	/* the caller promises that the vertex and face arrays passed to the grid builds only change
	 * through ugrt_animate, or that it calls ugrt_geometry_changed after changing them itself
	 * (the reference's other writer is the per-frame upload of scene.h:70 frames).  The builds then
	 * keep the per-triangle records of the previous build instead of rewriting them three times a
	 * frame.  Without the flag every build assumes new geometry, as the reference does. */
	UGRT_FLAG_STATIC_GEOMETRY = 4u,
	/* ray set-up (trace_kernel.cu:96-105, per_frame_funcs.h:421-433): the reference samples its 5x5 direction
	 * texture through the texture unit's linear filter, whose interpolation weights have 8 fractional bits.  By default
	 * the fetch is the exact float bilinear interpolation at texel coordinate 4 ftx, which has the
	 * same weights k/256 at width and height 1024 and finer ones elsewhere.  With this flag the weights are quantised
	 * as the CUDA programming guide states the rule (ugrt_tex_linear8 in ugrt_fmath.h: coordinate ftx*0.8+0.1 on 5
	 * texels, alpha rounded to 8 fractional bits): identical rays at 1024 x 1024, other rays at 1920 or 3840.  What the
	 * reference's GPU really does beyond the documented rule cannot be checked here: parity unpinned either way. */
	UGRT_FLAG_STRICT_TEXTURE = 8u
};

/* which grid of the context */
enum { UGRT_GRID_PERSPECTIVE = 0, UGRT_GRID_SPHERICAL = 1, UGRT_GRID_UNIFORM = 2 };

/* stages timed by the built-in hipEvent profiler (ugrt_prof_*) */
enum {
	UGRT_ST_BUILD_COUNT = 0, /* DSKernel / DS_spherical_Kernel / uniform count */
	UGRT_ST_BUILD_SCAN,      /* cudppScan inclusive, frustum_grid.h:249 */
	UGRT_ST_BUILD_FILL,      /* DSFillkernel */
	UGRT_ST_BUILD_SORT,      /* cudppSort, frustum_grid.h:298 */
	UGRT_ST_BUILD_BOUNDS,    /* do_scan_dump .. cudppScan exclusive */
	UGRT_ST_TRACE_PRIMARY,   /* rckernel_alpha */
	UGRT_ST_MAP_RAYS,        /* mapSort_Effective_kernel */
	UGRT_ST_SORT_RAYS,       /* processData */
	UGRT_ST_TRACE_SHADOW,    /* mod_light_rckernel: the exact per-ray pass */
	UGRT_ST_SHADE,           /* lambertian_shade / spot_shade / shadow_kernel */
	UGRT_ST_REFLECT_GEN,     /* secondary ray generation (not in reference) */
	UGRT_ST_TRACE_DDA,       /* 3D-DDA traversal (not in reference) */
	UGRT_ST_ANIMATE,         /* copy_data_transform */
	UGRT_ST_WORKLIST,        /* work-item list construction for the tracers */
	UGRT_ST_SHADOW_CULL,     /* shadow tracer's (beam, triangle) cull pass */
	UGRT_ST_SHADOW_PREP,     /* shadow tracer's private re-grouping: ray keys + sort, beams, candidate sort */
	UGRT_ST_COUNT
};

typedef struct ugrt_ctx ugrt_ctx;     /* device context */
typedef struct ugrt_scene ugrt_scene; /* host scene = class Model, scene.h:13-57 */

/* main.cu.h:1-42 turned into run-time parameters */
typedef struct ugrt_config {
	int width, height;       /* SCREEN_WIDTH, SCREEN_HEIGHT; multiples of tile */
	int tile;                /* NUM_THREADS_X = NUM_THREADS_Y; must be 8 */
	int slabs;               /* NUM_SLABS (main.cu.h:18), 1..64: z-slabs of the perspective and the light grid */
	int light_nbx, light_nby; /* spherical light grid, 128 x 128 in the reference; even */
	int row_begin, row_end;  /* tile rows [begin,end) this context renders (multi-GPU band);
				    0, height/tile for the whole image */
	unsigned flags;          /* UGRT_FLAG_* */
	int uniform_dims[3];     /* cells of the uniform (reflection) grid */
} ugrt_config;

/* public members of class Camera, camera.h:19-35, plus the 64-float block of
 * fillCoordinatesData, per_frame_funcs.h:18-39 */
typedef struct ugrt_camera {
	float worldori[4];
	float modelview_matrix[16];
	float projection_matrix[16];
	float mvp_matrix[16];
	float frustum_plane_eq[6][6];
	float frustumcorner[8][3];
	float camcoords[64];
} ugrt_camera;

/* outputs of FrustumGrid, frustum_grid.h:24-29 */
typedef struct ugrt_grid_info {
	unsigned *d_triangle_value_list; /* [total_refs] triangle ids, grouped by cell */
	unsigned *d_triangle_key_list;   /* [total_refs] cell ids, ascending */
	unsigned *d_span;                /* [num_cells] triangles per cell */
	unsigned *d_offset;              /* [num_cells] exclusive scan of d_span */
	unsigned total_refs;             /* "total_triangles", frustum_grid.h:254 */
	unsigned num_cells;
	unsigned cells_used;             /* "Number of actual cells", frustum_grid.h:337; final once the
					    stream has been synchronised after the build */
} ugrt_grid_info;

/* the z-slab stage of FrustumGrid::buildGrid / buildSphericalGrid (NUM_SLABS > 1) */
typedef struct ugrt_slab_info {
	int slabs;
	float *d_proj_coord_z; /* [num_faces] d_projCoordZ: min ndc z (perspective) / min distance from the light
				  (spherical) per triangle, grid_kernel.cu:212,651; NULL when slabs == 1 */
	float z_min, z_max;    /* the host loop's zMin / zMax, frustum_grid.h:221-241 / :384-404 */
} ugrt_slab_info;

/* ---- library ---------------------------------------------------------- */
int ugrt_version(void);
const char *ugrt_last_error(void);

/* ---- host: scene I/O (scene.h, obj_parser/) --------------------------- */
/* new Model(frames), scene.h:59 */
int ugrt_scene_create(ugrt_scene **out);
/* Model::some_material(char*), scene.h:370: positional material token file */
int ugrt_scene_some_material(ugrt_scene *s, const char *file);
/* Model::load_model(char*), scene.h:141 (static branch) -> objLoader::load,
 * obj_parser/objLoader.cpp:5; the mtllib is opened relative to the cwd as in
 * obj_parser.cpp:417 unless it is found next to the .obj first */
int ugrt_scene_load_model(ugrt_scene *s, const char *path);
/* Model::tmp_model(char*, int), scene.h:70: <dir>/f_<i>.obj, vertices only */
int ugrt_scene_load_frame(ugrt_scene *s, const char *dir, int frame);
int ugrt_scene_counts(const ugrt_scene *s, int *num_vertices, int *num_faces, int *num_materials);
const float *ugrt_scene_vertexlist(const ugrt_scene *s);        /* h_vertexlist [3V] */
const int *ugrt_scene_facelist(const ugrt_scene *s);            /* h_facelist [3F] */
const int *ugrt_scene_materiallist_index(const ugrt_scene *s);  /* h_materiallist_index [F] */
const float *ugrt_scene_materiallist(const ugrt_scene *s);      /* h_materiallist [6M] Ka,Kd */
/* obj_material.reflect of the mtllib (obj_parser.h:53, token "r"), [mtl_count] */
const float *ugrt_scene_reflectlist(const ugrt_scene *s, int *mtl_count);
/* Binary cache of a loaded scene (the strtok parser takes seconds on 1 M triangles).
 * Little-endian: "UGRTSCN1", counts, then the flat lists exactly as the accessors return them.
 * load_cache replaces the scene's contents; a truncated or foreign file gives UGRT_EIO. */
int ugrt_scene_save_cache(const ugrt_scene *s, const char *path);
int ugrt_scene_load_cache(ugrt_scene *s, const char *path);
/* xMin..zMax, scene.h:43 */
int ugrt_scene_bounds(const ugrt_scene *s, float bbmin[3], float bbmax[3]);
void ugrt_scene_destroy(ugrt_scene *s);

/* ---- host: camera (camera.h) ------------------------------------------ */
/* setCameraCenter/LookAt/Up/setNearFar :13-16 + adjustCameraAndPosition :135 +
 * getGLMatrices :86 + getFrustumProperties :115 + fillCoordinatesData's block.
 * fovy in degrees (FOVY, main.cu.h:14), aspect = width/height. */
int ugrt_camera_set(ugrt_camera *cam, const float eye[3], const float look[3], const float up[3],
		    float near_plane, float far_plane, float fovy, float aspect);
/* the 5x5x4 node table of setDirectionTexture, per_frame_funcs.h:161-419 */
int ugrt_camera_direction_table(const float camcoords[64], float table[100]);

/* writePPM(char*), per_app_funcs.h:39 (returns UGRT_EIO instead of exit(1)) */
int ugrt_write_ppm(const char *path, int width, int height, const unsigned char *rgb);
/* cosf/sinf of the animation angle as the library evaluates them */
int ugrt_rot_cos_sin(float rot, float *c, float *s);

/* ---- device: context --------------------------------------------------- */
int ugrt_ctx_create(ugrt_ctx **out, int device, const ugrt_config *cfg);
int ugrt_ctx_set_stream(ugrt_ctx *ctx, void *hip_stream);
/* launch-shape options of this context; none changes a result; value < 0 restores the default.  Keys:
 * "dda_kernel" 0 = window kernel (occupancy bitmap, jobs per occupied cell, (survivor, ray) pair rounds),
 * 1 = per-ray kernel of round 1 (kept as the cross-check); "dda_rays_per_wave" 1..64 (0 = default: 32);
 * "dda_cull_min" / "dda_cull_work": a job's list is culled against its ray bundle first from this
 * many triangles / (triangles x rays) on; "dda_coop" (kernel 1) list length from which a lone ray's cell is
 * tested by the whole wave; "dda_sort" 1 = the bounce's ray list sorted by (entry cell, octant) instead of tile
 * order; "dda_split" (window kernel) 1 = the ray groups that were long in the context's last bounce are cut into
 * segments of their walk that run on different waves and are merged per ray (default; the history is kept per pixel,
 * so it serves the next frame of a moving scene as far as it goes), 0 = off, 2..4 = every group is cut (tests);
 * "dda_split_load" a group's jobs, in percent of the average group's, per segment it is cut into (default 400),
 * "dda_split_segments" segments a group is cut into at most (4; 1 = none is cut: the long groups only start first);
 * "primary_seg" triangles per primary work item; "primary_order" 0 = a flush's jobs run in list order
 * (default: nearest triangles first), "primary_chunk" jobs between two looks at the rays' closest hits;
 * "shadow_beam", "shadow_xseg", "shadow_sizebits", "shadow_itemsort", "shadow_mbits", "shadow_key64", "shadow_sieve" shape the
 * shadow tracer's private regrouping; "sort_library" 1 = rocPRIM's radix sort instead of the built-in
 * one, "sort_items" 8 / 16 pairs per thread of a radix pass (default: by size), "sort_rank" 0 = the passes rank by
 * ballots instead of LDS atomics, "ray_sort" 1 = the deferred ugrt_sort_rays sorts at once also where nothing needs it
 * (see there); "dda_blocks", "primary_waves",
 * "shadow_waves": number of persistent single-wave workgroups of the bounce, the primary tracer and the two
 * shadow kernels (the primary tracer and the exact shadow pass run one wave per work item by default, "primary_xcd_run" /
 * "shadow_xcd_run" neighbouring items per XCD in turn; "primary_waves" set / "shadow_xcd_run" 0 restore their persistent
 * forms; "shadow_waves" is always the cull pass's; "primary_centre" 0 = the primary tracer's runs of items start in
 * list order instead of from the middle of the list outwards).
 * "async_build" 1: the grid builds and ugrt_trace_shadow stop waiting for the device.  The reference reads
 * total_triangles back to size its lists (frustum_grid.h:254); here the second and later builds of a grid size
 * buffers and launches by what the build before needed plus a quarter, every kernel takes the real counts from
 * device memory, and a count that does not fit raises a flag instead of writing out of bounds:
 * ugrt_ctx_synchronize then returns UGRT_EOVERFLOW once (the frames since the last synchronisation are
 * incomplete) and the next calls run in the waiting form again, which sizes everything exactly.
 * ugrt_grid_info.total_refs of such a build is final once the stream has been synchronised. */
int ugrt_ctx_set_option(ugrt_ctx *ctx, const char *key, int value);
/* counters and findings of this context (no reference counterpart; the bench line and the tests read them):
 * "radix_launches" histogram + pass kernels of the built-in radix sort enqueued so far, "sort_rank_atomic" 1 = the radix passes rank by LDS
 * atomics (the context's self-test found them served in lane order on this device), 0 = by ballots, -1 = no sort
 * has run yet; "recip_mismatches" runs every float bit pattern through the tracers' short reciprocal on the device and
 * returns the number of operands whose result differs from 1.0f / x (0), "lane_reduce_mismatches" compares the tracers'
 * DPP / permlane-swap reductions with the same reductions by __shfl_xor (0): both wait for the stream.
 * Unknown key: UGRT_EINVAL. */
int ugrt_ctx_get_state(ugrt_ctx *ctx, const char *key, long long *value);
int ugrt_ctx_synchronize(ugrt_ctx *ctx);
void ugrt_ctx_destroy(ugrt_ctx *ctx);

/* fillCoordinatesData(), per_frame_funcs.h:18: makes `camcoords` the current
 * camera block (dd_camcoords) and rebuilds the direction table (texdir) */
int ugrt_upload_camera(ugrt_ctx *ctx, const float camcoords[64]);
/* updateLightPosition(), per_frame_funcs.h:6: dd_light_position */
int ugrt_set_light_position(ugrt_ctx *ctx, const float pos[3]);

/* ---- device: grid build (frustum_grid.h) ------------------------------- */
/* FrustumGrid::buildGrid(int*, float*), frustum_grid.h:210 */
int ugrt_grid_build_perspective(ugrt_ctx *ctx, const int *d_facelist, const float *d_vertlist, int num_faces);
/* FrustumGrid::buildSphericalGrid(int*, float*, float, float), frustum_grid.h:368 */
int ugrt_grid_build_spherical(ugrt_ctx *ctx, const int *d_facelist, const float *d_vertlist, int num_faces,
			      float xM, float yM);
/* uniform world-space grid for the reflection bounce (README.md:1; no reference code) */
int ugrt_grid_build_uniform(ugrt_ctx *ctx, const int *d_facelist, const float *d_vertlist, int num_faces,
			    const float bbmin[3], const float bbmax[3]);
int ugrt_grid_get_info(ugrt_ctx *ctx, int which, ugrt_grid_info *out);
/* With slabs > 1 the arrays of ugrt_grid_info hold num_cells = cells * slabs entries, cell-major
 * (key = cell * slabs + slab, grid_kernel.cu:322).  This call synchronises the stream. */
int ugrt_grid_get_slabs(ugrt_ctx *ctx, int which, ugrt_slab_info *out);
/* Sharded build of the light grid and the uniform grid across the GPUs of a node (not in the reference, which
 * has one GPU).  Each rank restricts its builds to a window [begin, end) of the triangle list
 * (end < 0: to the last triangle, so 0, -1 restores the full build; begin == end: an empty shard), the ranks exchange the arrays of their shards
 * (ugrt_grid_get_info: keys, values, span; total_refs entries) and every rank merges them: part r must hold the
 * window of rank r, windows ascending with r and disjoint.  The merged arrays become the context's grid and are
 * element for element those of a full build.  The parts are the caller's buffers, not the context's arrays. */
int ugrt_ctx_set_face_window(ugrt_ctx *ctx, int begin, int end);
int ugrt_grid_merge_shards(ugrt_ctx *ctx, int which, int nparts, const unsigned *const *d_keys,
			   const unsigned *const *d_vals, const unsigned *const *d_span, const unsigned *counts);
/* Two grid builds that depend on the geometry only (a frame's light grid and uniform grid) in shared sort launches:
 * between _begin and _end each ugrt_grid_build_* call of this context enqueues its count, scan and fill and returns;
 * _end sorts the (at most two) reference lists together -- one histogram kernel and one kernel per pass level for
 * both -- and completes the builds.  Results are those of two separate builds.  Only builds in the asynchronous form
 * ("async_build") are deferred; any other is built at once.  The grids of the batch must not be used (traced,
 * queried, waited for by another stream) before _end has returned.  No reference counterpart. */
int ugrt_grid_build_batch_begin(ugrt_ctx *ctx);
int ugrt_grid_build_batch_end(ugrt_ctx *ctx);
/* with UGRT_FLAG_STATIC_GEOMETRY: the vertex or face array was rewritten by the caller */
int ugrt_geometry_changed(ugrt_ctx *ctx);
/* cudppSort(plan, keys, values, bits, n) with CUDPP_SORT_RADIX on (uint key, uint value) pairs
 * (cudpp/cudpp.h:426-471; call sites frustum_grid.h:298, decision_data.h:177): the stable radix sort
 * the builds and the ray sort run on, exposed for callers that drive the stages themselves and for
 * the tests.  Sorts on key bits [0, key_bits); inputs are left untouched; all pointers are device
 * memory, outputs must not alias inputs.  use_library != 0 runs rocPRIM's radix sort instead of the
 * built-in one (same result). */
int ugrt_sort_pairs(ugrt_ctx *ctx, const unsigned *d_keys_in, unsigned *d_keys_out, const unsigned *d_values_in,
		    unsigned *d_values_out, size_t n, int key_bits, int use_library);

/* ---- device: tracing --------------------------------------------------- */
/* FrustumTracer::trace(...), frustum_tracer.h:20-23 -> rckernel_alpha */
int ugrt_trace_primary(ugrt_ctx *ctx, const unsigned *d_value_list, const unsigned *d_span,
		       const unsigned *d_offset, float *d_normal, float *d_t_value, float *d_ray_dir,
		       int *d_shadowed, int *d_intersect_id, const float *d_vertlist, const int *d_trilist);
/* getEffectiveRayGridMapping(...), per_frame_funcs.h:97 -> mapSort_Effective_kernel.
 * d_map holds 2n entries, n = pixels of this context's band. */
int ugrt_map_rays_to_light(ugrt_ctx *ctx, const float *d_t_value, const float *d_ray_dir, unsigned *d_map,
			   const float *d_cam_position, float xM, float yM);
/* processData(), per_frame_funcs.h:116: sorts d_map by light cell and writes the
 * chunk start indices; *num_chunks = h_numCudaBlocks (decision_data.h:264).
 * prefix_capacity entries must fit: n/64 + light cells + 1 always does (a smaller map makes the call fail with
 * UGRT_EINVAL; in the deferred form below the shadow tracer then traces nothing and ugrt_sort_rays_chunks reports
 * the error). */
int ugrt_sort_rays(ugrt_ctx *ctx, unsigned *d_map, unsigned *d_prefix_map, unsigned prefix_capacity,
		   unsigned *num_chunks);
/* num_chunks may be NULL: the call then does not wait for the device; the count is passed on to
 * ugrt_trace_shadow as UGRT_CHUNKS_ON_DEVICE and can be fetched later (ugrt_sort_rays_chunks waits for the stream).
 * In a context with UGRT_FLAG_SHADOW_ALL_CHUNKS this deferred form also puts off the SORT: the chunk list only decides
 * which rays the reference's launch traces, with the flag that is every ray, and ugrt_trace_shadow reads the
 * (pixel, light cell) pairs of d_map in any order.  d_map and d_prefix_map are then left as they are until
 * ugrt_sort_rays_chunks is called, which sorts them (they must still hold what ugrt_map_rays_to_light wrote) and
 * returns the count: processData's outputs on demand, two radix passes and five launches less in a frame that never
 * asks.  Option "ray_sort" 1 sorts at once as before. */
int ugrt_sort_rays_chunks(ugrt_ctx *ctx, unsigned *num_chunks);
#define UGRT_CHUNKS_ON_DEVICE 0xFFFFFFFFu
/* check_for_shadows(int), per_frame_funcs.h:139 -> mod_light_rckernel; same
 * argument order as the kernel (light_kernel.cu:53) */
int ugrt_trace_shadow(ugrt_ctx *ctx, const unsigned *d_value_list, const float *d_vertlist,
		      const int *d_trilist, const unsigned *d_span, const unsigned *d_offset,
		      const float *d_t_value, const float *d_ray_dir, int *d_is_shadowed, const unsigned *d_map,
		      const unsigned *d_prefix_map, const float *d_cam_position, unsigned num_chunks);

/* ---- device: shading (shader.h:20-24) ---------------------------------- */
int ugrt_shade_simple(ugrt_ctx *ctx, unsigned char *d_img, const float *d_normal, const float *d_t_value,
		      const float *d_ray_dir, int *d_intersect_id, const float *d_cam_position,
		      const int *d_mat_idx, const float *d_mat_list, int num_materials);
int ugrt_shade_spotlight(ugrt_ctx *ctx, unsigned char *d_img, const float *d_normal, const float *d_t_value,
			 const float *d_ray_dir, int *d_intersect_id, const float *d_cam_position,
			 const int *d_mat_idx, const float *d_mat_list, int num_materials, float *d_dump);
int ugrt_shade_add_shadows(ugrt_ctx *ctx, unsigned char *d_img, const int *d_is_shadowed);
int ugrt_shade_perlin(ugrt_ctx *ctx, unsigned char *d_img, const float *d_t_value, const float *d_ray_dir,
		      const float *d_cam_position, const int *d_intersect_id);

/* ---- device: reflection bounce (not in the reference) ---- */
/* secondary rays for hit pixels whose material has reflect > 0 */
int ugrt_reflect_rays(ugrt_ctx *ctx, const float *d_cam_position, const float *d_t_value,
		      const float *d_ray_dir, const int *d_intersect_id, const int *d_mat_idx,
		      const float *d_reflect, int num_materials, const float *d_vertlist, const int *d_trilist,
		      float eps, float *d_rays, int *d_active);
/* Amanatides-Woo traversal of the context's uniform grid */
int ugrt_trace_dda(ugrt_ctx *ctx, const unsigned *d_value_list, const unsigned *d_span,
		   const unsigned *d_offset, const float *d_vertlist, const int *d_trilist,
		   const float *d_rays, const int *d_active, float *d_hit_t, int *d_hit_id);
/* lambertian_shade + blend (1-k)*local + k*reflected */
int ugrt_shade_reflect(ugrt_ctx *ctx, unsigned char *d_img, const float *d_normal, const float *d_t_value,
		       const float *d_ray_dir, int *d_intersect_id, const float *d_cam_position,
		       const int *d_mat_idx, const float *d_mat_list, const float *d_reflect, int num_materials,
		       const float *d_vertlist, const int *d_trilist, const float *d_rays, const int *d_active,
		       const float *d_hit_t, const int *d_hit_id);

/* ---- device: animation (scene.h:122,336) -------------------------------- */
/* Model::rotate_bunny(float) -> copy_data_transform, transformation_kernel.cu:4 */
int ugrt_animate(ugrt_ctx *ctx, float *d_vertlist, const float *d_orig_list, int size, int offset,
		 float rot_factor);

/* ---- profiling ---------------------------------------------------------- */
/* hipEvent pairs around stages, on the context's stream.  on = 0: off; 1: every stage; otherwise a
 * mask in which bit (s + 1) selects stage UGRT_ST_s */
int ugrt_prof_enable(ugrt_ctx *ctx, int on);
int ugrt_prof_reset(ugrt_ctx *ctx);
/* total milliseconds and number of timed launches of a stage since the reset
 * (synchronises the stream) */
int ugrt_prof_get(ugrt_ctx *ctx, int stage, double *ms_total, int *launches);
/* counters of the last tracer launches (synchronises the stream): [0] primary work-item
 * capacity, [1] shadow beams, [2] shadow chunks traced, [6] shadow cull pass: (triangle, beam)
 * tests, [7] shadow exact pass: candidates staged (candidate pairs x 64-ray sub-groups);
 * with UGRT_FLAG_COUNT_WORK also [3] DDA candidates tested, [4] DDA cells visited, [5] DDA active rays */
int ugrt_stats_get(ugrt_ctx *ctx, unsigned long long stats[8]);
/* how the beam kernel of ugrt_trace_dda shared its work in the last counting launch (UGRT_FLAG_COUNT_WORK):
 * [0] wave iterations (4 steps each), [1] cell groups processed, [2] rays in those groups, [3] cull batches,
 * [4] triangles culled against a ray bundle, [5] exact-test rounds (one broadcast triangle), [6] rays that ran
 * those rounds, [7] exact tests of lone rays, [8..23] waves by log2(shader cycles / 4096), [24] sum and
 * [25] maximum of the waves' cycles; entries past n are not written, entries past 25 are 0 */
int ugrt_stats_dda(ugrt_ctx *ctx, unsigned long long *stats, int n);
/* split walks of the context's last ugrt_trace_dda (option "dda_split"; synchronises the stream): [0] segments the
 * cut ray groups were listed as, [1] jobs of the launch before, [2] cut groups some of whose rays had to be walked
 * again in one piece, [3] those rays */
int ugrt_stats_dda_split(ugrt_ctx *ctx, unsigned out[4]);
/* The same for the primary tracer: [0] work items (tile x list segment), [1] batches of 64 references culled,
 * [2] batches with a survivor of the tile's box, [3] references, [4] survivors of the tile's box, [5] survivors of a
 * quadrant's box (staged in LDS), [6] jobs (survivor x quadrant), [7] flushes, [8] exact rounds (4 jobs x 16 rays),
 * [9] rounds that reach the division, [10] the v test, [11] t, [12] (ray, triangle) tests, [13] accepted hits */
int ugrt_stats_primary(ugrt_ctx *ctx, unsigned long long *stats, int n);

#ifdef __cplusplus
}
#endif
#endif /* UGRT_H */
