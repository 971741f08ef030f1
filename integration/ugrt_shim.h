// ugrt_shim.h -- drop-in replacements for the reference's host classes (same names, same method signatures),
// every body one call into libugrt.so.  The including program defines, before the #include, what main.cu.h
// defines - SCREEN_WIDTH, SCREEN_HEIGHT, FOVY - and PREFIX_CAPACITY >= SCREEN_WIDTH*SCREEN_HEIGHT/64 + 128*128 + 2,
// and, anywhere, the reference's globals model, camera, fGrid, dData, h_image, h_light_position.
#ifndef UGRT_SHIM_H
#define UGRT_SHIM_H
#include "ugrt.h"
#include <cstdio>
#include <cstdlib>

static ugrt_ctx *g_ctx; // replaces dd_camcoords / dd_light_position / texdir (main.cu.h:58-63)
// the reference aborts on every error (cutilSafeCall, frustum_grid.h:127-131)
#define UGRT_CHECK(call)                                        \
	do {                                                    \
		if (call) {                                     \
			fprintf(stderr, "%s\n", ugrt_last_error()); \
			exit(-1);                               \
		}                                               \
	} while (0)

// once, after initData() (main.cu:318)
static void init_ugrt(int W, int H, unsigned flags)
{
	ugrt_config cfg = { W, H, 8, 1, 128, 128, 0, H / 8, flags, { 128, 128, 64 } };
	UGRT_CHECK(ugrt_ctx_create(&g_ctx, 0, &cfg));
}

class Model { // scene.h:13-57
	ugrt_scene *s;

public:
	int *d_facelist;
	float *d_vertexlist, *d_materiallist;
	int *d_materiallist_index;
	float *d_vertex_orig_list;
	int size_orig_list, offset_orig_list;
	int num_faces, num_vertices, num_materials;
	float xMin, xMax, yMin, yMax, zMin, zMax;
	Model(int) : d_vertex_orig_list(nullptr), size_orig_list(0), offset_orig_list(0) { UGRT_CHECK(ugrt_scene_create(&s)); }
	void some_material(char *f) { UGRT_CHECK(ugrt_scene_some_material(s, f)); } // scene.h:370
	void load_model(char *p)                                                     // scene.h:141
	{
		UGRT_CHECK(ugrt_scene_load_model(s, p));
		ugrt_scene_counts(s, &num_vertices, &num_faces, &num_materials);
		float mn[3], mx[3];
		ugrt_scene_bounds(s, mn, mx);
		xMin = mn[0], yMin = mn[1], zMin = mn[2], xMax = mx[0], yMax = mx[1], zMax = mx[2];
		upload(); // the device copies of scene.h:326-328
	}
	void upload(); // hipMalloc + hipMemcpy of ugrt_scene_vertexlist / facelist / materiallist_index / materiallist
	const ugrt_scene *scene() const { return s; }
	void rotate_bunny(float rot) // scene.h:122
	{
		UGRT_CHECK(ugrt_animate(g_ctx, d_vertexlist, d_vertex_orig_list, size_orig_list, offset_orig_list, rot));
	}
};

class Camera { // camera.h:7-47 (no GL state machine any more)
	float c[3], l[3], u[3];

public:
	ugrt_camera cam;
	float nearPlane, farPlane;
	float *d_cam_position;
	float *worldori, *modelview_matrix, *projection_matrix, *mvp_matrix;
	Camera() : worldori(cam.worldori), modelview_matrix(cam.modelview_matrix), projection_matrix(cam.projection_matrix),
		   mvp_matrix(cam.mvp_matrix) {}
	void setCameraCenter(float x, float y, float z) { c[0] = x, c[1] = y, c[2] = z; }
	void setCameraLookAt(float x, float y, float z) { l[0] = x, l[1] = y, l[2] = z; }
	void setCameraUp(float x, float y, float z) { u[0] = x, u[1] = y, u[2] = z; }
	void setNearFar(float n, float f) { nearPlane = n, farPlane = f; }
	void adjustCameraAndPosition() // + getGLMatrices + getFrustumProperties, camera.h:86-253
	{
		UGRT_CHECK(ugrt_camera_set(&cam, c, l, u, nearPlane, farPlane, FOVY, (float)SCREEN_WIDTH / (float)SCREEN_HEIGHT));
	}
	void getGLMatrices() {}
	void getFrustumProperties() {}
};

class FrustumGrid { // frustum_grid.h:15-66
public:
	unsigned int *d_offset, *d_span, *d_triangle_value_list, *d_triangle_key_list;
	FrustumGrid(int) {}
	void fetch(ugrt_ctx *ctx, int which)
	{
		ugrt_grid_info gi;
		UGRT_CHECK(ugrt_grid_get_info(ctx, which, &gi));
		d_offset = gi.d_offset, d_span = gi.d_span;
		d_triangle_value_list = gi.d_triangle_value_list, d_triangle_key_list = gi.d_triangle_key_list;
	}
	void buildGrid(int *f, float *v); // :210
	void buildSphericalGrid(int *f, float *v, float xM, float yM); // :368
};

class FrustumTracer { // frustum_tracer.h:14-33
public:
	void trace(unsigned int *value, unsigned int *span, unsigned int *offset, float *normal, float *t, float *dir,
		   int *shadowed, int *id, float *verts, int *tris)
	{
		UGRT_CHECK(ugrt_trace_primary(g_ctx, value, span, offset, normal, t, dir, shadowed, id, verts, tris));
	}
};

class Shader { // shader.h:14-31
public:
	void simpleShade(unsigned char *img, float *n, float *t, float *dir, int *id, float *cam, int *mi, float *ml, int nm)
	{
		UGRT_CHECK(ugrt_shade_simple(g_ctx, img, n, t, dir, id, cam, mi, ml, nm));
	}
	void spotlight_shade(unsigned char *img, float *n, float *t, float *dir, int *id, float *cam, int *mi, float *ml,
			     int nm)
	{
		UGRT_CHECK(ugrt_shade_spotlight(g_ctx, img, n, t, dir, id, cam, mi, ml, nm, nullptr));
	}
	void add_shadows(unsigned char *img, int *sh) { UGRT_CHECK(ugrt_shade_add_shadows(g_ctx, img, sh)); }
	void perlinShade(unsigned char *img, float *t, float *dir, float *cam, int *id)
	{
		UGRT_CHECK(ugrt_shade_perlin(g_ctx, img, t, dir, cam, id));
	}
};

struct DecisionData { // decision_data.h:13-40: the per-pixel lists display() passes around (the scan/compact
		      // scratch arrays of processData are gone)
	float *d_primary_ray_normal, *d_primary_ray_t_value, *d_primary_ray_direction;
	int *d_is_shadowed, *d_intersect_id;
	unsigned int *d_map, *d_prefixMap;
	size_t *h_numCudaBlocks;
};

// the reference's globals (main.cu.h:64-80, main.cu:20-57), defined by the host program
extern Model *model;
extern Camera *camera;
extern FrustumGrid *fGrid;
extern DecisionData *dData;
extern unsigned char *h_image;
extern float h_light_position[3];

inline void FrustumGrid::buildGrid(int *f, float *v)
{
	UGRT_CHECK(ugrt_grid_build_perspective(g_ctx, f, v, model->num_faces));
	fetch(g_ctx, UGRT_GRID_PERSPECTIVE);
}
inline void FrustumGrid::buildSphericalGrid(int *f, float *v, float xM, float yM)
{
	UGRT_CHECK(ugrt_grid_build_spherical(g_ctx, f, v, model->num_faces, xM, yM));
	fetch(g_ctx, UGRT_GRID_SPHERICAL);
}
static void fillCoordinatesData() { UGRT_CHECK(ugrt_upload_camera(g_ctx, camera->cam.camcoords)); } // per_frame_funcs.h:18
static void updateLightPosition()                                                                  // per_frame_funcs.h:6
{
	UGRT_CHECK(ugrt_set_light_position(g_ctx, h_light_position));
}
static void getEffectiveRayGridMapping(float *t, float *dir, unsigned int *map, float *cam, float xM, float yM)
{ // per_frame_funcs.h:97
	UGRT_CHECK(ugrt_map_rays_to_light(g_ctx, t, dir, map, cam, xM, yM));
}
static void processData() // per_frame_funcs.h:116
{
	unsigned n;
	// d_prefixMap must hold PREFIX_CAPACITY entries (the reference allocates too few: decision_data.h:78)
	UGRT_CHECK(ugrt_sort_rays(g_ctx, dData->d_map, dData->d_prefixMap, PREFIX_CAPACITY, &n));
	*dData->h_numCudaBlocks = n;
}
static void check_for_shadows(int) // per_frame_funcs.h:139
{
	UGRT_CHECK(ugrt_trace_shadow(g_ctx, fGrid->d_triangle_value_list, model->d_vertexlist, model->d_facelist, fGrid->d_span,
				     fGrid->d_offset, dData->d_primary_ray_t_value, dData->d_primary_ray_direction,
				     dData->d_is_shadowed, dData->d_map, dData->d_prefixMap, camera->d_cam_position,
				     (unsigned)*dData->h_numCudaBlocks));
}
static void writePPM(char *f) // per_app_funcs.h:39
{
	if (ugrt_write_ppm(f, SCREEN_WIDTH, SCREEN_HEIGHT, h_image)) {
		perror("fopen");
		exit(1);
	}
}
#endif
