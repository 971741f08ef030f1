// display_main.cpp -- the reference's frame loop (display(), main.cu:59-302) as a C++ host program over libugrt.so.
//
// What a maintainer of the reference ends up with after swapping the class bodies for integration/ugrt_shim.h:
// display() keeps its order - updateLightPosition, camera, fillCoordinatesData, build_frustum_grid, trace, per light
// { light camera, fillCoordinatesData, getEffectiveRayGridMapping, build_secondary_frustum_grid, processData,
// check_for_shadows }, simpleShade | spotlight_shade, add_shadows, writePPM.  GLUT, the PBO and the ImageMagick shell
// calls are gone; resolution and cameras come from a parameter file instead of main.cu.h / the source text.
// `streams 2` runs the same frame on two contexts and two HIP streams from this ONE host thread (light grid on the
// side stream beside the camera pass; the builds do not wait for the device: option async_build), `reflect 1` adds
// the bounce (uniform grid + 3D-DDA).
//
// `ranks N` cuts the frame into N bands of tile rows, one PROCESS per GPU (started here, before anything touches a
// GPU): every rank renders its band with the same display(), and the RGB bands are gathered on rank 0 by RCCL
// (ncclSend / ncclRecv inside one group: point-to-point fan-in, each peer on its own xGMI link) - the one
// collective of a frame.  `ranks 1` runs the same gather on a communicator of one.
// A rank that fails is an exit code, never a hang: every process runs a watchdog thread (started after the forks).
// Rank 0's reaps the children as they end - a child that ends non-zero, or a rendezvous + ncclCommInitRank that
// takes longer than `rendezvous_timeout` seconds (default 60), raises a shared flag, kills the other children and
// exits 1 from under whatever call the main thread is blocked in; a child's leaves when the flag is up or its parent
// is gone.  Every error exit of rank 0 (the *_CHECK macros call exit) raises the flag on its way out (atexit).
//
//   display_main PARAMS OUT.ppm        PARAMS: lines "key v0 v1 ..." (see read_params)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static int g_width = 1024, g_height = 1024; // main.cu.h:10-11
#define SCREEN_WIDTH g_width
#define SCREEN_HEIGHT g_height
#define FOVY 45.0f // main.cu.h:14
#define PREFIX_CAPACITY ((unsigned)(g_width * g_height / 64 + 128 * 128 + 2))
#include "ugrt_shim.h"

#define HIP_CHECK(call)                                                              \
	do {                                                                         \
		hipError_t e_ = (call);                                              \
		if (e_ != hipSuccess) {                                              \
			fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); \
			exit(-1);                                                    \
		}                                                                    \
	} while (0)

Model *model;
Camera *camera;
FrustumGrid *fGrid;
FrustumTracer *fTracer;
DecisionData *dData;
Shader *shader;
unsigned char *h_image, *d_image;
float h_light_position[3] = { 10.0f, 12.0f, 6.0f }; // per_frame_funcs.h:8-10
int frame_cnt = 0;

struct Params {
	char obj[512], mat[512];
	float cam[11], light[11]; // eye, look, up, near, far
	int streams, reflect, frames, animate_size, animate_offset, ranks, rendezvous_timeout, time_from;
	unsigned flags;
} P;

static void read_params(const char *path)
{
	FILE *fp = fopen(path, "r");
	if (!fp) {
		perror(path);
		exit(1);
	}
	char key[64];
	P.streams = 1, P.frames = 1, P.ranks = 0, P.rendezvous_timeout = 60, P.time_from = 0;
	while (fscanf(fp, "%63s", key) == 1) {
		if (!strcmp(key, "obj"))
			(void)!fscanf(fp, "%511s", P.obj);
		else if (!strcmp(key, "mat"))
			(void)!fscanf(fp, "%511s", P.mat);
		else if (!strcmp(key, "size"))
			(void)!fscanf(fp, "%d %d", &g_width, &g_height);
		else if (!strcmp(key, "camera"))
			for (int i = 0; i < 11; i++)
				(void)!fscanf(fp, "%f", &P.cam[i]);
		else if (!strcmp(key, "light_camera"))
			for (int i = 0; i < 11; i++)
				(void)!fscanf(fp, "%f", &P.light[i]);
		else if (!strcmp(key, "shading_light"))
			(void)!fscanf(fp, "%f %f %f", &h_light_position[0], &h_light_position[1], &h_light_position[2]);
		else if (!strcmp(key, "streams"))
			(void)!fscanf(fp, "%d", &P.streams);
		else if (!strcmp(key, "reflect"))
			(void)!fscanf(fp, "%d", &P.reflect);
		else if (!strcmp(key, "frames"))
			(void)!fscanf(fp, "%d", &P.frames);
		else if (!strcmp(key, "ranks"))
			(void)!fscanf(fp, "%d", &P.ranks);
		else if (!strcmp(key, "time_from"))
			(void)!fscanf(fp, "%d", &P.time_from);
		else if (!strcmp(key, "rendezvous_timeout"))
			(void)!fscanf(fp, "%d", &P.rendezvous_timeout);
		else if (!strcmp(key, "flags"))
			(void)!fscanf(fp, "%u", &P.flags);
		else {
			fprintf(stderr, "unknown key %s\n", key);
			exit(1);
		}
	}
	fclose(fp);
}

template <typename T>
static T *dev_alloc(size_t n)
{
	T *p = nullptr;
	HIP_CHECK(hipMalloc((void **)&p, (n ? n : 1) * sizeof(T)));
	return p;
}
template <typename T>
static T *dev_copy(const T *h, size_t n)
{
	T *p = dev_alloc<T>(n);
	HIP_CHECK(hipMemcpy(p, h, n * sizeof(T), hipMemcpyHostToDevice));
	return p;
}

void Model::upload() // scene.h:326-328
{
	d_vertexlist = dev_copy(ugrt_scene_vertexlist(s), (size_t)3 * num_vertices);
	d_facelist = dev_copy(ugrt_scene_facelist(s), (size_t)3 * num_faces);
	d_materiallist_index = dev_copy(ugrt_scene_materiallist_index(s), (size_t)num_faces);
	d_materiallist = dev_copy(ugrt_scene_materiallist(s), (size_t)6 * num_materials);
}

static void set_camera(const float *c)
{
	camera->setCameraCenter(c[0], c[1], c[2]);
	camera->setCameraLookAt(c[3], c[4], c[5]);
	camera->setCameraUp(c[6], c[7], c[8]);
	camera->setNearFar(c[9], c[10]);
	camera->adjustCameraAndPosition();
	camera->getGLMatrices();
	camera->getFrustumProperties();
}

// the bounce and the second stream are not in the reference: their state lives here
static ugrt_ctx *g_aux;
static hipStream_t g_main, g_side;
static hipEvent_t ev_geometry, ev_primary, ev_light_grid, ev_side_done;
static float *d_reflect, *d_rays, *d_hit_t;
static int *d_active, *d_hit_id;

void display() // main.cu:59-302
{
	frame_cnt++;
	const bool two = P.streams == 2;
	updateLightPosition();
	set_camera(P.cam); // main.cu:87-126
	HIP_CHECK(hipMemcpyAsync(camera->d_cam_position, camera->worldori, sizeof(float) * 3, hipMemcpyHostToDevice, g_main)); // :128
	fillCoordinatesData();
	ugrt_camera light_cam;
	{ // the light's camera block is needed early by the side stream; display() itself sets it up below (main.cu:158-170)
		Camera lc;
		lc.setCameraCenter(P.light[0], P.light[1], P.light[2]);
		lc.setCameraLookAt(P.light[3], P.light[4], P.light[5]);
		lc.setCameraUp(P.light[6], P.light[7], P.light[8]);
		lc.setNearFar(P.light[9], P.light[10]);
		lc.adjustCameraAndPosition();
		light_cam = lc.cam;
	}
	float bbmin[3] = { model->xMin, model->yMin, model->zMin }, bbmax[3] = { model->xMax, model->yMax, model->zMax };
	FrustumGrid lightGrid(0), uniGrid(0);
	if (two) { // side stream: the grids that only depend on the geometry, beside the camera pass
		HIP_CHECK(hipEventRecord(ev_geometry, g_main));
		HIP_CHECK(hipStreamWaitEvent(g_side, ev_geometry, 0));
		UGRT_CHECK(ugrt_upload_camera(g_aux, light_cam.camcoords));
		// (with the bounce both grids are built in one batch: their reference lists are sorted in shared launches.  Three
		// launches less; beside other frames in flight it measured 2 % slower, so the Python harness keeps it off)
		if (P.reflect)
			UGRT_CHECK(ugrt_grid_build_batch_begin(g_aux));
		UGRT_CHECK(ugrt_grid_build_spherical(g_aux, model->d_facelist, model->d_vertexlist, model->num_faces, (float)M_PI,
						     (float)M_PI));
		if (P.reflect) {
			UGRT_CHECK(ugrt_grid_build_uniform(g_aux, model->d_facelist, model->d_vertexlist, model->num_faces, bbmin, bbmax));
			UGRT_CHECK(ugrt_grid_build_batch_end(g_aux));
		}
		HIP_CHECK(hipEventRecord(ev_light_grid, g_side));
		lightGrid.fetch(g_aux, UGRT_GRID_SPHERICAL);
		if (P.reflect)
			uniGrid.fetch(g_aux, UGRT_GRID_UNIFORM);
	}
	fGrid->buildGrid(model->d_facelist, model->d_vertexlist); // build_frustum_grid, main.cu:133
	fTracer->trace(fGrid->d_triangle_value_list, fGrid->d_span, fGrid->d_offset, dData->d_primary_ray_normal,
		       dData->d_primary_ray_t_value, dData->d_primary_ray_direction, dData->d_is_shadowed, dData->d_intersect_id,
		       model->d_vertexlist, model->d_facelist); // main.cu:141
	if (two && P.reflect) { // the bounce only needs the primary hits
		HIP_CHECK(hipEventRecord(ev_primary, g_main));
		HIP_CHECK(hipStreamWaitEvent(g_side, ev_primary, 0));
		UGRT_CHECK(ugrt_reflect_rays(g_aux, camera->d_cam_position, dData->d_primary_ray_t_value, dData->d_primary_ray_direction,
					     dData->d_intersect_id, model->d_materiallist_index, d_reflect, model->num_materials,
					     model->d_vertexlist, model->d_facelist, 1e-3f, d_rays, d_active));
		UGRT_CHECK(ugrt_trace_dda(g_aux, uniGrid.d_triangle_value_list, uniGrid.d_span, uniGrid.d_offset, model->d_vertexlist,
					  model->d_facelist, d_rays, d_active, d_hit_t, d_hit_id));
	}
	if (two)
		HIP_CHECK(hipEventRecord(ev_side_done, g_side));
	for (int l = 0; l < 1; l++) { // h_numLights = 1, main.cu.h:40
		set_camera(P.light); // main.cu:158-168
		fillCoordinatesData();
		// getRayGridMapping + the host max loop (main.cu:172-185) only feed values that :186-187 overwrite
		const float x_max = (float)M_PI, y_max = (float)M_PI;
		getEffectiveRayGridMapping(dData->d_primary_ray_t_value, dData->d_primary_ray_direction, dData->d_map,
					   camera->d_cam_position, x_max, y_max);
		if (two) {
			HIP_CHECK(hipStreamWaitEvent(g_main, ev_light_grid, 0));
			*fGrid = lightGrid;
		} else {
			fGrid->buildSphericalGrid(model->d_facelist, model->d_vertexlist, x_max, y_max); // build_secondary_frustum_grid
		}
		processData();
		check_for_shadows(l);
	}
	if (two)
		HIP_CHECK(hipStreamWaitEvent(g_main, ev_side_done, 0));
	if (P.reflect) {
		if (!two) {
			UGRT_CHECK(ugrt_reflect_rays(g_ctx, camera->d_cam_position, dData->d_primary_ray_t_value,
						     dData->d_primary_ray_direction, dData->d_intersect_id, model->d_materiallist_index,
						     d_reflect, model->num_materials, model->d_vertexlist, model->d_facelist, 1e-3f, d_rays,
						     d_active));
			UGRT_CHECK(ugrt_grid_build_uniform(g_ctx, model->d_facelist, model->d_vertexlist, model->num_faces, bbmin, bbmax));
			uniGrid.fetch(g_ctx, UGRT_GRID_UNIFORM);
			UGRT_CHECK(ugrt_trace_dda(g_ctx, uniGrid.d_triangle_value_list, uniGrid.d_span, uniGrid.d_offset,
						  model->d_vertexlist, model->d_facelist, d_rays, d_active, d_hit_t, d_hit_id));
		}
		UGRT_CHECK(ugrt_shade_reflect(g_ctx, d_image, dData->d_primary_ray_normal, dData->d_primary_ray_t_value,
					      dData->d_primary_ray_direction, dData->d_intersect_id, camera->d_cam_position,
					      model->d_materiallist_index, model->d_materiallist, d_reflect, model->num_materials,
					      model->d_vertexlist, model->d_facelist, d_rays, d_active, d_hit_t, d_hit_id));
	} else if (frame_cnt < 2) { // main.cu:205-219
		shader->simpleShade(d_image, dData->d_primary_ray_normal, dData->d_primary_ray_t_value, dData->d_primary_ray_direction,
				    dData->d_intersect_id, camera->d_cam_position, model->d_materiallist_index, model->d_materiallist,
				    model->num_materials);
	} else {
		shader->spotlight_shade(d_image, dData->d_primary_ray_normal, dData->d_primary_ray_t_value,
					dData->d_primary_ray_direction, dData->d_intersect_id, camera->d_cam_position,
					model->d_materiallist_index, model->d_materiallist, model->num_materials);
	}
	shader->add_shadows(d_image, dData->d_is_shadowed); // main.cu:223
}

#define NCCL_CHECK(call)                                                              \
	do {                                                                          \
		ncclResult_t r_ = (call);                                             \
		if (r_ != ncclSuccess) {                                              \
			fprintf(stderr, "%s: %s\n", #call, ncclGetErrorString(r_)); \
			exit(-1);                                                     \
		}                                                                     \
	} while (0)

// what the ranks share before (and beside) their communicator: rank 0's RCCL id, how far the start-up has come, and
// whether any rank has failed
struct Rendezvous {
	ncclUniqueId id;
	volatile int ready;  // the id is there
	volatile int inited; // ranks whose ncclCommInitRank has returned
	volatile int failed; // a rank ended badly, or the start-up took too long: everybody leaves
	volatile int done;   // rank 0 has everything it needs from the others
};
static Rendezvous *g_rv;
static std::vector<pid_t> g_children;
static std::atomic<int> g_reaped{ 0 }, g_child_bad{ 0 };

static double now_s()
{
	timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void kill_children()
{
	for (pid_t pid : g_children)
		kill(pid, SIGKILL);
	for (pid_t pid : g_children)
		(void)waitpid(pid, nullptr, 0);
}

// rank 0 leaves through exit() on any error (the *_CHECK macros): the others must not wait for it
static void parent_atexit()
{
	if (g_rv && !g_rv->done) {
		g_rv->failed = 1;
		kill_children();
	}
}

// rank 0: reap the children as they end; a bad end or an overdue start-up ends the whole job with exit code 1
static void parent_watchdog(int ranks, double deadline)
{
	for (;;) {
		for (pid_t pid : g_children) {
			int st = 0;
			const pid_t r = waitpid(pid, &st, WNOHANG);
			if (r == pid) {
				g_reaped++;
				if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
					fprintf(stderr, "display_main: rank process %d ended with status 0x%x\n", (int)pid, st);
					g_child_bad = 1;
				}
			}
		}
		const bool late = g_rv->inited < ranks && now_s() > deadline;
		if (late)
			fprintf(stderr, "display_main: %d of %d ranks joined the communicator within the rendezvous timeout\n", g_rv->inited, ranks);
		if ((g_child_bad && !g_rv->done) || late || g_rv->failed) {
			g_rv->failed = 1;
			for (pid_t pid : g_children)
				kill(pid, SIGKILL);
			for (pid_t pid : g_children)
				(void)waitpid(pid, nullptr, 0);
			_exit(1); // (from under whatever RCCL / HIP call the main thread is blocked in)
		}
		if (g_reaped == (int)g_children.size())
			return;
		usleep(20000);
	}
}

// ranks > 0: leave when another rank has failed or rank 0 is gone
static void child_watchdog(pid_t parent, double deadline, int ranks)
{
	for (;;) {
		if (g_rv->failed || getppid() != parent || (g_rv->inited < ranks && now_s() > deadline))
			_exit(1);
		usleep(20000);
	}
}

// tile rows [begin, end) of `rank`: sizes differ by at most one row (the rule of the Python harness: parallel.band_rows)
static void band_rows(int rank, int ranks, int nby, int *begin, int *end)
{
	*begin = (int)((long long)rank * nby / ranks);
	*end = (int)((long long)(rank + 1) * nby / ranks);
}

int main(int argc, char **argv)
{
	if (argc != 3) {
		fprintf(stderr, "usage: %s PARAMS OUT.ppm\n", argv[0]);
		return 2;
	}
	read_params(argv[1]);
	const size_t N = (size_t)g_width * g_height;
	// one process per GPU: the children are started BEFORE any HIP or RCCL call of this process
	const int ranks = P.ranks > 0 ? P.ranks : 1;
	int rank = 0;
	Rendezvous *rv = nullptr;
	const pid_t parent = getpid();
	if (P.ranks > 0) {
		rv = (Rendezvous *)mmap(nullptr, sizeof(Rendezvous), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
		if (rv == MAP_FAILED) {
			perror("mmap");
			return 1;
		}
		memset(rv, 0, sizeof(*rv));
		g_rv = rv;
		for (int r = 1; r < ranks; r++) {
			const pid_t pid = fork();
			if (pid < 0) {
				perror("fork");
				rv->failed = 1;
				kill_children();
				return 1;
			}
			if (pid == 0) {
				rank = r;
				g_children.clear();
				break;
			}
			g_children.push_back(pid);
		}
		// (threads only now: the forks are behind us)
		const double deadline = now_s() + (double)P.rendezvous_timeout;
		if (rank == 0) {
			atexit(parent_atexit);
			std::thread(parent_watchdog, ranks, deadline).detach();
		} else {
			std::thread(child_watchdog, parent, deadline, ranks).detach();
		}
	}
	int row_begin = 0, row_end = g_height / 8;
	band_rows(rank, ranks, g_height / 8, &row_begin, &row_end);
	HIP_CHECK(hipSetDevice(rank));
	HIP_CHECK(hipStreamCreate(&g_main));
	ncclComm_t comm = nullptr;
	if (P.ranks > 0) {
		if (rank == 0) {
			NCCL_CHECK(ncclGetUniqueId(&rv->id));
			__sync_synchronize();
			rv->ready = 1;
		} else {
			while (!rv->ready) // (the watchdog ends this process if rank 0 fails or the deadline passes)
				usleep(1000);
			__sync_synchronize();
		}
		ncclUniqueId id = rv->id;
		NCCL_CHECK(ncclCommInitRank(&comm, ranks, id, rank));
		__sync_fetch_and_add(&rv->inited, 1);
	}
	if (P.ranks == 0) {
		init_ugrt(g_width, g_height, P.flags);
	} else { // (init_ugrt of the shim, with this rank's band of tile rows and device)
		ugrt_config cfg = { g_width, g_height, 8, 1, 128, 128, row_begin, row_end, P.flags, { 128, 128, 64 } };
		UGRT_CHECK(ugrt_ctx_create(&g_ctx, rank, &cfg));
	}
	UGRT_CHECK(ugrt_ctx_set_stream(g_ctx, g_main));
	if (P.streams == 2) {
		ugrt_config cfg = { g_width, g_height, 8, 1, 128, 128, row_begin, row_end, P.flags, { 128, 128, 64 } };
		UGRT_CHECK(ugrt_ctx_create(&g_aux, rank, &cfg));
		HIP_CHECK(hipStreamCreate(&g_side));
		UGRT_CHECK(ugrt_ctx_set_stream(g_aux, g_side));
		// builds that never wait for the device: this one host thread keeps both streams fed
		UGRT_CHECK(ugrt_ctx_set_option(g_ctx, "async_build", 1));
		UGRT_CHECK(ugrt_ctx_set_option(g_aux, "async_build", 1));
		HIP_CHECK(hipEventCreateWithFlags(&ev_geometry, hipEventDisableTiming));
		HIP_CHECK(hipEventCreateWithFlags(&ev_primary, hipEventDisableTiming));
		HIP_CHECK(hipEventCreateWithFlags(&ev_light_grid, hipEventDisableTiming));
		HIP_CHECK(hipEventCreateWithFlags(&ev_side_done, hipEventDisableTiming));
	}
	model = new Model(1); // initData(), main.cu:304-329
	model->some_material(P.mat);
	model->load_model(P.obj);
	camera = new Camera();
	camera->d_cam_position = dev_alloc<float>(3);
	fGrid = new FrustumGrid(0);
	fTracer = new FrustumTracer();
	shader = new Shader();
	dData = new DecisionData(); // decision_data.h:64-78
	dData->d_primary_ray_normal = dev_alloc<float>(3 * N);
	dData->d_primary_ray_t_value = dev_alloc<float>(N);
	dData->d_primary_ray_direction = dev_alloc<float>(3 * N);
	dData->d_is_shadowed = dev_alloc<int>(N);
	dData->d_intersect_id = dev_alloc<int>(N);
	dData->d_map = dev_alloc<unsigned>(2 * N);
	dData->d_prefixMap = dev_alloc<unsigned>(PREFIX_CAPACITY);
	dData->h_numCudaBlocks = new size_t(0);
	d_image = dev_alloc<unsigned char>(3 * N);
	h_image = (unsigned char *)malloc(3 * N);
	if (P.reflect) {
		int nm = 0;
		const float *refl = ugrt_scene_reflectlist(model->scene(), &nm);
		std::vector<float> r((size_t)model->num_materials, 0.0f);
		for (int i = 0; i < nm && i < model->num_materials; i++)
			r[i] = refl[i];
		d_reflect = dev_copy(r.data(), r.size());
		d_rays = dev_alloc<float>(6 * N);
		d_active = dev_alloc<int>(N);
		d_hit_t = dev_alloc<float>(N);
		d_hit_id = dev_alloc<int>(N);
	}
	unsigned char *d_gather = comm && rank == 0 ? dev_alloc<unsigned char>(3 * N) : nullptr;
	// `time_from F` (F > 0): the frames from frame F on are timed as a whole (the streams are drained before frame F and
	// behind the last one; the frames before are the warm-up that sizes the asynchronous builds)
	double t_begin = 0.0;
	for (int f = 0; f < P.frames; f++) {
		if (P.time_from > 0 && f == P.time_from) {
			HIP_CHECK(hipStreamSynchronize(g_main));
			if (g_side)
				HIP_CHECK(hipStreamSynchronize(g_side));
			t_begin = now_s();
		}
		display();
		if (comm) {
			// the frame's one collective: every rank's band of RGB rows to rank 0, in stream order behind the shading
			NCCL_CHECK(ncclGroupStart());
			const size_t off = (size_t)row_begin * 8 * g_width * 3, bytes = (size_t)(row_end - row_begin) * 8 * g_width * 3;
			NCCL_CHECK(ncclSend(d_image + off, bytes, ncclUint8, 0, comm, g_main));
			if (rank == 0)
				for (int r = 0; r < ranks; r++) {
					int b, e;
					band_rows(r, ranks, g_height / 8, &b, &e);
					NCCL_CHECK(ncclRecv(d_gather + (size_t)b * 8 * g_width * 3, (size_t)(e - b) * 8 * g_width * 3, ncclUint8, r, comm,
							    g_main));
				}
			NCCL_CHECK(ncclGroupEnd());
		}
	}
	double ms_per_frame = 0.0;
	if (P.time_from > 0 && P.frames > P.time_from) {
		HIP_CHECK(hipStreamSynchronize(g_main));
		if (g_side)
			HIP_CHECK(hipStreamSynchronize(g_side));
		ms_per_frame = (now_s() - t_begin) * 1e3 / (double)(P.frames - P.time_from);
	}
	if (rank == 0)
		HIP_CHECK(hipMemcpyAsync(h_image, comm ? d_gather : d_image, 3 * N, hipMemcpyDeviceToHost, g_main)); // main.cu:244
	UGRT_CHECK(ugrt_ctx_synchronize(g_ctx)); // (reports an asynchronous build that outgrew its estimate)
	if (g_aux)
		UGRT_CHECK(ugrt_ctx_synchronize(g_aux));
	if (comm)
		NCCL_CHECK(ncclCommDestroy(comm));
	if (rank != 0)
		return 0;
	// the children end by themselves now; the watchdog reaps them (and ends the job if one ends badly)
	if (rv) {
		const double until = now_s() + 60.0;
		while (g_reaped < (int)g_children.size() && !g_child_bad && now_s() < until)
			usleep(1000);
		if (g_child_bad || g_reaped < (int)g_children.size()) {
			fprintf(stderr, "a rank failed\n");
			return 1; // (atexit: the flag goes up and what is left of the children is killed)
		}
		rv->done = 1;
	}
	writePPM(argv[2]);
	printf("frames %d streams %d ranks %d chunks %zu\n", frame_cnt, P.streams, ranks, *dData->h_numCudaBlocks);
	if (ms_per_frame > 0.0)
		printf("timed_frames %d ms_per_frame %.4f\n", P.frames - P.time_from, ms_per_frame);
	return 0;
}
