// ugrt_host.cpp -- host side of libugrt.so: scene I/O, camera set-up, PPM.
//
// Mirrors the reference's host classes for the render path:
//   class Model      scene.h:13-57      -> ugrt_scene_*
//   class objLoader  obj_parser/*       -> ObjFile (own parser, same grammar)
//   class Camera     camera.h:7-47      -> ugrt_camera_set
//   writePPM         per_app_funcs.h:39 -> ugrt_write_ppm
// No GPU is needed for anything in this file.
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ugrt_internal.h"

// ---------------------------------------------------------------------------
// error string
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int ugrt_fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return code;
}

extern "C" const char *ugrt_last_error(void) { return g_err; }
extern "C" int ugrt_version(void) { return UGRT_VERSION; }

// ---------------------------------------------------------------------------
// Wavefront OBJ / MTL.  Grammar and quirks follow obj_parser/obj_parser.cpp:
//  - lines are cut at 499 characters (fgets into OBJ_LINE_SIZE, :305,331)
//  - tokens split on " \t\n\r"; a line whose first token starts with '#' is a
//    comment (:336)
//  - v/vn/vt: three atof() values (:163-170)
//  - f: up to 4 "v", "v/t", "v//n", "v/t/n" tokens, atoi() per field (:49-84);
//    index 0 -> -1, negative -> relative to the vertices read so far, else
//    1-based -> 0-based (:16-26)
//  - usemtl: the FIRST material whose stored name starts with the token
//    (list_find is a strncmp prefix match, list.cpp:115-126), else -1
//  - mtllib: parsed immediately (:415-420)
//  - sp/pl/p/lp/ld/lq/c/o/s/g are accepted and ignored by the renderer
// MTL (:182-296): newmtl opens a material with the defaults of :28-45; names
// are split on " \t" only, so they keep the line's trailing newline; Ka is
// split on " " only; Kd Ks Ns d r sharpness Ni map_Ka as usual.
// ---------------------------------------------------------------------------
namespace {

const char *WS = " \t\n\r";

struct Material {
	std::string name;
	double amb[3], diff[3], spec[3];
	double reflect, trans, shiny, glossy, refract_index;
	std::string texture;
};

struct ObjFile {
	std::vector<double> v;  // 3 per vertex
	size_t nvn = 0, nvt = 0;
	std::vector<int> fidx;  // 4 per face (vertex indices, converted)
	std::vector<int> fmat;  // material per face
	std::vector<Material> mtl;
};

int convert_index(int current_max, int index)
{
	if (index == 0)
		return -1;
	if (index < 0)
		return current_max + index;
	return index - 1;
}

// (strtok_r: the parsers keep their position in a local, so scene loads are safe beside other threads, and the
// MTL parser can run inside the OBJ parser's line loop)
double tok_atof(const char *delim, char **save)
{
	char *t = strtok_r(nullptr, delim, save);
	return t ? atof(t) : 0.0;
}

bool parse_mtl(const std::string &path, std::vector<Material> &out)
{
	FILE *fp = fopen(path.c_str(), "r");
	if (!fp)
		return false;
	char line[500], *save = nullptr;
	bool open = false;
	while (fgets(line, sizeof(line), fp)) {
		char *tok = strtok_r(line, WS, &save);
		if (!tok || !strcmp(tok, "//") || !strcmp(tok, "#"))
			continue;
		if (!strcmp(tok, "newmtl")) {
			Material m;
			m.amb[0] = m.amb[1] = m.amb[2] = 0.2;
			m.diff[0] = m.diff[1] = m.diff[2] = 0.8;
			m.spec[0] = m.spec[1] = m.spec[2] = 1.0;
			m.reflect = 0.0;
			m.trans = 1;
			m.glossy = 98;
			m.shiny = 0;
			m.refract_index = 1;
			char *nm = strtok_r(nullptr, " \t", &save);
			m.name = nm ? nm : "";
			out.push_back(m);
			open = true;
		} else if (!open) {
			continue;
		} else if (!strcmp(tok, "Ka")) {
			Material &m = out.back();
			m.amb[0] = tok_atof(" ", &save);
			m.amb[1] = tok_atof(" ", &save);
			m.amb[2] = tok_atof(" ", &save);
		} else if (!strcmp(tok, "Kd")) {
			Material &m = out.back();
			m.diff[0] = tok_atof(" \t", &save);
			m.diff[1] = tok_atof(" \t", &save);
			m.diff[2] = tok_atof(" \t", &save);
		} else if (!strcmp(tok, "Ks")) {
			Material &m = out.back();
			m.spec[0] = tok_atof(" \t", &save);
			m.spec[1] = tok_atof(" \t", &save);
			m.spec[2] = tok_atof(" \t", &save);
		} else if (!strcmp(tok, "Ns")) {
			out.back().shiny = tok_atof(" \t", &save);
		} else if (!strcmp(tok, "d")) {
			out.back().trans = tok_atof(" \t", &save);
		} else if (!strcmp(tok, "r")) {
			out.back().reflect = tok_atof(" \t", &save);
		} else if (!strcmp(tok, "sharpness")) {
			out.back().glossy = tok_atof(" \t", &save);
		} else if (!strcmp(tok, "Ni")) {
			out.back().refract_index = tok_atof(" \t", &save);
		} else if (!strcmp(tok, "map_Ka")) {
			char *t = strtok_r(nullptr, " \t", &save);
			out.back().texture = t ? t : "";
		}
	}
	fclose(fp);
	return true;
}

int find_material(const std::vector<Material> &mtl, const char *name)
{
	if (!name)
		return -1;
	size_t n = strlen(name);
	for (size_t i = 0; i < mtl.size(); i++)
		if (strncmp(mtl[i].name.c_str(), name, n) == 0)
			return (int)i;
	return -1;
}

bool parse_obj(const char *path, ObjFile &o)
{
	FILE *fp = fopen(path, "r");
	if (!fp)
		return false;
	std::string dir(path);
	size_t slash = dir.find_last_of('/');
	dir = (slash == std::string::npos) ? std::string() : dir.substr(0, slash + 1);
	char line[500], *save = nullptr;
	int current_material = -1;
	while (fgets(line, sizeof(line), fp)) {
		char *tok = strtok_r(line, WS, &save);
		if (!tok || tok[0] == '#')
			continue;
		if (!strcmp(tok, "v")) {
			double x = tok_atof(WS, &save), y = tok_atof(WS, &save), z = tok_atof(WS, &save);
			o.v.push_back(x);
			o.v.push_back(y);
			o.v.push_back(z);
		} else if (!strcmp(tok, "vn")) {
			o.nvn++;
		} else if (!strcmp(tok, "vt")) {
			o.nvt++;
		} else if (!strcmp(tok, "f")) {
			int idx[4] = { 0, 0, 0, 0 };
			int cnt = 0;
			char *t;
			while ((t = strtok_r(nullptr, WS, &save)) != nullptr) {
				if (cnt < 4)
					idx[cnt] = atoi(t);
				cnt++;
			}
			int nv = (int)(o.v.size() / 3);
			for (int k = 0; k < 4; k++)
				o.fidx.push_back(convert_index(nv, idx[k]));
			o.fmat.push_back(current_material);
		} else if (!strcmp(tok, "usemtl")) {
			current_material = find_material(o.mtl, strtok_r(nullptr, WS, &save));
		} else if (!strcmp(tok, "mtllib")) {
			char *fn = strtok_r(nullptr, WS, &save);
			if (fn) {
				// next to the .obj first, then relative to the cwd (reference behaviour)
				if (!parse_mtl(dir + fn, o.mtl))
					parse_mtl(fn, o.mtl);
			}
		}
	}
	fclose(fp);
	return true;
}

} // namespace

struct ugrt_scene {
	std::vector<float> vertexlist;  // h_vertexlist
	std::vector<int> facelist;      // h_facelist
	std::vector<int> matidx;        // h_materiallist_index
	std::vector<float> materiallist; // h_materiallist (6 per material)
	std::vector<float> reflect;     // obj_material.reflect
	int num_materials = 0;
	float bbmin[3] = { 9999.9f, 9999.9f, 9999.9f };
	float bbmax[3] = { -9999.9f, -9999.9f, -9999.9f };
};

extern "C" int ugrt_scene_create(ugrt_scene **out)
{
	if (!out)
		return ugrt_fail(UGRT_EINVAL, "ugrt_scene_create: null out");
	*out = new ugrt_scene();
	return UGRT_OK;
}

extern "C" void ugrt_scene_destroy(ugrt_scene *s) { delete s; }

// scene.h:370-439
extern "C" int ugrt_scene_some_material(ugrt_scene *s, const char *file)
{
	if (!s || !file)
		return ugrt_fail(UGRT_EINVAL, "some_material: null argument");
	FILE *fp = fopen(file, "r");
	if (!fp)
		return ugrt_fail(UGRT_EIO, "some_material: cannot open %s: %s", file, strerror(errno));
	char cjunk[512];
	int num = 0;
	while (fscanf(fp, "%511s", cjunk) != EOF)
		if (strcmp(cjunk, "newmtl") == 0)
			num++;
	fclose(fp);
	s->num_materials = num;
	s->materiallist.assign((size_t)num * 6, 0.0f);
	fp = fopen(file, "r");
	if (!fp)
		return ugrt_fail(UGRT_EIO, "some_material: cannot reopen %s", file);
	bool ok = true;
	for (int mt = 0; mt < num && ok; mt++) {
		for (int i = 0; i < 3 && ok; i++)
			ok = fscanf(fp, "%511s", cjunk) == 1;
		for (int i = 0; i < 3 && ok; i++)
			ok = fscanf(fp, "%f", &s->materiallist[mt * 6 + i]) == 1;
		ok = ok && fscanf(fp, "%511s", cjunk) == 1;
		for (int i = 0; i < 3 && ok; i++)
			ok = fscanf(fp, "%f", &s->materiallist[mt * 6 + 3 + i]) == 1;
		for (int i = 0; i < 12 && ok; i++)
			ok = fscanf(fp, "%511s", cjunk) == 1;
	}
	fclose(fp);
	if (!ok)
		return ugrt_fail(UGRT_EIO, "some_material: %s does not follow the material token layout", file);
	return UGRT_OK;
}

static void scene_take_vertices(ugrt_scene *s, const ObjFile &o)
{
	size_t nv = o.v.size() / 3;
	s->vertexlist.resize(nv * 3);
	for (int k = 0; k < 3; k++) {
		s->bbmin[k] = 9999.9f;
		s->bbmax[k] = -9999.9f;
	}
	for (size_t v = 0; v < nv; v++)
		for (int k = 0; k < 3; k++) {
			float e = (float)o.v[v * 3 + k];
			s->vertexlist[v * 3 + k] = e;
			if (e < s->bbmin[k])
				s->bbmin[k] = e;
			if (e > s->bbmax[k])
				s->bbmax[k] = e;
		}
}

// scene.h:225-331 (static branch)
extern "C" int ugrt_scene_load_model(ugrt_scene *s, const char *path)
{
	if (!s || !path)
		return ugrt_fail(UGRT_EINVAL, "load_model: null argument");
	ObjFile o;
	if (!parse_obj(path, o))
		return ugrt_fail(UGRT_EIO, "load_model: cannot open %s: %s", path, strerror(errno));
	size_t nf = o.fmat.size();
	size_t nv = o.v.size() / 3;
	s->facelist.resize(nf * 3);
	s->matidx.resize(nf);
	for (size_t f = 0; f < nf; f++) {
		for (int k = 0; k < 3; k++) {
			int idx = o.fidx[f * 4 + k];
			if (idx < 0 || (size_t)idx >= nv)
				return ugrt_fail(UGRT_EIO, "load_model: %s: face %zu references vertex %d of %zu", path, f,
						 idx, nv);
			s->facelist[f * 3 + k] = idx;
		}
		s->matidx[f] = o.fmat[f];
	}
	scene_take_vertices(s, o);
	s->reflect.resize(o.mtl.size());
	for (size_t i = 0; i < o.mtl.size(); i++)
		s->reflect[i] = (float)o.mtl[i].reflect;
	return UGRT_OK;
}

// scene.h:70-120 tmp_model: next frame of a dynamic directory, vertices only
extern "C" int ugrt_scene_load_frame(ugrt_scene *s, const char *dir, int frame)
{
	if (!s || !dir)
		return ugrt_fail(UGRT_EINVAL, "load_frame: null argument");
	char name[64];
	snprintf(name, sizeof(name), "/f_%d.obj", frame);
	std::string path = std::string(dir) + name;
	ObjFile o;
	if (!parse_obj(path.c_str(), o))
		return ugrt_fail(UGRT_EIO, "load_frame: cannot open %s", path.c_str());
	if (!s->vertexlist.empty() && o.v.size() != s->vertexlist.size())
		return ugrt_fail(UGRT_EIO, "load_frame: %s has %zu vertices, scene has %zu", path.c_str(),
				 o.v.size() / 3, s->vertexlist.size() / 3);
	scene_take_vertices(s, o);
	return UGRT_OK;
}

// ---------------------------------------------------------------------------
// binary scene cache
// ---------------------------------------------------------------------------
static const char CACHE_MAGIC[8] = { 'U', 'G', 'R', 'T', 'S', 'C', 'N', '1' };

template <typename T> static bool put_vec(FILE *fp, const std::vector<T> &v)
{
	uint64_t n = v.size();
	return fwrite(&n, 8, 1, fp) == 1 && (n == 0 || fwrite(v.data(), sizeof(T), n, fp) == n);
}
template <typename T> static bool get_vec(FILE *fp, std::vector<T> &v, uint64_t limit)
{
	uint64_t n = 0;
	if (fread(&n, 8, 1, fp) != 1 || n > limit)
		return false;
	v.resize(n);
	return n == 0 || fread(v.data(), sizeof(T), n, fp) == n;
}

extern "C" int ugrt_scene_save_cache(const ugrt_scene *s, const char *path)
{
	if (!s || !path)
		return ugrt_fail(UGRT_EINVAL, "scene_save_cache: null argument");
	FILE *fp = fopen(path, "wb");
	if (!fp)
		return ugrt_fail(UGRT_EIO, "scene_save_cache: cannot open %s: %s", path, strerror(errno));
	int32_t nm = s->num_materials;
	bool ok = fwrite(CACHE_MAGIC, 1, 8, fp) == 8 && fwrite(&nm, 4, 1, fp) == 1 && fwrite(s->bbmin, 4, 3, fp) == 3 &&
		  fwrite(s->bbmax, 4, 3, fp) == 3 && put_vec(fp, s->vertexlist) && put_vec(fp, s->facelist) &&
		  put_vec(fp, s->matidx) && put_vec(fp, s->materiallist) && put_vec(fp, s->reflect);
	ok = (fclose(fp) == 0) && ok;
	if (!ok)
		return ugrt_fail(UGRT_EIO, "scene_save_cache: short write to %s", path);
	return UGRT_OK;
}

extern "C" int ugrt_scene_load_cache(ugrt_scene *s, const char *path)
{
	if (!s || !path)
		return ugrt_fail(UGRT_EINVAL, "scene_load_cache: null argument");
	FILE *fp = fopen(path, "rb");
	if (!fp)
		return ugrt_fail(UGRT_EIO, "scene_load_cache: cannot open %s: %s", path, strerror(errno));
	char magic[8];
	ugrt_scene t;
	int32_t nm = 0;
	// no list can hold more elements than the file has bytes: a corrupt length never triggers a huge allocation
	uint64_t lim = 0;
	if (fseek(fp, 0, SEEK_END) == 0) {
		const long sz = ftell(fp);
		lim = sz > 0 ? (uint64_t)sz : 0;
		rewind(fp);
	}
	bool ok = fread(magic, 1, 8, fp) == 8 && memcmp(magic, CACHE_MAGIC, 8) == 0 && fread(&nm, 4, 1, fp) == 1 &&
		  fread(t.bbmin, 4, 3, fp) == 3 && fread(t.bbmax, 4, 3, fp) == 3 && get_vec(fp, t.vertexlist, lim) &&
		  get_vec(fp, t.facelist, lim) && get_vec(fp, t.matidx, lim) && get_vec(fp, t.materiallist, lim) &&
		  get_vec(fp, t.reflect, lim);
	fclose(fp);
	ok = ok && nm >= 0 && t.materiallist.size() == (size_t)nm * 6 && t.vertexlist.size() % 3 == 0 &&
	     t.facelist.size() == t.matidx.size() * 3;
	if (ok)
		for (int m : t.matidx)
			if (m >= nm) { // (negative = no material, as the loader leaves it)
				ok = false;
				break;
			}
	if (ok) {
		const size_t nv = t.vertexlist.size() / 3;
		for (int idx : t.facelist)
			if (idx < 0 || (size_t)idx >= nv) {
				ok = false;
				break;
			}
	}
	if (!ok)
		return ugrt_fail(UGRT_EIO, "scene_load_cache: %s is not a valid ugrt scene cache", path);
	t.num_materials = nm;
	*s = t;
	return UGRT_OK;
}

extern "C" int ugrt_scene_counts(const ugrt_scene *s, int *nv, int *nf, int *nm)
{
	if (!s)
		return ugrt_fail(UGRT_EINVAL, "scene_counts: null scene");
	if (nv)
		*nv = (int)(s->vertexlist.size() / 3);
	if (nf)
		*nf = (int)(s->facelist.size() / 3);
	if (nm)
		*nm = s->num_materials;
	return UGRT_OK;
}
extern "C" const float *ugrt_scene_vertexlist(const ugrt_scene *s) { return s ? s->vertexlist.data() : nullptr; }
extern "C" const int *ugrt_scene_facelist(const ugrt_scene *s) { return s ? s->facelist.data() : nullptr; }
extern "C" const int *ugrt_scene_materiallist_index(const ugrt_scene *s) { return s ? s->matidx.data() : nullptr; }
extern "C" const float *ugrt_scene_materiallist(const ugrt_scene *s) { return s ? s->materiallist.data() : nullptr; }
extern "C" const float *ugrt_scene_reflectlist(const ugrt_scene *s, int *n)
{
	if (!s)
		return nullptr;
	if (n)
		*n = (int)s->reflect.size();
	return s->reflect.data();
}
extern "C" int ugrt_scene_bounds(const ugrt_scene *s, float bbmin[3], float bbmax[3])
{
	if (!s || !bbmin || !bbmax)
		return ugrt_fail(UGRT_EINVAL, "scene_bounds: null argument");
	memcpy(bbmin, s->bbmin, sizeof(float) * 3);
	memcpy(bbmax, s->bbmax, sizeof(float) * 3);
	return UGRT_OK;
}

// ---------------------------------------------------------------------------
// Camera.  GL is gone: gluPerspective / gluLookAt are evaluated here with the
// Mesa-GLU formulas (double for the projection, float for the look-at), then
// camera.h's own code follows (MVP :150, planes :167, corners :241).
// ---------------------------------------------------------------------------
static void cross3(float *d, const float *a, const float *b)
{
	d[0] = a[1] * b[2] - a[2] * b[1];
	d[1] = a[2] * b[0] - a[0] * b[2];
	d[2] = a[0] * b[1] - a[1] * b[0];
}

static void plane_isect(float *p, const float *n1, const float *n2, const float *n3)
{
	float n2n3[3], n3n1[3], n1n2[3];
	n1n2[0] = (n1[1] * n2[2] - n2[1] * n1[2]);
	n1n2[1] = (n1[2] * n2[0] - n1[0] * n2[2]);
	n1n2[2] = (n1[0] * n2[1] - n2[0] * n1[1]);
	n2n3[0] = (n2[1] * n3[2] - n3[1] * n2[2]);
	n2n3[1] = (n2[2] * n3[0] - n2[0] * n3[2]);
	n2n3[2] = (n2[0] * n3[1] - n3[0] * n2[1]);
	n3n1[0] = (n3[1] * n1[2] - n1[1] * n3[2]);
	n3n1[1] = (n3[2] * n1[0] - n3[0] * n1[2]);
	n3n1[2] = (n3[0] * n1[1] - n1[0] * n3[1]);
	float den = n1[0] * n2n3[0] + n1[1] * n2n3[1] + n1[2] * n2n3[2];
	p[0] = -(n1[3] * n2n3[0] + n2[3] * n3n1[0] + n3[3] * n1n2[0]) / den;
	p[1] = -(n1[3] * n2n3[1] + n2[3] * n3n1[1] + n3[3] * n1n2[1]) / den;
	p[2] = -(n1[3] * n2n3[2] + n2[3] * n3n1[2] + n3[3] * n1n2[2]) / den;
}

extern "C" int ugrt_camera_set(ugrt_camera *cam, const float eye[3], const float look[3], const float up[3],
			       float zn, float zf, float fovy, float aspect)
{
	if (!cam || !eye || !look || !up)
		return ugrt_fail(UGRT_EINVAL, "camera_set: null argument");
	if (!(zf > zn) || !(aspect > 0))
		return ugrt_fail(UGRT_EINVAL, "camera_set: need far > near and aspect > 0");
	memset(cam, 0, sizeof(*cam));
	float *P = cam->projection_matrix, *MV = cam->modelview_matrix, *MVP = cam->mvp_matrix;
	// gluPerspective
	double radians = (double)fovy / 2.0 * 3.14159265358979323846 / 180.0;
	double deltaZ = (double)zf - (double)zn;
	double sine = sin(radians);
	double cotangent = cos(radians) / sine;
	P[0] = (float)(cotangent / (double)aspect);
	P[5] = (float)cotangent;
	P[10] = (float)(-((double)zf + (double)zn) / deltaZ);
	P[11] = -1.0f;
	P[14] = (float)(-2.0 * (double)zn * (double)zf / deltaZ);
	// gluLookAt
	float fw[3] = { look[0] - eye[0], look[1] - eye[1], look[2] - eye[2] }, s[3], u[3];
	float r = __builtin_sqrtf(fw[0] * fw[0] + fw[1] * fw[1] + fw[2] * fw[2]);
	if (r != 0.0f) {
		fw[0] /= r;
		fw[1] /= r;
		fw[2] /= r;
	}
	cross3(s, fw, up);
	r = __builtin_sqrtf(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
	if (r != 0.0f) {
		s[0] /= r;
		s[1] /= r;
		s[2] /= r;
	}
	cross3(u, s, fw);
	MV[0] = s[0];
	MV[4] = s[1];
	MV[8] = s[2];
	MV[1] = u[0];
	MV[5] = u[1];
	MV[9] = u[2];
	MV[2] = -fw[0];
	MV[6] = -fw[1];
	MV[10] = -fw[2];
	MV[15] = 1.0f;
	for (int i = 0; i < 3; i++)
		MV[12 + i] = MV[i] * (-eye[0]) + MV[4 + i] * (-eye[1]) + MV[8 + i] * (-eye[2]) + 0.0f;
	cam->worldori[0] = eye[0];
	cam->worldori[1] = eye[1];
	cam->worldori[2] = eye[2];
	cam->worldori[3] = 1.0f;
	// camera.h:150-162
	for (int i = 0; i < 4; i++)
		for (int k = 0; k < 4; k++) {
			MVP[i * 4 + k] = 0;
			for (int j = 0; j < 4; j++)
				MVP[i * 4 + k] += (MV[i * 4 + j] * P[j * 4 + k]);
		}
	// camera.h:167-213: row3 -/+ row0, row3 +/- row1, row3 +/- row2
	static const int row[6] = { 0, 0, 1, 1, 2, 2 };
	static const int plus[6] = { 0, 1, 1, 0, 1, 0 };
	for (int i = 0; i < 6; i++) {
		float *pl = cam->frustum_plane_eq[i];
		for (int c = 0; c < 4; c++)
			pl[c] = plus[i] ? MVP[c * 4 + 3] + MVP[c * 4 + row[i]] : MVP[c * 4 + 3] - MVP[c * 4 + row[i]];
		pl[4] = __builtin_sqrtf(pl[0] * pl[0] + pl[1] * pl[1] + pl[2] * pl[2]);
	}
	for (int i = 0; i < 6; i++) {
		float *pl = cam->frustum_plane_eq[i];
		pl[0] /= pl[4];
		pl[1] /= pl[4];
		pl[2] /= pl[4];
		pl[3] /= pl[4];
	}
	// camera.h:241-253
	static const int tri[8][3] = { { 0, 2, 4 }, { 1, 2, 4 }, { 1, 3, 4 }, { 0, 3, 4 },
				       { 0, 2, 5 }, { 1, 2, 5 }, { 1, 3, 5 }, { 0, 3, 5 } };
	for (int i = 0; i < 8; i++)
		plane_isect(cam->frustumcorner[i], cam->frustum_plane_eq[tri[i][0]], cam->frustum_plane_eq[tri[i][1]],
			    cam->frustum_plane_eq[tri[i][2]]);
	// per_frame_funcs.h:20-37
	float *cc = cam->camcoords;
	for (int i = 0; i < 4; i++)
		cc[i] = cam->worldori[i];
	for (int i = 0; i < 4; i++)
		for (int k = 0; k < 3; k++)
			cc[i * 3 + k + 4] = cam->frustumcorner[i][k];
	for (int i = 0; i < 16; i++) {
		cc[16 + i] = MV[i];
		cc[32 + i] = P[i];
		cc[48 + i] = MVP[i];
	}
	return UGRT_OK;
}

// per_frame_funcs.h:161-419: node(j,i) of the 5x5 table.  Differences in
// float, products and sums in double, every stored value rounded to float --
// exactly what the reference's C expressions do.
static float lerp_host(float a, float b, double w) { return (float)((double)a + w * (double)(float)(b - a)); }

extern "C" int ugrt_camera_direction_table(const float cc[64], float table[100])
{
	if (!cc || !table)
		return ugrt_fail(UGRT_EINVAL, "direction_table: null argument");
	static const double wq[5] = { 0.0, 0.25, 0.5, 0.75, 1.0 };
	for (int j = 0; j < 5; j++)
		for (int i = 0; i < 5; i++) {
			for (int k = 0; k < 3; k++) {
				float c0 = cc[4 + k], c1 = cc[7 + k], c2 = cc[10 + k], c3 = cc[13 + k];
				float a = (i == 0) ? c0 : (i == 4) ? c1 : lerp_host(c0, c1, wq[i]);
				float b = (i == 0) ? c3 : (i == 4) ? c2 : lerp_host(c3, c2, wq[i]);
				table[(j * 5 + i) * 4 + k] = (j == 0) ? a : (j == 4) ? b : lerp_host(a, b, wq[j]);
			}
			table[(j * 5 + i) * 4 + 3] = 0.0f;
		}
	return UGRT_OK;
}

// per_app_funcs.h:39-66
extern "C" int ugrt_write_ppm(const char *path, int W, int H, const unsigned char *rgb)
{
	if (!path || !rgb || W <= 0 || H <= 0)
		return ugrt_fail(UGRT_EINVAL, "write_ppm: bad argument");
	FILE *fp = fopen(path, "w");
	if (!fp)
		return ugrt_fail(UGRT_EIO, "write_ppm: fopen %s: %s", path, strerror(errno));
	// one large buffer: the reference's 3*W*H fprintf calls are the slow part of its frame
	std::string out;
	out.reserve((size_t)W * H * 12 + 64);
	char num[32];
	out += "P3\n";
	snprintf(num, sizeof(num), "%d %d\n", W, H);
	out += num;
	out += "255\n";
	const size_t total = (size_t)3 * W * H, rowlen = (size_t)3 * W;
	for (size_t i = 0; i < total; i++) {
		if (i % rowlen == 0)
			out += '\n';
		unsigned v = rgb[i];
		if (v >= 100) {
			out += (char)('0' + v / 100);
			out += (char)('0' + (v / 10) % 10);
		} else if (v >= 10) {
			out += (char)('0' + v / 10);
		}
		out += (char)('0' + v % 10);
		out += ' ';
	}
	out += '\n';
	size_t w = fwrite(out.data(), 1, out.size(), fp);
	int rc = fclose(fp);
	if (w != out.size() || rc != 0)
		return ugrt_fail(UGRT_EIO, "write_ppm: short write to %s", path);
	return UGRT_OK;
}

extern "C" int ugrt_rot_cos_sin(float rot, float *c, float *s)
{
	if (!c || !s)
		return ugrt_fail(UGRT_EINVAL, "rot_cos_sin: null argument");
	*c = cosf(rot);
	*s = sinf(rot);
	return UGRT_OK;
}
