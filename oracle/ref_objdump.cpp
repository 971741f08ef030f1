// ref_objdump -- test-infrastructure driver around the REFERENCE's OBJ loader.
// Linked against /root/reference/obj_parser/*.cpp (oracle/Makefile target "ref");
// it flattens objLoader's output exactly as Model::load_model does
// (scene.h:225-331: first three vertex indices of every face, material_index
// per face, double -> float vertices, min/max with the 9999.9 sentinels) and
// writes it as a little-endian binary blob:
//   int32 num_vertices, num_faces, num_materials
//   float32 verts[3V]; int32 faces[3F]; int32 mat_idx[F]; float32 bbox[6]
//   per material: float64 amb[3], diff[3], spec[3], reflect, refract, trans,
//                 shiny, glossy, refract_index
// Usage: ref_objdump scene.obj out.bin      (run with cwd = the .obj's directory:
// the reference opens the mtllib path relative to the cwd, obj_parser.cpp:417)
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "objLoader.h"

int main(int argc, char **argv)
{
	if (argc < 3) {
		fprintf(stderr, "usage: %s scene.obj out.bin\n", argv[0]);
		return 2;
	}
	objLoader *objData = new objLoader();
	if (!objData->load(argv[1]))
		return 3;
	FILE *fp = fopen(argv[2], "wb");
	if (!fp)
		return 4;
	int32_t hdr[3] = { objData->vertexCount, objData->faceCount, objData->materialCount };
	fwrite(hdr, sizeof(int32_t), 3, fp);
	float xMin = 9999.9f, yMin = 9999.9f, zMin = 9999.9f;
	float xMax = -9999.9f, yMax = -9999.9f, zMax = -9999.9f;
	for (int v = 0; v < objData->vertexCount; v++) {
		float e[3];
		e[0] = objData->vertexList[v]->e[0];
		e[1] = objData->vertexList[v]->e[1];
		e[2] = objData->vertexList[v]->e[2];
		if (e[0] < xMin) xMin = e[0];
		if (e[0] > xMax) xMax = e[0];
		if (e[1] < yMin) yMin = e[1];
		if (e[1] > yMax) yMax = e[1];
		if (e[2] < zMin) zMin = e[2];
		if (e[2] > zMax) zMax = e[2];
		fwrite(e, sizeof(float), 3, fp);
	}
	for (int f = 0; f < objData->faceCount; f++) {
		int32_t idx[3] = { objData->faceList[f]->vertex_index[0], objData->faceList[f]->vertex_index[1],
				   objData->faceList[f]->vertex_index[2] };
		fwrite(idx, sizeof(int32_t), 3, fp);
	}
	for (int f = 0; f < objData->faceCount; f++) {
		int32_t m = objData->faceList[f]->material_index;
		fwrite(&m, sizeof(int32_t), 1, fp);
	}
	float bbox[6] = { xMin, yMin, zMin, xMax, yMax, zMax };
	fwrite(bbox, sizeof(float), 6, fp);
	for (int i = 0; i < objData->materialCount; i++) {
		obj_material *m = objData->materialList[i];
		double d[15] = { m->amb[0], m->amb[1], m->amb[2], m->diff[0], m->diff[1], m->diff[2],
				 m->spec[0], m->spec[1], m->spec[2], m->reflect, m->refract, m->trans,
				 m->shiny, m->glossy, m->refract_index };
		d[10] = 0.0; /* obj_material.refract is never initialised by the reference */
		fwrite(d, sizeof(double), 15, fp);
	}
	fclose(fp);
	return 0;
}
