// ref_obj_harness.cpp -- TEST INFRASTRUCTURE: the reference's own OBJ loader (RayTracer/src/tiny_obj_loader.h, unmodified,
// compiled with TINYOBJLOADER_IMPLEMENTATION exactly as RaytracerApp.cpp:15-16 does) behind a C function, so that the
// triangulation of raytracer_amd/ingest.py can be pinned to it.  Only built where /root/reference exists (oracle/Makefile,
// output under oracle/_ref/); used by tests/golden/make_golden.py to generate the committed fixture.
#define TINYOBJLOADER_IMPLEMENTATION
#include "tiny_obj_loader.h"

#include <cstdint>

#define API extern "C" __attribute__((visibility("default")))

// Parses `path` the way RaytracerLayer does (tinyobj::ObjReader with the default config: triangulate = true,
// RaytracerApp.cpp:28-32) and flattens the faces in its order (RaytracerApp.cpp:60-127).  Writes up to `cap` triangles:
// vertex indices (3 per triangle), normal indices (3, -1 = none) and the face's material id.  Returns the triangle count, or
// -1 when the file does not parse.
API int ref_obj_triangles(const char* path, int cap, int32_t* v_idx, int32_t* n_idx, int32_t* mat)
{
    tinyobj::ObjReader reader;
    tinyobj::ObjReaderConfig config;
    if (!reader.ParseFromFile(path, config)) return -1;
    const auto& shapes = reader.GetShapes();
    int n = 0;
    for (size_t s = 0; s < shapes.size(); s++) {
        size_t offset = 0;
        for (size_t f = 0; f < shapes[s].mesh.num_face_vertices.size(); f++) {
            const size_t fv = size_t(shapes[s].mesh.num_face_vertices[f]);
            if (fv == 3 && n < cap) {
                for (size_t k = 0; k < 3; k++) {
                    const tinyobj::index_t idx = shapes[s].mesh.indices[offset + k];
                    v_idx[3 * n + k] = idx.vertex_index; n_idx[3 * n + k] = idx.normal_index;
                }
                mat[n] = shapes[s].mesh.material_ids[f];
            }
            if (fv == 3) ++n;
            offset += fv;
        }
    }
    return n;
}
