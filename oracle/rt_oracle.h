/*
 * rt_oracle.h -- CPU restatement of the Soulztorm/Raytracer per-pixel tracing path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle: only tests, the smoke check and the
 * benchmark's cpu_baseline leg may load it.  The product library never links or calls it.
 *
 * Every function follows a reference function op-for-op in glm 0.9.9.9's evaluation order
 * (reference file:line cited at each definition in rt_oracle.c).  Plain C, IEEE binary32,
 * compiled with -ffp-contract=off so no multiply-add is ever fused.
 *
 * Pinning: validated against the reference's own objects (oracle/_ref: KDAccel, Camera,
 * Walnut::Random/pcg32, glm, built unmodified from /root/reference) and against outputs the
 * unmodified reference Renderer produced in this container (accumulation-buffer checksums,
 * camera/RNG known answers) -- see tests/test_oracle_*.py.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z, radius; uint32_t material; } orc_sphere;          /* Scene.h:10-15 */
typedef struct { float albedo[3], emission[3], roughness, transparency, ior; } orc_material; /* Scene.h:26-35 */

/* Renderer::Settings (Renderer.h:103-110) + the scene selector extended to "both". */
typedef struct {
    int32_t  accumulate;      /* Settings::Accumulate                                  */
    int32_t  scene_mode;      /* 0 = mesh only (UseSphereScene=false), 1 = spheres only, 2 = both */
    int32_t  use_ace;         /* Settings::UseACE_Color                                */
    int32_t  anti_aliasing;   /* Settings::AntiAliasing                                */
    uint32_t bounces;         /* Settings::Bounces                                     */
} orc_settings;

enum { ORC_RNG_SERIAL_LIBM = 0,      /* mode A: one default-seeded pcg32 stream in row-major pixel
                                        order + libm trig/pow: what a serial run of the reference does */
       ORC_RNG_SERIAL_LIBM_DOUBLE = 2,/* mode A': as A, but RandomUnitVector's unqualified cos/sin/sqrt
                                        resolve to the double overloads (libstdc++ without <math.h>)   */
       ORC_RNG_PIXEL_HASH = 1 };     /* mode B: pcg32 seeded per (pixel, frame) + portable sincos/pow:
                                        the parity definition the GPU is compared against            */

typedef struct {
    uint64_t rays;            /* TraceRay invocations (Renderer.cpp:154)       */
    uint64_t nodes_visited;   /* aabbIntersect calls  (KDTreeCPU.cpp:654)      */
    uint64_t tris_tested;     /* triIntersect calls   (KDTreeCPU.cpp:676)      */
    uint64_t sphere_tests;    /* sphere loop bodies   (Renderer.cpp:250-270)   */
} orc_counters;

typedef struct orc_scene orc_scene;

orc_scene* orc_scene_create(void);
void       orc_scene_destroy(orc_scene*);
int  orc_scene_set_spheres(orc_scene*, const orc_sphere*, uint32_t n);
int  orc_scene_set_materials(orc_scene*, const orc_material*, uint32_t n);
/* verts: nverts*3 floats; tri_idx: ntris*3 uint32; normals: ntris*9 floats; tri_mat: ntris uint32 */
int  orc_scene_set_mesh(orc_scene*, const float* verts, uint32_t nverts, const uint32_t* tri_idx,
                        const float* normals9, const uint32_t* tri_mat, uint32_t ntris);
/* KD tree: naive restatement of constructTreeStackless (KDTreeCPU.cpp:424-632). */
int  orc_scene_build_kd(orc_scene*);
/* KD tree: import a flat tree (children -1 = none; leaves carry first/count into leaf_idx). */
int  orc_scene_import_kd(orc_scene*, uint32_t n_nodes, const float* center3, const float* extent3,
                         const int32_t* left, const int32_t* right, const int32_t* first,
                         const int32_t* count, uint32_t n_leaf_idx, const int32_t* leaf_idx);
/* sizes then arrays (call twice); node order: preorder, left before right; every node's own triangle
   list is exported (interior nodes keep theirs, as the reference does). */
int  orc_scene_kd_sizes(const orc_scene*, uint32_t* n_nodes, uint32_t* n_leaf_idx, uint32_t* depth);
int  orc_scene_export_kd(const orc_scene*, float* center3, float* extent3, int32_t* left, int32_t* right,
                         int32_t* first, int32_t* count, int32_t* leaf_idx);

/* Camera (Camera.cpp:11-19,122-150): matrices are column-major float[16] like glm::mat4. */
void orc_camera_forward_default(float fwd[3]);                           /* Camera.cpp:14 */
void orc_camera_projection(float fov_deg, float w, float h, float znear, float zfar,
                           float proj[16], float inv_proj[16]);          /* Camera.cpp:122-126 */
void orc_camera_view(const float pos[3], const float fwd[3], float view[16], float inv_view[16]); /* :128-132 */
void orc_camera_ray_dirs(const float inv_proj[16], const float inv_view[16], uint32_t w, uint32_t h,
                         float* dirs3);                                  /* Camera.cpp:134-150 */

/* One Renderer::Render call (Renderer.cpp:45-135) for frame index `frame_index` (1-based).
   accum3: w*h*3 floats, read-modify-written (cleared first when frame_index == 1);
   image4: w*h*4 floats RGBA32F; primary_ids (optional): -1 miss, sphere index, or 0x40000000|triangle.
   Rows are spread over `nthreads` threads only in ORC_RNG_PIXEL_HASH mode (serial modes run on one). */
int  orc_render_frame(const orc_scene*, const orc_settings*, const float cam_pos[3], const float* dirs3,
                      uint32_t w, uint32_t h, uint32_t frame_index, int rng_mode, uint32_t seed,
                      float* accum3, float* image4, int32_t* primary_ids, orc_counters* counters,
                      int nthreads, uint32_t row_step);
/* Resets the serial pcg32 stream to its default-constructed state (pcg_random.hpp:484). */
void orc_serial_rng_reset(void);

/* Primary rays only: id / t / u / v per pixel (for ID parity and the KD traversal pin). */
int  orc_trace_primary(const orc_scene*, int scene_mode, const float cam_pos[3], const float* dirs3,
                       uint32_t n, int32_t* ids, float* t, float* u, float* v, orc_counters* counters);

/* Unit pieces for known-answer tests. */
uint32_t orc_pcg32_next(uint64_t* state);                                /* pcg_random.hpp:413-437,845-873 */
uint64_t orc_pcg32_seed(uint64_t seed);                                  /* pcg_random.hpp:484-489 */
uint64_t orc_pixel_seed(uint32_t pixel, uint32_t frame, uint32_t seed);
float    orc_u32_to_float(uint32_t u);                                   /* Random.h:11,35-38 */
void     orc_sincos(float a, float* s, float* c);
float    orc_pow_inv24(float x);
int      orc_aabb(const float center[3], const float extent[3], const float o[3], const float inv_d[3],
                  float* t_near);                                        /* Intersections.cpp:23-47 */
int      orc_tri(const float o[3], const float d[3], const float v0[3], const float v1[3],
                 const float v2[3], float* t, float* u, float* v);       /* Intersections.cpp:56-94 */
int      orc_refract(const float d[3], const float n[3], const float p[3], float ior,
                     float out_o[3], float out_d[3]);                    /* Renderer.cpp:334-361 */
void     orc_tonemap(const float rgb_in[3], int use_ace, int portable_pow, float rgb_out[3]); /* Renderer.h:63-98 */
uint32_t orc_fnv1a(const void* data, uint64_t nbytes);   /* FNV-1a folded over 32-bit words */
/* glm 0.9.9.9 primitives as restated here (func_geometric.inl, func_common.inl, func_matrix.inl, type_mat4x4.inl) */
void  orc_glm_normalize(const float a[3], float o[3]);
void  orc_glm_reflect(const float i[3], const float n[3], float o[3]);
void  orc_glm_mix(const float x[3], const float y[3], float a, float o[3]);
float orc_glm_dot(const float a[3], const float b[3]);
void  orc_glm_cross(const float a[3], const float b[3], float o[3]);
void  orc_glm_inverse(const float m[16], float o[16]);
void  orc_glm_mat_vec(const float m[16], const float v[4], float o[4]);

#ifdef __cplusplus
}
#endif
#endif
