// rt_device.h -- data layouts shared by the host side of the library and the gfx950 kernels.
#pragma once
#include <stdint.h>

#define RT_TILE 16                 // a 256-thread workgroup covers 16x16 pixels; each wave an 8x8 tile
#define RT_ID_TRI_BIT 0x40000000

#define RT_MODE_MESH    0     // rt_settings.scene_mode values
#define RT_MODE_SPHERES 1
#define RT_MODE_BOTH    2

#define RT_UNIT 16                 // pixels per work unit: an 8x2 strip of an 8x8 tile
#ifndef RT_TREE_BLOCK
#define RT_TREE_BLOCK 1024         /* threads per workgroup of the tree-scene trace kernels (its waves share the path pool in LDS) */
#endif
#define RT_SPHERE_BLOCK 1024       // threads per workgroup of the sphere-scene trace kernel (16 waves sharing one unit counter in LDS)
#ifndef RT_SPHERE_MERGE
#define RT_SPHERE_MERGE 48         /* sphere kernel, end of a share: a wave with at most this many paths hands them to another wave (0 = off) */
#endif
#ifndef RT_SPHERE_PRIO
#define RT_SPHERE_PRIO 6           /* sphere kernel: iterations of every 8 in which the younger workgroup of a CU has issue priority */
#endif
#ifndef RT_SPHERE_PRIO_PERIOD
#define RT_SPHERE_PRIO_PERIOD 8
#endif
#define RT_SPHERE_POOL_SLOTS 128   // paths waiting for another wave (at most 63 + RT_SPHERE_MERGE at a time)
#define RT_SPHERE_POOL_WORDS 20    // 32-bit words of a sphere-scene path
#define RT_PARK_SLOTS 16384        // bounce rays the tree cannot prune that one round can hand to whole-chip kernels (a launch of 16 frames of
                                   // spheres + mesh at 4K parks about 5,600)
#define RT_PARK_BOXES (RT_PARK_SLOTS / 16 + RT_PARK_SLOTS / 64 + RT_PARK_SLOTS / 256)   // boxes of 16, 64 and 256 sorted rays of a family (two float4 each)
#define RT_PARK_FAMILIES 2         // ... grouped by (origin.x, sign of the zero): rays of a family visit the same nodes
#define RT_PARK_MIN_NODES (1u << 18)   // ... in trees at least this large by default (in smaller ones the wave that owns the ray walks it)
#define RT_PARK_MIN_NODES_MIXED (1u << 12)   // ... and with spheres and mesh together, where such rays come by the hundred
#define RT_PARK_WORDS 5            // float4 per parked path
#define RT_MAX_LAUNCH_FRAMES 16    // tree scenes: Renders one launch can hold (a plane of per-pixel radiance slots each)
#define RT_WALK_FETCH 6           /* float4 read per walk trip at most: up to three consecutive nodes or two leaf records */
#ifndef RT_EXACT_REF_COOP
#define RT_EXACT_REF_COOP 1       /* experiments: 0 = the last walks of a wave in the reference's order use the windowed walk of the front-to-back order */
#endif
#ifndef RT_TRIP_NODES
#define RT_TRIP_NODES 2           /* node steps a lane makes from one read, while it keeps descending to the next node in memory (1..3) */
#endif
#ifndef RT_TRIP_RECS
#define RT_TRIP_RECS 1            /* leaf records a lane tests from one read (1..2).  Measured (ms per Render / per frame at 16 Renders per launch,
                                     70 k triangles | 10 M triangles | spheres + 10 M triangles): 3 + 2: 0.400 / 0.117 | 2.44 / 1.82 | 8.27 / 7.03;
                                     2 + 2: 0.392 / 0.114 | 2.40 / 1.78 | 8.21 / 6.95; 2 + 1: 0.402 / 0.113 | 2.42 / 1.72 | 7.97 / 6.67;
                                     1 + 1 and 3 + 1 are slower than 3 + 2.  Later steps of a trip run for the few lanes that kept descending */
#endif
#ifndef RT_NEAR_TIES
#define RT_NEAR_TIES 1             /* 0 compiles the near-tie test and the repeat walk out (A/B measurements) */
#endif
/* Near ties of the front-to-back walk: see RtParams::tie_* for the distance within which two different triangles count as one. */
#ifndef RT_SLICE_LEAVE
#define RT_SLICE_LEAVE 2           /* a walk slice ends when this many quarters of its walkers are left ... */
#endif
#ifndef RT_SLICE_LEAVE_LARGE
#define RT_SLICE_LEAVE_LARGE 3     /* ... in a mesh-only scene with a tree of at least RT_LARGE_TREE nodes, or a launch of several Renders (walks
                                      there are long and alike: shorter slices keep more lanes walking; the host chooses, see fill_params) */
#endif
#define RT_LARGE_TREE (1u << 18)
#ifndef RT_COOP_LANES
#define RT_COOP_LANES 2            /* a wave with at most this many walks left lends all its lanes to one walk at a time */
#endif
/* Tree kernels, end of a frame: a wave that has run out of work adopts paths from the other waves of its workgroup (through a
   pool in LDS) and, holding at most RT_COOP_LANES of them, walks the tree for them with all its lanes. */
#ifndef RT_SHARE
#define RT_SHARE 1                 /* 0 compiles the hand-over out (A/B measurements) */
#endif
#ifndef RT_ADOPT
#define RT_ADOPT 2                 /* paths a wave without work takes from the pool at a time */
#endif
#ifndef RT_KEEP
#define RT_KEEP 1                  /* paths a wave keeps for itself when it gives others away (2: +5 % at 70 k triangles, +2 % at 10 M) */
#endif
#ifndef RT_PLAIN_SLAB
#define RT_PLAIN_SLAB 1            /* 0 compiles the short form of the box test out (A/B measurements) */
#endif
#ifndef RT_CHECKED
#define RT_CHECKED 0               /* 1: the kernels test the indices their per-lane state machines carry (walk cursors, the triangle / sphere /
                                      material of a hit, slots of parked and column rays) against their arrays before using them; a failed
                                      test is recorded in column_stats[4], the index replaced by 0, and rt_read_counters reports it as an
                                      error -- a fault of a state machine then shows as a message, not as a GPU memory fault (for the
                                      randomised probe, tools/fuzz_parity.py) */
#endif
#define RT_POOL_SLOTS 16           // paths the pool holds while they change waves
#define RT_POOL_WORDS 38           // 32-bit words of one path's state
#define RT_LONG_WALK 48            // node + triangle steps after which a tree walk raises its wave's issue priority
#define RT_WAVE_CLOCK_WORDS 8
#define RT_WORK_HEADS 8            // ticket counters units are pulled from on tree scenes (one per XCD under round-robin dispatch), each on a 128-byte line of its own
#define RT_HEAD_STRIDE 32          // words between heads
#define RT_INLINE_SPHERES 32       // spheres carried in the kernel arguments (scalar loads); more go through HBM
#define RT_LDS_SPHERES 128         // scenes with at most this many spheres and materials keep them in LDS for the per-lane
#define RT_LDS_MATERIALS 128       // lookups of the closest hit (a few KB per workgroup); larger scenes read them from HBM

// rt_render_spheres_noalloc: fixed limits of the one entry point whose working memory is a static device array
#define RT_STATIC_MAX_PIXELS (1u << 21)      // 2,097,152 >= 1920 x 1080; one float4 ray direction each = 32 MiB per GPU
#define RT_STATIC_MAX_SPHERES 32             // = RT_INLINE_SPHERES: the sphere loop reads them from the kernel arguments
#define RT_STATIC_MAX_MATERIALS 32

#define RT_LEAF_INTERIOR (-1)      // RtNode.leaf_first: not a leaf
#define RT_LEAF_EMPTY    (-2)      // a leaf that owns no triangle

// HBM layouts (all 16-byte aligned, read with dwordx4 loads):
//
//  nodes    : 8 arrays of 2 x float4 per node.  Array 0 holds the nodes in PREORDER (left subtree before right subtree),
//             so the traversal order of KDTreeCPU::intersect (KDAccel/KDTreeCPU.cpp:649-708: box test, then left, then
//             right) is "next node = i + 1 on descend, skip[i] otherwise" with no per-lane stack:
//               [0] = { center.x, center.y, center.z, bits(skip) }       skip = index after this node's subtree
//               [1] = { extent.x, extent.y, extent.z, bits(leaf_first) } first leaf record, or RT_LEAF_*
//             Array o (bit a of o set = direction[a] < 0) holds the same nodes in the preorder that puts the child on
//             the near side of such a ray first, with skip links of its own; leaf_first is the same in all arrays.
//  leafrecs : 3 x float4 per (leaf, triangle) reference, leaves in preorder, triangles in the leaf's own order,
//             vertices pre-gathered so a leaf is one contiguous run:
//               [0] = { v0.x, v0.y, v0.z, e1.x }   e1 = v1 - v0, e2 = v2 - v0 (the first two operations of
//               [1] = { e1.y, e1.z, e2.x, e2.y }   Intersections::triIntersect, Intersections.cpp:58-59, done once)
//               [2] = { e2.z, bits(tri | last<<31), bits(node index of the leaf), 0 }
//  trishade : 3 x float4 per triangle (read once per closest hit): the three corner normals + material
//               [0] = { n0.x, n0.y, n0.z, n1.x }  [1] = { n1.y, n1.z, n2.x, n2.y }  [2] = { n2.z, bits(material), 0, 0 }
//  spheres  : float4 { center.xyz, radius*radius } + uint32 material, index order = Scene::spheres order
//  materials: 3 x float4 { albedo.rgb, roughness } { emission.rgb, transparency } { ior, 0, 0, 0 }
//  dirs     : float4 per pixel { direction.xyz, 0 }: Camera::m_RayDirections (Camera.h:43), kept until the camera moves
//  accum    : float4 per pixel { r, g, b, unused }, row-major, row 0 first (m_AccumulationBuffer, Renderer.h:157)
//  image    : float4 per pixel RGBA32F (m_ImageData, Renderer.h:156)

struct RtParams {
    float cam_pos[3];
    float inv_proj[16];
    float inv_view[16];
    uint32_t width, height;
    uint32_t frame_first, n_frames;
    uint32_t seed, bounces;
    uint32_t n_spheres, n_materials;
    uint32_t n_nodes, n_tris;
    uint32_t band_index, band_count;       // screen-tile split: this context renders bands b*band_count + band_index
    uint32_t tile_rows, n_units;           // 8-row tile rows / RT_UNIT-pixel work units in the bands this context renders
    uint32_t n_units_main;                 // ... of which the first n_units_main are the 8x2 strips of the owned bands; the rest are 16x1
                                           // strips of the rows that border an owned band (`halo`: the 3x3 filter reads them under a tile split)
    uint32_t halo;
    uint32_t slab_plain;                   // every node box has |centre| + extent <= 2^60 on each axis and every triangle |e1|_1 |e2|_1 <= 2^56 (see slab_plain_ray)
    uint32_t* work_heads;                  // RT_WORK_HEADS ticket counters (RT_HEAD_STRIDE words apart), zero before every trace launch
    int32_t  use_ace;
    uint32_t fuse_resolve;                  // sphere trace kernel: each workgroup also resolves its share of the image
    const float4* spheres;
    const uint32_t* sphere_mat;
    const float4* materials;
    const float4* nodes;
    const float4* leafrecs;
    const float4* trishade;
    const float4* dirs;                     // ray direction per pixel { x, y, z, 0 } (generated on the device or uploaded)
    float4* dirs_out;                       // raygen kernel's output (same buffer)
    float4* accum;
    float4* image;
    float4* slots;                          // tree scenes: n_frames planes of width*height { radiance.rgb, walk steps } -- the value each
                                            // Render of the launch adds to its pixel, summed into `accum` in frame order by the resolve pass
    int32_t* primary_ids;                   // optional
    // Tree scenes, optional: the primary ray of a pixel (Renderer.cpp:141-143,154 at bounce 0) is the same in every Render while
    // camera, directions and tree stay as they are, and TraceRay draws no random number, so its hit can be kept per pixel.
    // first_hit_mode 1: trace only the primary rays and store their mesh hits { t, u, v, bits(triangle id) } (no shading, no
    // radiance); 2: every path starts from the stored hit instead of walking the tree for bounce 0; 0: off.
    float4*  first_hits;
    uint32_t first_hit_mode;
    // Mesh-only scenes, optional: work units none of whose primary rays enters the root box ("empty": their paths end in
    // Miss() with a colour of 0 + 1 * 0, Renderer.cpp:160-165).  Found when camera, directions, bands or tree change; the trace
    // kernel then draws its tickets over unit_list (the n_list other units, ascending) and the resolve pass takes a radiance of
    // zero for the units marked in unit_empty.  unit_*_out: where the classification writes (unit_empty, list, counts = {n_list,
    // pixels of empty units, node steps and triangle tests that unit_entry spares the primary walks of one Render}).
    // The host launches all of this without waiting for any of it: the trace kernel and the entry kernel read the length of the list
    // from unit_counts[0] on the device; n_list is the host's copy, zero until it has arrived (only the resolve pass over the listed
    // units, which runs from the second Render of an accumulation on, needs it on the host).
    const uint32_t* unit_list;
    uint32_t n_list;
    const uint32_t* unit_counts;
    const unsigned char* unit_empty;
    unsigned char* unit_empty_out; uint32_t* unit_list_out; uint32_t* unit_counts_out;
    uint32_t* unit_block_out;               // scratch: per 64 units, listed units | pixels of empty units << 8; then the listed units before them
    // ... and where the walk of a primary ray may start.  The sixteen primary rays of a unit make the same decisions at the first
    // nodes of their walk (same origin, neighbouring directions): unit_entry[pixel] is the first node at which they do not --
    // a box some enter and some miss, or a leaf in which one of them hits a triangle.  Up to there no ray of the unit has hit
    // anything, so a walk that starts at that node with no hit is in exactly the state the walk from the root would be in.
    // Found with the unit list, for the walk order in `entry_ordered`; null = every walk starts at the root.
    const uint32_t* unit_entry;
    uint32_t* unit_entry_out; uint32_t entry_ordered;
    unsigned long long* counters;           // 4 per wave of the persistent grid: rays, nodes, tris, sphere tests
    unsigned long long* wave_clock;         // RT_WAVE_CLOCK_WORDS per wave: [6] ticks spent waiting for paths of other waves, [7] ticks in the shading block; start, end (100 MHz clock), main-loop iterations, walk-loop
                                            // trips, sum over trips of the walking lanes (counting kernels), ticks inside the walk loop
    unsigned long long* column_stats;       // 8: nodes / triangles the reference's walk spends on the column rays; rays walked again in the
                                            // reference's order because of a near tie; primary rays answered from first_hits; [4] the
                                            // largest code of a failed index test (RT_CHECKED builds, else 0)
    // primary rays with direction.x == +0 ("column rays"): numbered when directions change, resolved per frame by a pre-pass
    uint32_t  n_refs;                        // leaf records
    uint32_t  column_cap;
    uint32_t* column_count;                  // number of column rays (may exceed column_cap: the excess is not numbered)
    uint32_t* column_pixels;                 // their pixel indices in pixel order; dirs[pixel].w holds slot + 1 (0 = not a column ray)
    float4*   column_dirs;                   // their directions, by slot
    float4*   column_box16;                  // { min dy, max dy, min dz, max dz } of the directions of slots 16i .. 16i+15
    float4*   column_box64;                  // the same for 64 and 256 consecutive slots
    float4*   column_box256;
    uint32_t* column_block;                  // scratch: per-block counts / maxima and their scans
    uint2*    column_queue;                  // (leaf record, 16-slot box) pairs that need the real triangle test
    uint32_t* column_queue_count;
    uint32_t  column_queue_cap;
    unsigned long long* column_best;         // per slot: bits(t) << 32 | leaf record index of the closest hit, ~0 = miss
    uint32_t* column_flags;                  // bit 0: the pre-pass result may not be used (a decision depended on t, or the queue overflowed)
    unsigned long long* reached;             // one bit per node (bit j & 63 of word j >> 6): a leaf-bearing box the rays of the family enter (pre-pass scratch)
    const uint32_t* rec_leaf;                // per leaf record: node index of its leaf (what leafrecs[3 r + 2].z holds, packed: the kernels that
                                             // stream over all records look here first and read only the records of visited leaves)
    // Bounce rays with direction.x == 0 in a large tree ("parked" paths): the trace kernel stores the path and frees the lane,
    // whole-chip kernels find the hit (same reach + test as for the column rays, one ray at a time), a one-workgroup launch of
    // the trace kernel (resume != 0) continues the paths.
    uint32_t* park_count;                    // paths parked by the running trace launch (may exceed RT_PARK_SLOTS: the excess was not parked)
    float4*   park_paths;                    // RT_PARK_WORDS float4 per slot: {origin, column slot} {direction, bounce} {contribution, pixel} {colour, frame} {rng lo, rng hi}
    const uint32_t* resume_count;            // the list being resolved / resumed (what the previous launch parked): same layout
    const float4*   resume_paths;
    unsigned long long* park_best;           // per slot of that list: bits(t) << 32 | leaf record index of the closest hit, ~0 = miss
    uint32_t* park_flags;                    // per slot of that list, bit 0: no whole-chip result (a decision depended on t, or a third family)
    uint32_t* park_list;                     // RT_PARK_FAMILIES x RT_PARK_SLOTS: the slots of each family, sorted by (pixel, frame)
    unsigned long long* park_keys;           // ... and the sort keys they are sorted by: pixel << 19 | frame << 14 | slot
    float4*   park_boxes;                    // per family: (RT_PARK_BOXES + 1) x 2 float4, boxes of the origins / directions of 16, 64, 256 sorted
                                             // rays, and last of all its rays
    uint2*    park_queue;                    // per family: (leaf record, 16-ray box) pairs that need the real triangle test
    uint32_t* park_queue_count;              // per family
    uint32_t  park_queue_cap;
    int32_t*  park_info;                     // per family: [f] representative slot (-1 = none), [2 + f] size, [4 + f] bit 0: visited set depends on t
    uint32_t  park_min_nodes;                // park only in trees with at least this many nodes
    uint32_t  slice_leave;                   // a walk slice of the trace kernel ends when this many quarters of its walkers are left
    float     root_center[3];                // centre of the tree's root box
    // Near ties of the front-to-back walk.  Its result is KDTreeCPU::intersect's whenever no two different triangles are hit within
    // 2 eps(t) of each other, eps(t) = e_box + e_tri(t) bounding how far a box's computed entry distance can lie behind the computed
    // hit distance of a triangle whose hit point the box contains (proof in the design notes: a subtree is pruned only beyond
    // t + 2 eps, so neither walk can have missed the other's answer).  Rays that do see two such hits are walked again in the
    // reference's order.
    //   e_box    = 4 u max_a (|origin_a - root_centre_a| + tie_box[a]) |1 / direction_a|, u = 2^-24: the slab test's four
    //              roundings per plane, plus the two of the builder's centre / half-extent form of the box; tie_box[a] =
    //              max over the nodes of (|centre_a - root_centre_a| + extent_a) + max over the nodes of (|centre_a| + extent_a)
    //   e_tri(t) = t u (48 K + 4) + 56 u K e_max: first-order rounding analysis of Intersections::triIntersect with a factor
    //              of two to spare, for ANY triangle of the mesh (also one never tested): K = max |e1|_1 |e2|_1 / 1e-5 -- the
    //              test rejects |determinant| < 1e-5, which caps the conditioning -- and e_max the longest edge component
    // tie_ulp = 8 u, tie_rel = 2 u (48 K + 4), tie_abs = 112 u K e_max hold the factor 2.
    float     tie_box[3];
    float     tie_ulp, tie_rel, tie_abs;
    int32_t   ray_slot;                      // which rays the reach / test kernels serve: -1 = the column rays, else a family of parked rays
    uint32_t  resume;                        // trace kernel: 0 = draw units; 1 = continue the paths of the resume list (may park again);
                                             // 2 = the same, but rays the tree cannot prune are walked by their wave
    uint32_t  resume_lanes;                  // ... by the first resume_lanes lanes of each wave, one path each (path w * resume_lanes + lane)
    // Experiments (the counting kernels only): every ray that starts a walk is appended here as { origin, bounce } { direction, pixel }
    float4*   ray_dump;
    uint32_t* ray_dump_count;
    uint32_t  ray_dump_cap;
    float4   sph_inline[RT_INLINE_SPHERES]; // first spheres { center.xyz, radius^2 } by value
};
