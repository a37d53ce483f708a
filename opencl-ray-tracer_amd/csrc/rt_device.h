// Internal interface between the C ABI (rt_capi.cpp) and the HIP kernels (rt_kernels.hip).
#ifndef WBRT_RT_DEVICE_H
#define WBRT_RT_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

#define WBRT_NUM_COUNTERS 26

// Counter slots; same events and order as wbo_counters in the test oracle.
enum
{
	C_PATHS = 0, C_SEGMENTS, C_RNG_DRAWS,
	C_SPHERE_TESTS, C_SPHERE_DISC_GE0, C_SPHERE_SECOND_ROOT, C_SPHERE_ACCEPT,
	C_BBOX_TESTS, C_BBOX_HITS,
	C_TRI_TESTS, C_TRI_EXIT_PARALLEL, C_TRI_EXIT_U, C_TRI_EXIT_V, C_TRI_REACH_T, C_TRI_ACCEPT,
	C_LAMBERT, C_METAL, C_METAL_FUZZ, C_DIELECTRIC, C_DIELECTRIC_DRAW, C_EMISSIVE,
	C_SPHERE_TRIES, C_DISK_TRIES,
	C_MISS, C_DEPTH_EXHAUSTED, C_ABSORBED
};

// Device-resident scene, repacked from the reference's AoS structs on upload.
struct SceneDev
{
	// spheres, structure of arrays
	const float4 *sph_geom;   // [S] center.x, center.y, center.z, radius*radius
	const float  *sph_radius; // [S]
	const float4 *sph_mat_a;  // [S] albedo.rgb, fuzz
	const float2 *sph_mat_b;  // [S] ior, type (int bits)
	int S;
	int sph_moderate;         // 1: every centre coordinate and radius is finite and below 1e8
	// triangles grouped by bounding-box index (stable), 12 floats each:
	//   p0.xyz e0.x | e0.yz e1.xy | e1.z 0 0 0     with e0 = p1-p0, e1 = p2-p0
	const float4 *tri_geom;   // [3*T']
	// what shading a triangle hit needs, 16 bytes per triangle: the outward normal Unit(Cross(e0, e1)) of gpu_kernel.cl:524
	// — it depends on the triangle only, so the upload evaluates the reference's float sequence once, IEEE-exact on the host —
	// and the triangle's entry in the table of DISTINCT materials (meshes share one material, utility.cpp:99-170)
	const float4 *tri_shade;  // [T'] normal.xyz, material index (int bits)
	const float4 *mat_table;  // 2 per distinct material: albedo.rgb fuzz | ior type(int bits) - -
	const int    *tri_orig;   // [T'] index in the caller's triangle array
	int T;                    // T' = triangles whose boundingBoxId names an existing box
	// bounding boxes
	const float4 *box_min;    // [B] min.xyz, 0
	const float4 *box_max;    // [B] max.xyz, 0
	const int2   *box_range;  // [B] first, last+1 into the grouped triangle arrays
	int B;
	// triangle hierarchies, one per bounding box (absent: bvh_nodes == nullptr)
	const float4 *bvh_nodes;  // 4 per node, children interleaved: lmin.x rmin.x lmin.y rmin.y | lmin.z rmin.z lmax.x rmax.x |
	                          // lmax.y rmax.y lmax.z rmax.z | left right E_left E_right
	const float4 *bvh_tris;   // 3 per leaf slot of the sphere hierarchy: center.xyz r^2 | - | - index - -
	const float4 *bvh_pairs;  // 5 per pair slot of a triangle leaf, two triangles interleaved:
	                          // p0.x p0.x' p0.y p0.y' | p0.z p0.z' e0.x e0.x' | e0.y e0.y' e0.z e0.z' | e1.x e1.x' e1.y e1.y' | e1.z e1.z' pos pos'
	                          // (pos: position in the grouped order as int bits; pos' = -1: no second triangle)
	const float4 *bvh_roots;  // 2 per box: min.xyz ref | max.xyz E   (E < 0: box without triangles)
	const uint4  *bvh_mbox;   // 1 per node (proven-margin trees only): bytes 0-2 of x|y|z|w = left centre, left half-extent, right centre,
	                          // right half-extent of the children's m-boxes, m = (e0 x e1)/(|e0||e1|); centre byte b -> b/127 - 1, half-extent b/127
	float bvh_rho;
	int sph_tree;             // 1: bvh_roots[B] is a hierarchy over the spheres (leaf slot: center.xyz r^2 | - | - index - -)
	int bvh_box_slots;        // stack entries free for queued boxes: WBRT_BVH_STACK - tree depth + 1
	int bvh_steal_min_idle;   // idle lanes of a wave needed before a hand-over round is run
	// hierarchy over the boxes themselves (scenes with many boxes; absent: top_nodes == nullptr)
	const float4 *top_nodes;  // 4 per node: lmin.xyz lmax.x | lmax.yz rmin.xy | rmin.z rmax.xyz | left right - -  (leaf: ~rank)
	const int    *top_box;    // rank (left-to-right leaf order) -> box index
	// sky: skyw*skyh texels + skyw+1 zero guard texels, 3 floats each
	const float *sky;
	int skyw, skyh;
};

struct RenderParams
{
	// camera (gpu_kernel.cl:53-69)
	float org[3], hor[3], ver[3], llc[3], ddu[3], ddv[3];
	float aperture;
	int width, height, spp, max_depth;
	float inv_width, inv_height;    // correctly rounded 1/width, 1/height
	// tiling: blocks of bsx*bsy pixels; this launch covers nbx*nby blocks whose block
	// indices start at (off_bx, off_by)
	int bsx, bsy, nbx, nby, off_bx, off_by;
	int plain_coords;               // 1: every pixel column / row of the launch (block offsets included) lies in [0, 2^23)
	// output: texel (row, col) relative to the first block lands at out[(row*out_w + col)*3]
	int out_w;
	int out_mode;
	float *out;
	int *ids;                       // primary-id launch only
	unsigned long long *counters;   // counting launch only
	int *costs;                     // pilot launch: loop iterations per work index, then (mesh scenes) walk rounds per
	                                // 64 work indices (no image is written)
	int *record;                    // optional, full launch: the same counts as `costs`, written next to the image
	                                // (indexed by work index, whatever the launch order)
	const int *order;               // optional: waves (groups of 64 work indices) sorted by estimated cost, descending
	const uint32_t *seeds;
	// fused seed shards (n_shards > 1): ONE launch renders n_shards shards per pixel and adds every shard's sum to the pixel in
	// shard order (out_mode 2): concurrently, 2^shard_lanes_log2 (>= n_shards) adjacent lanes per pixel, or (shard_lanes_log2 ==
	// 0) in turn in one lane.  Shard k reads seeds + k * seeds_stride and takes spp samples, one less from shard n_full on.
	int shard_lanes_log2;
	int n_shards, n_full, seeds_stride;
	int tolerant;                   // 1: the tolerant arithmetic build of the render kernel (render launches only)
	int wg;                         // 4 or 8: scenes without meshes use the workgroup kernel (that many waves) that re-deals pixel streams
	                                // between its waves; 0: the per-wave kernel
	int wg_per_cu;                  // workgroups per compute unit of that kernel's launch (0: 6 of 4 waves, 3 of 8)
	unsigned *next;                 // workgroup kernel: work counter in device memory (the launch zeroes it)
	SceneDev sc;
};

// Binary hierarchy over the triangles of every bounding box.  A child reference >= 0 names a
// node; < 0 names a leaf: ~ref = (first leaf slot << 4) | (triangles - 1) as built; on the device a triangle leaf counts
// pair slots instead (see SceneDev::bvh_pairs).
#ifndef WBRT_BVH_STACK
#define WBRT_BVH_STACK 24
#endif
struct WbrtBvhNode
{
	float lmin[3], lmax[3], rmin[3], rmax[3];
	int left, right;
	float le, re; // max |e0||e1| over the triangles below each child
};
struct WbrtBvhRoot
{
	float mn[3], mx[3], e;
	int ref, empty;
};
struct WbrtBvh
{
	std::vector<WbrtBvhNode> nodes;
	std::vector<int> tri_index; // leaf slot -> position in the grouped triangle order
	std::vector<WbrtBvhRoot> roots;
	// 12 per node (left lo.xyz hi.xyz, right lo.xyz hi.xyz): box around (e0 x e1)/(|e0||e1|) of the triangles below each
	// child, each vector flipped into the child's own hemisphere; [-1,1]^3 where nothing is known
	std::vector<float> mbox;
	int depth = 0;
};
void wbrt_build_bvh(const float4 *tri_geom, const int2 *ranges, int B, bool split_by_e, WbrtBvh &bvh);
bool wbrt_build_box_tree(const float4 *box_min, const float4 *box_max, const int2 *ranges, int B,
                         std::vector<float4> &nodes, std::vector<int> &box_of_rank, int &depth);
void wbrt_append_sphere_tree(const float4 *sph_geom, const float *radius, int S, WbrtBvh &bvh);

// Schedule arrays of one render launch (pilot costs, tile costs, launch order, histogram): either
// buffers the context allocated, or slices of the library's static device array (the entry point
// that allocates nothing).
struct WbrtSchedule
{
	int *costs;        // pixels + pixels/64 + 1
	int *tile_costs;   // pixels/64 + 1
	int *order;        // pixels/64 + 1
	unsigned *hist;    // 256
};
// Largest launch (pixels of the frame) the static array serves, and its size in bytes.
#define WBRT_NOALLOC_MAX_PIXELS (1 << 21)
#define WBRT_ALIGN256(n) (((size_t)(n) + 255) & ~(size_t)255)
#define WBRT_NOALLOC_COSTS_BYTES WBRT_ALIGN256(((size_t)WBRT_NOALLOC_MAX_PIXELS + WBRT_NOALLOC_MAX_PIXELS / 64 + 1) * sizeof(int))
#define WBRT_NOALLOC_TILES_BYTES WBRT_ALIGN256(((size_t)WBRT_NOALLOC_MAX_PIXELS / 64 + 1) * sizeof(int))
#define WBRT_NOALLOC_SCRATCH_BYTES (WBRT_NOALLOC_COSTS_BYTES + 2 * WBRT_NOALLOC_TILES_BYTES + 256 * sizeof(unsigned))
// Looks up this device's copy of the static array (once per device, with that device current).
hipError_t wbrt_scratch_init(int device);
// The schedule arrays inside this device's copy; false before wbrt_scratch_init(device) succeeded.
bool wbrt_scratch_schedule(int device, WbrtSchedule &s);

hipError_t wbrt_launch_render(const RenderParams &p, hipStream_t stream);
hipError_t wbrt_launch_count(const RenderParams &p, hipStream_t stream);
hipError_t wbrt_launch_ids(const RenderParams &p, hipStream_t stream);
// walk: per-tile rounds of the hierarchy walk (mesh scenes), or null
// exact: the counts come from a full launch of the same view (not from a pilot launch)
hipError_t wbrt_launch_cost_sort(const int *costs, const int *walk, int n, int pilot_spp, int exact, int *tile_costs, unsigned *hist, int *order, hipStream_t stream);
hipError_t wbrt_launch_tonemap(const float *image, float scale, int width, int height, uint8_t *rgb8, hipStream_t stream);
// seeds[i] = the i-th seed of the SplitMix64 stream that starts at state0 (see wbrt_make_seeds)
hipError_t wbrt_launch_make_seeds(unsigned long long state0, uint32_t *seeds, size_t n, hipStream_t stream);
// evaluates the device's float / double primitives on arrays (known-answer tests)
hipError_t wbrt_launch_math_probe(int op, const float *a, const float *b, float *out, int n, hipStream_t stream);

#endif
