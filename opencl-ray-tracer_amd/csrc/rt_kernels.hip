// HIP kernels of the path tracer for gfx950 (MI355X).  Compile with -ffp-contract=off:
// every float result that can decide a hit or feed the image is produced by the same
// IEEE binary32 operations, in the same order, as the reference kernel
// (src/opencl/gpu_kernel.cl), so results match a non-contracting CPU build bit for bit.
//
// Design (not a translation of the reference's one-work-item-per-pixel nested loops):
//   * one lane owns one pixel's random stream (the stream is sequential per pixel,
//     gpu_kernel.cl:606), but the sample loop and the bounce loop are flattened into ONE
//     loop whose body is "one ray segment": lanes whose path ended accumulate, regenerate
//     the next camera ray of the same stream in place and rejoin, so the intersection
//     phase always runs with every live lane of the wave;
//   * the closest hit is kept as (t, primitive id); hit point, normal and material are
//     derived once per segment for the winner only;
//   * primitives are repacked on upload (SceneDev): sphere centre + radius^2, triangle
//     p0 + precomputed edges, triangles grouped per bounding box;
//   * mesh scenes and scenes with many spheres: hierarchies built on upload, walked by the
//     whole wave with work handed between lanes (triangles_steal), hits merged by a 64-bit
//     atomic min whose key order is the reference's tie rule;
//   * waves are launched costliest first; the costs are the ray-segment counts the previous launch of
//     the same view recorded, or those of a pilot launch with a few samples per pixel (a frame lasts
//     as long as its longest pixel stream).
#include "rt_device.h"

#include <math.h>
#include <mutex>

namespace {

// the strict build of the render code (see rt_render.inc)
#pragma clang fp contract(off)
#define WBRT_TOL 0
namespace strict_impl {
#include "rt_render.inc"
} // namespace strict_impl
#undef WBRT_TOL
// the tolerant build
#pragma clang fp contract(fast)
#define WBRT_TOL 1
namespace tolerant_impl {
#include "rt_render.inc"
} // namespace tolerant_impl
#undef WBRT_TOL
#pragma clang fp contract(off)
using namespace strict_impl;

// Primary-hit primitive id of sample 0's first segment, in the caller's numbering.
__global__ __launch_bounds__(256) void k_primary_ids(const RenderParams p)
{
	__shared__ TriTiles s_tiles[4];
	Counters<false> ctr;
	long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	int gid, bx, by;
	if (!locate(p, w, gid, bx, by)) return;
	Lane L;
	init_lane(p, L, gid, bx, by);
	camera_ray<false>(p, L, ctr);
	float t;
	int id;
	closest_sphere<false>(p.sc, L.o, L.d, t, id, ctr);
	if (p.sc.B > 0) boxes_and_triangles<false>(p.sc, s_tiles[threadIdx.x >> 6], L.o, L.d, t, id, ctr);
	if (id >= p.sc.S) id = p.sc.S + p.sc.tri_orig[id - p.sc.S];
	const int row = gid / p.bsx + p.bsy * by, colx = gid % p.bsx + p.bsx * bx;
	p.ids[(size_t)row * p.out_w + colx] = id;
}

// colour.cpp:5-15 per channel; output rows top-down, R,G,B.
__global__ void k_tonemap(const float *image, float scale, int width, int height, uint8_t *rgb8)
{
	const int n = width * height * 3;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		const int row = i / (width * 3), rest = i - row * (width * 3);
		const float c = image[(size_t)(height - 1 - row) * width * 3 + rest] * scale;
		const float g = (float)(sqrt((double)c) * (double)255.0f);
		rgb8[i] = (uint8_t)g;
	}
}

// Seed buffer of one shard, generated where it is used: the host generator (wbrt_make_seeds) draws from a SplitMix64
// stream, whose i-th output depends only on state0 + (i+1) * golden, so every seed is computed independently here.
__global__ void k_make_seeds(unsigned long long state0, uint32_t *seeds, unsigned n)
{
	const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	unsigned long long z = state0 + (unsigned long long)(i + 1u) * 0x9E3779B97F4A7C15ULL;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	z ^= z >> 31;
	const unsigned long long r = z >> 33; // 31 bits, like a POSIX rand()
	seeds[i] = (uint32_t)(r * 0x5DEECE66DULL + 0xBULL); // main.cpp:151: the low 32 bits
}

// op: 0 a/b, 1 sqrtf(a), 2 atan2(a,b), 3 asin(a), 4 a^5, 5 rng float of seed bits a (32-bit) after b draws,
// 6 shared-reciprocal division a/b, 7/8 sky texel index (fast with fallback / exact), 9 division by a constant
__global__ void k_math_probe(int op, const float *a, const float *b, float *out, int n)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	float x = a[i], y = b ? b[i] : 0.0f, r = 0.0f;
	switch (op)
	{
	case 0: r = x / y; break;
	case 1: r = sqrtf(x); break;
	case 2: r = pt_atan2f(x, y); break;
	case 3: r = pt_asinf(x); break;
	case 4: r = pt_pow5f(x); break;
	case 5:
	{
		Rng g; g.seed(__float_as_uint(x));
		int k = (int)__float_as_uint(y);
		for (int j = 0; j < k; ++j) r = g.next();
		break;
	}
	case 6: r = div3(f3(x, x, x), y).z; break;
	case 9: r = div_by_const(x, y, 1.0f / y); break;
	case 7: // sky texel of direction (a, b, 0.25 - a) in a 64x32 map: fast path with exact fallback
	case 8: // same, exact path only
	{
		SceneDev sc{};
		sc.skyw = 64; sc.skyh = 32;
		F3 d = f3(x, y, 0.25f - x);
		int idx = 0;
		if (op == 8 || !sky_texel_fast(sc, d, idx))
		{
			F3 n = unit(d);
			float u = 0.5f + pt_atan2f(n.z, n.x) / (2.0f * 3.14159274101257324f);
			float v = 0.5f - pt_asinf(n.y) / 3.14159274101257324f;
			float fx = u * 64.0f, fy = v * 32.0f;
			int xi = (fx >= 0.0f && fx <= 64.0f) ? (int)fx : 0;
			int yi = (fy >= 0.0f && fy <= 32.0f) ? (int)fy : 0;
			idx = yi * 64 + xi;
			if (op == 7) idx += 1 << 20; // marks a fallback
		}
		r = (float)idx;
		break;
	}
	}
	out[i] = r;
}

// Cost of a wave's 64 work indices, scaled into 0..255: the sum of their iteration counts
// plus 192 x the largest of them when the counts are a previous launch's exact ones (64 x: 1.3 % slower
// frame, 448 x: 0.7 % slower), plus 64 x the largest for the noisier counts of a pilot launch.  A wave lasts as
// long as its LONGEST pixel stream; tiles with a few long streams among cheap ones (an object's edge against the sky) have a small sum, and
// ranked by the sum alone they started late and were the last to finish (weights 16..512 tried).
__global__ void k_tile_cost(const int *costs, const int *walk, int n, int n_tiles, int shift, int exact, int *tile_costs)
{
	const int tile = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
	const int i = tile * 64 + (threadIdx.x & 63);
	int c = (tile < n_tiles && i < n) ? costs[i] : 0;
	int mx = c;
	for (int off = 32; off > 0; off >>= 1)
	{
		c += __shfl_down(c, off, 64);
		const int other = __shfl_down(mx, off, 64);
		mx = other > mx ? other : mx;
	}
	// mesh scenes, where the walk is shared by the wave: sum + 3 per round of the walk (measured best)
	if ((threadIdx.x & 63) == 0 && tile < n_tiles)
		tile_costs[tile] = walk ? (c + 3 * walk[tile]) >> shift : exact ? (c + 192 * mx) >> (shift + 2) : (c + 64 * mx) >> (shift + 1);
}

// Counting sort of indices by cost, descending (256 buckets).
__global__ void k_cost_hist(const int *costs, int n, unsigned *hist)
{
	__shared__ unsigned local[256];
	local[threadIdx.x] = 0;
	__syncthreads();
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
		atomicAdd(&local[min(costs[i], 255)], 1u);
	__syncthreads();
	if (local[threadIdx.x]) atomicAdd(&hist[threadIdx.x], local[threadIdx.x]);
}

__global__ void k_cost_scan(unsigned *hist) // one thread: bucket 255 first
{
	unsigned run = 0;
	for (int b = 255; b >= 0; --b) { unsigned c = hist[b]; hist[b] = run; run += c; }
}

// Each block ranks a contiguous chunk in LDS and reserves one range per bucket in the global offsets.
__global__ void k_cost_scatter(const int *costs, int n, unsigned *offsets, int *order)
{
	__shared__ unsigned count[256], first[256];
	const int per = (n + gridDim.x - 1) / gridDim.x;
	const int begin = blockIdx.x * per, end = min(n, begin + per);
	count[threadIdx.x] = 0;
	__syncthreads();
	for (int i = begin + threadIdx.x; i < end; i += blockDim.x) atomicAdd(&count[min(costs[i], 255)], 1u);
	__syncthreads();
	first[threadIdx.x] = count[threadIdx.x] ? atomicAdd(&offsets[threadIdx.x], count[threadIdx.x]) : 0u;
	count[threadIdx.x] = 0;
	__syncthreads();
	for (int i = begin + threadIdx.x; i < end; i += blockDim.x)
	{
		const int b = min(costs[i], 255);
		order[first[b] + atomicAdd(&count[b], 1u)] = i;
	}
}

inline unsigned grid_for(const RenderParams &p, int block)
{
	long long total = (long long)p.bsx * p.bsy * p.nbx * p.nby;
	if (p.n_shards > 1) total <<= p.shard_lanes_log2; // fused seed shards: that many lanes per pixel
	return (unsigned)((total + block - 1) / block);
}

} // namespace

// Schedule arrays of the entry point that allocates nothing (wbrt_render_frame_device_noalloc): one
// file-scope device array per GPU, held until the process ends.  Every launch on a GPU that goes
// through that entry point shares it.  On the host the array's name only stands in for the symbol:
// the device address is looked up once per device and used from then on.
static __device__ __align__(256) unsigned char scratch[WBRT_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[kMaxDevices] = {};
} // namespace

hipError_t wbrt_scratch_init(int device)
{
	if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
	std::lock_guard<std::mutex> lock(g_scratch_mutex);
	if (g_scratch_ptr[device]) return hipSuccess;
	int current = -1;
	hipError_t e = hipGetDevice(&current);
	if (e != hipSuccess) return e;
	if (current != device) return hipErrorInvalidDevice; // the address belongs to the CURRENT device's copy
	void *ptr = nullptr;
	e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
	if (e != hipSuccess) return e;
	if (!ptr) return hipErrorInvalidSymbol;
	g_scratch_ptr[device] = (unsigned char *)ptr;
	return hipSuccess;
}

bool wbrt_scratch_schedule(int device, WbrtSchedule &s)
{
	if (device < 0 || device >= kMaxDevices) return false;
	unsigned char *base;
	{
		std::lock_guard<std::mutex> lock(g_scratch_mutex);
		base = g_scratch_ptr[device];
	}
	if (!base) return false;
	s.costs = (int *)base;
	s.tile_costs = (int *)(base + WBRT_NOALLOC_COSTS_BYTES);
	s.order = (int *)(base + WBRT_NOALLOC_COSTS_BYTES + WBRT_NOALLOC_TILES_BYTES);
	s.hist = (unsigned *)(base + WBRT_NOALLOC_COSTS_BYTES + 2 * WBRT_NOALLOC_TILES_BYTES);
	return true;
}

hipError_t wbrt_launch_render(const RenderParams &p, hipStream_t stream)
{
	const dim3 grid(grid_for(p, WBRT_RENDER_BLOCK)), block(WBRT_RENDER_BLOCK);
	const bool box_tree = p.sc.top_nodes && p.sc.bvh_nodes, mesh = p.sc.B > 0 || (p.sc.sph_tree && p.sc.bvh_nodes);
	if (p.wg && !mesh && p.spp > 0 && p.next && p.n_shards <= 1)
	{
		// opt-in form (measured slower than k_render on the headline frame; kept selectable for that measurement):
		// persistent workgroups of p.wg waves, as many as the device holds at once (or fewer, for a small launch)
		int device = -1, cus = 0;
		hipError_t e = hipGetDevice(&device);
		if (e != hipSuccess) return e;
		e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
		if (e != hipSuccess) return e;
		const int nt = p.wg >= 8 ? 512 : 256;
		const int per_cu = p.wg_per_cu > 0 ? p.wg_per_cu : (nt == 512 ? 3 : 6);
		const long long slots = (long long)cus * per_cu;
		const long long total = (long long)p.bsx * p.bsy * p.nbx * p.nby;
		const long long groups = (total + nt - 1) / nt;
		const dim3 wg_grid((unsigned)(groups < slots ? groups : slots)), wg_block(nt);
		e = hipMemsetAsync(p.next, 0, sizeof(unsigned), stream);
		if (e != hipSuccess) return e;
		if (nt == 512)
		{
			if (p.tolerant) hipLaunchKernelGGL(tolerant_impl::k_render_wg<512>, wg_grid, wg_block, 0, stream, p);
			else hipLaunchKernelGGL(strict_impl::k_render_wg<512>, wg_grid, wg_block, 0, stream, p);
		}
		else
		{
			if (p.tolerant) hipLaunchKernelGGL(tolerant_impl::k_render_wg<256>, wg_grid, wg_block, 0, stream, p);
			else hipLaunchKernelGGL(strict_impl::k_render_wg<256>, wg_grid, wg_block, 0, stream, p);
		}
		return hipGetLastError();
	}
	if (p.n_shards > 1)
	{
		// fused seed shards (accumulating launches of scenes without a hierarchy over the boxes; the caller checks)
		if (box_tree || p.out_mode != 2 || !p.out || p.costs || p.shard_lanes_log2 < 0 || p.shard_lanes_log2 > 6 ||
		    (p.shard_lanes_log2 > 0 && p.n_shards > (1 << p.shard_lanes_log2))) return hipErrorInvalidValue;
		if (((long long)p.bsx * p.bsy * p.nbx * p.nby << p.shard_lanes_log2) > 0xFFFFFFFFLL * 64) return hipErrorInvalidValue;
		if (p.tolerant)
		{
			if (mesh) hipLaunchKernelGGL((tolerant_impl::k_render<false, true, false, true>), grid, block, 0, stream, p);
			else hipLaunchKernelGGL((tolerant_impl::k_render<false, false, false, true>), grid, block, 0, stream, p);
		}
		else
		{
			if (mesh) hipLaunchKernelGGL((strict_impl::k_render<false, true, false, true>), grid, block, 0, stream, p);
			else hipLaunchKernelGGL((strict_impl::k_render<false, false, false, true>), grid, block, 0, stream, p);
		}
		return hipGetLastError();
	}
	if (p.tolerant)
	{
		if (box_tree) hipLaunchKernelGGL((tolerant_impl::k_render<false, true, true>), grid, block, 0, stream, p);
		else if (mesh) hipLaunchKernelGGL((tolerant_impl::k_render<false, true>), grid, block, 0, stream, p);
		else hipLaunchKernelGGL((tolerant_impl::k_render<false, false>), grid, block, 0, stream, p);
	}
	else
	{
		if (box_tree) hipLaunchKernelGGL((strict_impl::k_render<false, true, true>), grid, block, 0, stream, p);
		else if (mesh) hipLaunchKernelGGL((strict_impl::k_render<false, true>), grid, block, 0, stream, p);
		else hipLaunchKernelGGL((strict_impl::k_render<false, false>), grid, block, 0, stream, p);
	}
	return hipGetLastError();
}

hipError_t wbrt_launch_count(const RenderParams &p, hipStream_t stream)
{
	const dim3 grid(grid_for(p, WBRT_RENDER_BLOCK)), block(WBRT_RENDER_BLOCK);
	if (p.sc.B > 0) hipLaunchKernelGGL((strict_impl::k_render<true, true>), grid, block, 0, stream, p);
	else hipLaunchKernelGGL((strict_impl::k_render<true, false>), grid, block, 0, stream, p);
	return hipGetLastError();
}

hipError_t wbrt_launch_cost_sort(const int *costs, const int *walk, int n, int pilot_spp, int exact, int *tile_costs, unsigned *hist, int *order, hipStream_t stream)
{
	const int n_tiles = (n + 63) / 64;
	int shift = 0; // a tile whose 64 streams average 16 segments per sample lands in the last bucket
	while (((64 * 16 * pilot_spp) >> shift) > 255) ++shift;
	hipError_t e = hipMemsetAsync(hist, 0, 256 * sizeof(unsigned), stream);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(k_tile_cost, dim3((n_tiles + 3) / 4), dim3(256), 0, stream, costs, walk, n, n_tiles, shift, exact, tile_costs);
	hipLaunchKernelGGL(k_cost_hist, dim3(64), dim3(256), 0, stream, (const int *)tile_costs, n_tiles, hist);
	hipLaunchKernelGGL(k_cost_scan, dim3(1), dim3(1), 0, stream, hist);
	hipLaunchKernelGGL(k_cost_scatter, dim3(64), dim3(256), 0, stream, (const int *)tile_costs, n_tiles, hist, order);
	return hipGetLastError();
}

hipError_t wbrt_launch_ids(const RenderParams &p, hipStream_t stream)
{
	hipLaunchKernelGGL(k_primary_ids, dim3(grid_for(p, 256)), dim3(256), 0, stream, p);
	return hipGetLastError();
}

hipError_t wbrt_launch_tonemap(const float *image, float scale, int width, int height, uint8_t *rgb8, hipStream_t stream)
{
	hipLaunchKernelGGL(k_tonemap, dim3(2048), dim3(256), 0, stream, image, scale, width, height, rgb8);
	return hipGetLastError();
}

hipError_t wbrt_launch_make_seeds(unsigned long long state0, uint32_t *seeds, size_t n, hipStream_t stream)
{
	if (!n) return hipSuccess;
	hipLaunchKernelGGL(k_make_seeds, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, state0, seeds, (unsigned)n);
	return hipGetLastError();
}

hipError_t wbrt_launch_math_probe(int op, const float *a, const float *b, float *out, int n, hipStream_t stream)
{
	hipLaunchKernelGGL(k_math_probe, dim3((n + 255) / 256), dim3(256), 0, stream, op, a, b, out, n);
	return hipGetLastError();
}

#ifdef WBRT_WALK_STATS
// Instrumented build only: reads (and clears) the walk counters of rt_render.inc.
extern "C" void wbrt_dbg_rounds_(unsigned long long *out, int reset)
{
	(void)hipDeviceSynchronize();
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(strict_impl::g_walk_stats), 8 * sizeof(unsigned long long));
	if (reset) { unsigned long long z[8] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(strict_impl::g_walk_stats), z, sizeof z); }
}
#endif

#ifdef WBRT_PHASE_STATS
// Instrumented build only: reads (and clears) the phase counters of rt_render.inc.
extern "C" void wbrt_dbg_phases_(unsigned long long *out, int reset)
{
	(void)hipDeviceSynchronize();
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(strict_impl::g_phase), 32 * sizeof(unsigned long long));
	if (reset) { unsigned long long z[32] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(strict_impl::g_phase), z, sizeof z); }
}
#endif
