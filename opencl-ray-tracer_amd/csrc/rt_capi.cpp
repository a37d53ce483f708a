// Device half of the C ABI (include/ocl_rt_amd.h): context, uploads with repacking,
// launches, read-back.  Replaces the OpenCL host calls of the reference's main()
// (src/cpp/main.cpp:25-77,160-205,229-247); each entry point cites its lines in the header.
#include "../../include/ocl_rt_amd.h"
#include "rt_device.h"

#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

void wbrt_set_error_(const char *msg);
extern "C" uint64_t wbrt_seed_stream_state_(uint64_t base_seed, uint32_t shard);

namespace {

int fail(int code, const std::string &msg)
{
	wbrt_set_error_(msg.c_str());
	return code;
}

#define HIP_TRY(expr)                                                                       \
	do {                                                                                    \
		hipError_t e_ = (expr);                                                             \
		if (e_ != hipSuccess)                                                               \
			return fail(e_ == hipErrorOutOfMemory ? WBRT_ERR_OOM : WBRT_ERR_HIP,           \
			            std::string(#expr) + ": " + hipGetErrorName(e_) + " (" + hipGetErrorString(e_) + ")"); \
	} while (0)

struct DevBuf
{
	void *p = nullptr;
	size_t bytes = 0;
	int upload(const void *src, size_t n)
	{
		if (n > bytes || !p)
		{
			if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
			size_t cap = n ? n : 16;
			HIP_TRY(hipMalloc(&p, cap));
			bytes = cap;
		}
		if (n) HIP_TRY(hipMemcpy(p, src, n, hipMemcpyHostToDevice));
		return WBRT_OK;
	}
	int reserve(size_t n)
	{
		if (n <= bytes && p) return WBRT_OK;
		if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
		HIP_TRY(hipMalloc(&p, n ? n : 16));
		bytes = n ? n : 16;
		return WBRT_OK;
	}
	void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

} // namespace

// What the recorded per-pixel costs of a launch depend on (apart from seeds and sample count).
struct ViewKey
{
	float cam[18];
	float aperture;
	int dims[11];
	unsigned long long scene_generation;
	int walks;
};

struct wbrt_context
{
	int device = 0;
	hipStream_t stream = nullptr; // used by the blocking entry points
	DevBuf sph_geom, sph_radius, sph_mat_a, sph_mat_b;
	DevBuf tri_geom, tri_shade, mat_table, tri_orig;
	DevBuf box_min, box_max, box_range;
	DevBuf top_nodes, top_box;    // hierarchy over the boxes (scenes with box_tree_min or more boxes)
	int top_depth = 0;            // 0: no such hierarchy
	int box_tree_min = 48;        // measured crossover between 36 and 64 boxes
	DevBuf bvh_nodes[2], bvh_tris[2], bvh_roots[2], bvh_pairs[2]; // [0] plain surface-area trees (capped margins), [1] built for the proven margins
	DevBuf bvh_mbox;              // m-boxes of the nodes of variant [1]
	const float4 *bvh_ptr[2][3] = {};
	int bvh_depth[2] = {0, 0};
	int mesh_mode = WBRT_MESH_BVH;
	int bvh_steal_min_idle = 8;
	int fuse_shards = 1;          // wbrt_render_shards renders a rank's shards in one launch where their sample counts allow
	int shard_lanes_max_pixels = 1 << 20; // ... as concurrent lanes per pixel up to this many pixels, in turn in one lane beyond
	int wg_kernel = 0, wg_per_cu = 0; // opt-in: waves (4 or 8) of the workgroup render kernel for scenes without meshes, and its workgroups per CU
	int strict_loop_max = 256;    // proven-margin mode: meshes up to this many triangles use the cooperative loop (measured crossover:
	                              // 256 triangles 15.7 vs 15.1 ms, 128: 6.1 vs 9.1 ms, 512: 31.6 vs 21.8 ms, loop vs walk)
	float bvh_rho = 2e-4f; // WBRT_MESH_BVH_CAPPED: box margin as a fraction of the ray's distance bound
	DevBuf sky, seeds, image, ids, counters, rgb8, costs, tile_costs, order, hist;
	int lpt = 1;        // longest-first scheduling from a pilot pass
	int lpt_min_stream_spp = 1; // launches whose longest stream has fewer samples go unscheduled (tune key "lpt_min_spp")
	int pilot_spp = 4;
	int pilot_depth = 8;          // bounce limit of a pilot launch: its counts only rank tiles, and an unlimited pilot lasts as long as its longest
	                              // stream (headline cold frame 10.87 ms without a limit, 10.52 / 10.43 / 10.43 ms at 16 / 8 / 5); 0: no limit
	int reuse_costs = 1;          // schedule a launch from the counts its predecessor recorded (same view)
	int recorded_spp = 0;         // 0: nothing recorded
	ViewKey recorded_view = {};
	unsigned long long scene_generation = 0;
	SceneDev sc{};
	bool have_scene = false, have_sky = false, have_bvh = false, have_sphere_tree = false;
	bool mesh_in_range = true; // every finite triangle coordinate lies below 2^40: no product of the reference's triangle test can overflow
	int sphere_tree_min = 48;  // sphere count from which the render build walks a sphere hierarchy
	size_t n_seeds = 0;
	bool timing = false;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> events; // pool; the first events_used pairs hold launches
	size_t events_used = 0;
	hipEvent_t render_done = nullptr; // recorded after every render launch: the launch's use of the context's buffers ends there
	bool render_busy = false;
	hipStream_t render_stream = nullptr; // the stream render_done was last recorded on
	hipEvent_t seeds_ready = nullptr; // recorded after a device-side seed generation: launches on other streams wait for it
	bool seeds_pending = false;
	hipStream_t seeds_stream = nullptr;
	int arith_mode = 0;
};

namespace {

int bind(wbrt_context *ctx)
{
	if (!ctx) return fail(WBRT_ERR_INVALID, "null context");
	HIP_TRY(hipSetDevice(ctx->device));
	return WBRT_OK;
}

// Uploads replace buffers a render launch of this context may still read: wait for that launch only (other
// contexts and streams of the device keep running).
int wait_renders(wbrt_context *ctx)
{
	// every launch waits for its predecessor on another stream (await_inputs), so the last event covers all of them
	if (ctx->render_busy) HIP_TRY(hipEventSynchronize(ctx->render_done));
	if (ctx->seeds_pending) HIP_TRY(hipEventSynchronize(ctx->seeds_ready));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	ctx->render_busy = false;
	ctx->seeds_pending = false;
	return WBRT_OK;
}

// Before a launch that reads the context's buffers is enqueued on `stream`: the seeds a device-side generation is still
// writing on another stream, and the previous render launch of this context on another stream (it shares the seed buffer
// and, when scheduled, the schedule arrays; an upload that follows waits for the LAST launch only).
int await_inputs(wbrt_context *ctx, hipStream_t stream)
{
	if (ctx->seeds_pending && ctx->seeds_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, ctx->seeds_ready, 0));
	if (ctx->render_busy && ctx->render_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, ctx->render_done, 0));
	return WBRT_OK;
}

// After a device-side seed generation was enqueued on `stream`.
int seeds_generated(wbrt_context *ctx, hipStream_t stream)
{
	HIP_TRY(hipEventRecord(ctx->seeds_ready, stream));
	ctx->seeds_pending = true;
	ctx->seeds_stream = stream;
	return WBRT_OK;
}

// Whether render launches of this context walk the hierarchies built on upload (else: the loops over boxes and triangles).
bool walks_hierarchies(const wbrt_context *ctx)
{
	// strict margins on a small mesh widen every box to the whole mesh: the cooperative loop is faster there
	const bool strict_small = ctx->mesh_mode == WBRT_MESH_BVH && ctx->sc.T <= ctx->strict_loop_max && !ctx->have_sphere_tree;
	// (meshes with coordinates beyond 2^40 keep the loops: the margins of the walk assume that the reference's float
	// arithmetic, gpu_kernel.cl:489-520, does not overflow)
	return ctx->mesh_mode != WBRT_MESH_LOOP && ctx->have_bvh && !strict_small && ctx->mesh_in_range;
}

// Fills launch parameters from the camera; validates what the kernels assume.
int make_params(wbrt_context *ctx, const wbrt_camera *cam, bool whole_frame, RenderParams &p, bool check_seeds = true)
{
	if (!cam) return fail(WBRT_ERR_INVALID, "null camera");
	if (!ctx->have_scene) return fail(WBRT_ERR_INVALID, "no scene uploaded (wbrt_upload_scene)");
	if (!ctx->have_sky) return fail(WBRT_ERR_INVALID, "no sky uploaded (wbrt_upload_sky)");
	const int bsx = (int)cam->blockSize.x, bsy = (int)cam->blockSize.y; // gpu_kernel.cl:618-619
	if (bsx <= 0 || bsy <= 0) return fail(WBRT_ERR_INVALID, "camera.blockSize must be positive");
	if (cam->width <= 0 || cam->height <= 0) return fail(WBRT_ERR_INVALID, "camera width/height must be positive");
	if (cam->samplesPerPixel < 0) return fail(WBRT_ERR_INVALID, "samplesPerPixel must be >= 0");
	if (check_seeds && (size_t)bsx * (size_t)bsy > ctx->n_seeds)
		return fail(WBRT_ERR_INVALID, "seed buffer smaller than one block (wbrt_upload_seeds)");
	if ((long long)cam->width * cam->height > (1LL << 30) || cam->width > (1 << 23) || cam->height > (1 << 23))
		return fail(WBRT_ERR_INVALID, "image too large");
	std::memset(&p, 0, sizeof p);
	const wbrt_vec3 *src[6] = { &cam->origin, &cam->horizontal, &cam->vertical, &cam->lowerLeftCorner,
	                            &cam->defocusDiscU, &cam->defocusDiscV };
	float *dst[6] = { p.org, p.hor, p.ver, p.llc, p.ddu, p.ddv };
	for (int i = 0; i < 6; ++i) { dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z; }
	p.aperture = cam->aperture;
	p.width = cam->width; p.height = cam->height;
	p.inv_width = (float)(1.0 / (double)cam->width); p.inv_height = (float)(1.0 / (double)cam->height);
	p.spp = cam->samplesPerPixel; p.max_depth = cam->maxDepth;
	p.bsx = bsx; p.bsy = bsy;
	if (whole_frame)
	{
		if (cam->width % bsx || cam->height % bsy)
			return fail(WBRT_ERR_INVALID, "blockSize must divide the image (main.cpp:222-223)");
		p.nbx = cam->width / bsx; p.nby = cam->height / bsy;
		p.off_bx = 0; p.off_by = 0;
		p.out_w = cam->width;
	}
	else
	{
		p.nbx = 1; p.nby = 1;
		p.off_bx = (int)cam->blockOffset.x; p.off_by = (int)cam->blockOffset.y;
		p.out_w = bsx;
	}
	{
		// columns bsx * (block + offset) + in-block column, rows likewise (gpu_kernel.cl:626-627, evaluated in float)
		const long long c0 = (long long)p.bsx * p.off_bx, c1 = (long long)p.bsx * (p.off_bx + p.nbx);
		const long long r0 = (long long)p.bsy * p.off_by, r1 = (long long)p.bsy * (p.off_by + p.nby);
		p.plain_coords = c0 >= 0 && r0 >= 0 && c1 <= (1 << 23) && r1 <= (1 << 23) && cam->width <= (1 << 23) && cam->height <= (1 << 23) ? 1 : 0;
	}
	p.seeds = (const uint32_t *)ctx->seeds.p;
	p.tolerant = ctx->arith_mode == WBRT_ARITH_TOLERANT ? 1 : 0;
	p.wg = ctx->wg_kernel;
	p.wg_per_cu = ctx->wg_per_cu;
	p.sc = ctx->sc;
	if (walks_hierarchies(ctx))
	{
		const int variant = ctx->mesh_mode == WBRT_MESH_BVH ? 1 : 0;
		p.sc.bvh_nodes = ctx->bvh_ptr[variant][0];
		p.sc.bvh_tris = ctx->bvh_ptr[variant][1];
		p.sc.bvh_roots = ctx->bvh_ptr[variant][2];
		p.sc.bvh_pairs = (const float4 *)ctx->bvh_pairs[variant].p;
		p.sc.bvh_mbox = variant == 1 ? (const uint4 *)ctx->bvh_mbox.p : nullptr;
		p.sc.bvh_box_slots = WBRT_BVH_STACK - ctx->bvh_depth[variant] + 1;
		if (ctx->top_depth > 0)
		{
			// the walk over the boxes uses the top of each lane's stack while hit boxes are queued at its bottom
			p.sc.top_nodes = (const float4 *)ctx->top_nodes.p;
			p.sc.top_box = (const int *)ctx->top_box.p;
			p.sc.bvh_box_slots = std::min(p.sc.bvh_box_slots, WBRT_BVH_STACK - ctx->top_depth);
		}
		p.sc.sph_tree = ctx->have_sphere_tree ? 1 : 0;
	}
	p.sc.bvh_rho = ctx->mesh_mode == WBRT_MESH_BVH_CAPPED ? ctx->bvh_rho : 0.0f;
	p.sc.bvh_steal_min_idle = ctx->bvh_steal_min_idle;
	return WBRT_OK;
}

// Event pairs of the kernel-only timing live in a pool owned by the context: a pair is created once,
// reused after wbrt_get_timing(reset) and destroyed with the context, whatever a launch in between returns.
int timing_begin(wbrt_context *ctx, hipStream_t stream)
{
	if (ctx->events_used == ctx->events.size())
	{
		hipEvent_t a = nullptr, b = nullptr;
		HIP_TRY(hipEventCreate(&a));
		hipError_t e = hipEventCreate(&b);
		if (e != hipSuccess) { (void)hipEventDestroy(a); HIP_TRY(e); }
		ctx->events.emplace_back(a, b);
	}
	HIP_TRY(hipEventRecord(ctx->events[ctx->events_used].first, stream));
	return WBRT_OK;
}

int timing_end(wbrt_context *ctx, hipStream_t stream)
{
	HIP_TRY(hipEventRecord(ctx->events[ctx->events_used].second, stream));
	ctx->events_used++;
	return WBRT_OK;
}

// One render launch, scheduled longest pixel streams first (see k_render): a pilot launch with a few
// samples per pixel (or the counts the previous launch of the same view recorded) prices every pixel
// stream, a counting sort orders the waves, the real launch follows that order.
// fixed == nullptr: the schedule arrays are the context's own buffers (grown on demand) and the launch
// records its counts for the next one.  fixed != nullptr: the arrays are given, nothing is allocated,
// recorded or reused, and no timing events are touched.
int scheduled_launch(wbrt_context *ctx, const RenderParams &p, hipStream_t stream, const WbrtSchedule *fixed)
{
	const bool timing = ctx->timing && !fixed;
	{ int rc = await_inputs(ctx, stream); if (rc) return rc; }
	if (timing) { int rc = timing_begin(ctx, stream); if (rc) return rc; }
	RenderParams q = p;
	const long long total = (long long)p.bsx * p.bsy * p.nbx * p.nby;
	if (ctx->wg_kernel && !fixed)
	{
		// the work counter of the persistent workgroups follows the histogram of the cost sort (uses alternate on one stream)
		int rc = ctx->hist.reserve(257 * sizeof(unsigned));
		if (rc) return rc;
		q.next = (unsigned *)ctx->hist.p + 256;
	}
	// samples per pixel of the whole launch (fused seed shards: all of them)
	const int launch_spp = p.n_shards > 1 ? p.n_shards * p.spp - (p.n_shards - p.n_full) : p.spp;
	// samples of the longest stream of the launch: what a schedule can shorten (fused shards run concurrently)
	const int stream_spp = p.spp;
	// a pilot takes at most a 24th of the launch's samples: short launches (seed shards) get a shorter pilot.  One 31-spp shard
	// of the headline frame, cold: 1.66 ms with a 1-spp pilot, 1.72 / 1.74 / 1.78 ms with 2 / 3 / 4 spp, 2.03 ms unscheduled.
	const int pilot_spp = std::max(1, std::min(ctx->pilot_spp, launch_spp / 24));
	if (ctx->lpt && total >= 65536 && total % 64 == 0 && launch_spp >= 8 && stream_spp >= ctx->lpt_min_stream_spp)
	{
		WbrtSchedule s;
		if (fixed) s = *fixed;
		else
		{
			int rc = ctx->costs.reserve((size_t)(total + total / 64 + 1) * sizeof(int));
			if (!rc) rc = ctx->order.reserve((size_t)(total / 64 + 1) * sizeof(int));
			if (!rc) rc = ctx->tile_costs.reserve((size_t)(total / 64 + 1) * sizeof(int));
			if (!rc) rc = ctx->hist.reserve(257 * sizeof(unsigned));
			if (rc) return rc;
			s.costs = (int *)ctx->costs.p; s.order = (int *)ctx->order.p;
			s.tile_costs = (int *)ctx->tile_costs.p; s.hist = (unsigned *)ctx->hist.p;
			// (the arrays are shared by every launch of this context: await_inputs above made a launch on another
			// stream wait until the previous launch has finished with them)
		}
		// The full launch records the exact counts of every pixel stream; while the view (camera,
		// tiling, depth, scene, mesh mode) stays the same they price the next launch, whatever its
		// seeds or sample count, and no pilot launch is needed.  The schedule never changes a result.
		ViewKey key;
		std::memset(&key, 0, sizeof key);
		std::memcpy(key.cam, p.org, sizeof key.cam);
		key.aperture = p.aperture;
		const int dims[] = { p.width, p.height, p.max_depth, p.bsx, p.bsy, p.nbx, p.nby, p.off_bx, p.off_by, ctx->mesh_mode, ctx->arith_mode };
		static_assert(sizeof dims == sizeof key.dims, "view key");
		std::memcpy(key.dims, dims, sizeof dims);
		key.scene_generation = ctx->scene_generation;
		key.walks = p.sc.bvh_nodes != nullptr;
		int cost_spp = pilot_spp;
		const bool reuse = ctx->reuse_costs && !fixed;
		const bool recorded = reuse && ctx->recorded_spp > 0 && std::memcmp(&key, &ctx->recorded_view, sizeof key) == 0;
		if (recorded) cost_spp = ctx->recorded_spp;
		else
		{
			RenderParams pilot = q;
			pilot.spp = pilot_spp;
			pilot.n_shards = 0; // the first shard's streams price the pixels
			if (ctx->pilot_depth > 0 && ctx->pilot_depth < pilot.max_depth) pilot.max_depth = ctx->pilot_depth;
			pilot.costs = s.costs;
			pilot.out = nullptr;
			HIP_TRY(wbrt_launch_render(pilot, stream));
		}
		HIP_TRY(wbrt_launch_cost_sort(s.costs, key.walks ? s.costs + total : nullptr, (int)total, cost_spp, recorded ? 1 : 0, s.tile_costs,
		                              s.hist, s.order, stream));
		q.order = s.order;
		if (reuse)
		{
			q.record = s.costs;
			std::memcpy(&ctx->recorded_view, &key, sizeof key);
			ctx->recorded_spp = launch_spp;
		}
		HIP_TRY(wbrt_launch_render(q, stream));
	}
	else HIP_TRY(wbrt_launch_render(q, stream));
	HIP_TRY(hipEventRecord(ctx->render_done, stream));
	ctx->render_busy = true;
	ctx->render_stream = stream;
	if (timing) return timing_end(ctx, stream);
	return WBRT_OK;
}

int timed_launch(wbrt_context *ctx, const RenderParams &p, hipStream_t stream)
{
	return scheduled_launch(ctx, p, stream, nullptr);
}

} // namespace

extern "C" {

int wbrt_create(int device_ordinal, wbrt_context **out)
{
	if (!out) return fail(WBRT_ERR_INVALID, "null out pointer");
	*out = nullptr;
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0) return fail(WBRT_ERR_NO_DEVICE, "no HIP device available");
	if (device_ordinal < 0 || device_ordinal >= n) return fail(WBRT_ERR_INVALID, "device ordinal out of range");
	HIP_TRY(hipSetDevice(device_ordinal));
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device_ordinal));
	if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return fail(WBRT_ERR_NO_DEVICE, std::string("kernels are built for gfx950 only, device is ") + prop.gcnArchName);
	wbrt_context *ctx = new wbrt_context();
	ctx->device = device_ordinal;
	e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
	if (e != hipSuccess) { delete ctx; return fail(WBRT_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorName(e)); }
	e = hipEventCreateWithFlags(&ctx->render_done, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->seeds_ready, hipEventDisableTiming);
	if (e != hipSuccess)
	{
		if (ctx->render_done) (void)hipEventDestroy(ctx->render_done);
		(void)hipStreamDestroy(ctx->stream); delete ctx;
		return fail(WBRT_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorName(e));
	}
	// this device's copy of the static schedule array (wbrt_render_frame_device_noalloc)
	e = wbrt_scratch_init(device_ordinal);
	if (e != hipSuccess)
	{
		(void)hipEventDestroy(ctx->render_done); (void)hipEventDestroy(ctx->seeds_ready); (void)hipStreamDestroy(ctx->stream); delete ctx;
		return fail(WBRT_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorName(e));
	}
	*out = ctx;
	return WBRT_OK;
}

int wbrt_destroy(wbrt_context *ctx)
{
	if (!ctx) return WBRT_OK;
	(void)hipSetDevice(ctx->device);
	(void)hipDeviceSynchronize();
	for (auto &ev : ctx->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
	if (ctx->render_done) (void)hipEventDestroy(ctx->render_done);
	if (ctx->seeds_ready) (void)hipEventDestroy(ctx->seeds_ready);
	DevBuf *all[] = { &ctx->sph_geom, &ctx->sph_radius, &ctx->sph_mat_a, &ctx->sph_mat_b, &ctx->tri_geom,
	                  &ctx->tri_shade, &ctx->mat_table, &ctx->tri_orig, &ctx->box_min, &ctx->box_max,
	                  &ctx->box_range, &ctx->bvh_nodes[0], &ctx->bvh_tris[0], &ctx->bvh_roots[0],
	                  &ctx->bvh_nodes[1], &ctx->bvh_tris[1], &ctx->bvh_roots[1], &ctx->bvh_pairs[0], &ctx->bvh_pairs[1], &ctx->bvh_mbox, &ctx->top_nodes, &ctx->top_box, &ctx->sky, &ctx->seeds, &ctx->image, &ctx->ids, &ctx->counters, &ctx->rgb8,
	                  &ctx->costs, &ctx->tile_costs, &ctx->order, &ctx->hist };
	for (DevBuf *b : all) b->release();
	if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
	delete ctx;
	return WBRT_OK;
}

int wbrt_upload_scene(wbrt_context *ctx, const wbrt_sphere *spheres, int32_t S,
                      const wbrt_triangle *tris, int32_t T, const wbrt_bounding_box *boxes, int32_t B)
{
	int rc = bind(ctx);
	if (rc) return rc;
	if (S < 0 || T < 0 || B < 0) return fail(WBRT_ERR_INVALID, "negative primitive count");
	if ((S && !spheres) || (T && !tris) || (B && !boxes)) return fail(WBRT_ERR_INVALID, "null primitive array");
	rc = wait_renders(ctx);
	if (rc) return rc;
	// until every buffer of the new scene is in place the context has no scene (a failed upload leaves it so)
	ctx->have_scene = ctx->have_bvh = ctx->have_sphere_tree = false;
	ctx->mesh_in_range = true;
	ctx->top_depth = 0;
	ctx->recorded_spp = 0;

	// spheres -> SoA; radius*radius is one rounding, as in gpu_kernel.cl:460
	std::vector<float4> geom((size_t)S), mat_a((size_t)S);
	std::vector<float> radius((size_t)S);
	std::vector<float2> mat_b((size_t)S);
	for (int i = 0; i < S; ++i)
	{
		const wbrt_sphere &s = spheres[i];
		volatile float r2 = s.radius * s.radius;
		geom[(size_t)i] = make_float4(s.center.x, s.center.y, s.center.z, r2);
		radius[(size_t)i] = s.radius;
		mat_a[(size_t)i] = make_float4(s.material.albedo.x, s.material.albedo.y, s.material.albedo.z, s.material.fuzz);
		float ty; std::memcpy(&ty, &s.material.type, 4);
		mat_b[(size_t)i] = make_float2(s.material.ior, ty);
	}
	// triangles grouped by the bounding-box INDEX they name (gpu_kernel.cl:380 compares
	// boundingBoxId with the loop index, not with BoundingBox.id), original order kept inside
	// a group (first triangle wins ties, gpu_kernel.cl:520); others are never tested.
	std::vector<int> first((size_t)B + 1, 0);
	for (int j = 0; j < T; ++j)
	{
		int b = tris[j].boundingBoxId;
		if (b >= 0 && b < B) first[(size_t)b + 1]++;
	}
	for (int b = 0; b < B; ++b) first[(size_t)b + 1] += first[(size_t)b];
	const int Tg = B ? first[(size_t)B] : 0;
	std::vector<float4> tgeom((size_t)Tg * 3), tshade((size_t)Tg), mtable;
	std::map<std::array<uint32_t, 6>, int> material_index; // distinct materials by bit pattern
	std::vector<int> torig((size_t)Tg);
	std::vector<int> cursor(first.begin(), first.end());
	for (int j = 0; j < T; ++j)
	{
		const wbrt_triangle &t = tris[j];
		int b = t.boundingBoxId;
		if (b < 0 || b >= B) continue;
		size_t k = (size_t)cursor[(size_t)b]++;
		volatile float e0x = t.p1.x - t.p0.x, e0y = t.p1.y - t.p0.y, e0z = t.p1.z - t.p0.z; // gpu_kernel.cl:491
		volatile float e1x = t.p2.x - t.p0.x, e1y = t.p2.y - t.p0.y, e1z = t.p2.z - t.p0.z; // gpu_kernel.cl:492
		tgeom[3 * k + 0] = make_float4(t.p0.x, t.p0.y, t.p0.z, e0x);
		tgeom[3 * k + 1] = make_float4(e0y, e0z, e1x, e1y);
		tgeom[3 * k + 2] = make_float4(e1z, 0.0f, 0.0f, 0.0f);
		// gpu_kernel.cl:524: outwardNormal = Unit(Cross(edge0, edge1)) = Cross / sqrt(x*x + y*y + z*z), every product, sum,
		// root and quotient rounded once (the same float sequence the kernel would run per hit)
		volatile float cx = e0y * e1z, cx2 = e0z * e1y, cy = e0z * e1x, cy2 = e0x * e1z, cz = e0x * e1y, cz2 = e0y * e1x;
		volatile float nx = cx - cx2, ny = cy - cy2, nz = cz - cz2;
		volatile float xx = nx * nx, yy = ny * ny, zz = nz * nz;
		volatile float sxy = xx + yy;
		volatile float len2 = sxy + zz;
		volatile float len = std::sqrt(len2);
		volatile float ux = nx / len, uy = ny / len, uz = nz / len;
		std::array<uint32_t, 6> key;
		const float mvals[5] = { t.material.albedo.x, t.material.albedo.y, t.material.albedo.z, t.material.fuzz, t.material.ior };
		std::memcpy(key.data(), mvals, 20);
		std::memcpy(&key[5], &t.material.type, 4);
		auto found = material_index.find(key);
		if (found == material_index.end())
		{
			found = material_index.emplace(key, (int)(mtable.size() / 2)).first;
			float ty; std::memcpy(&ty, &t.material.type, 4);
			mtable.push_back(make_float4(mvals[0], mvals[1], mvals[2], mvals[3]));
			mtable.push_back(make_float4(mvals[4], ty, 0.0f, 0.0f));
		}
		float mi; std::memcpy(&mi, &found->second, 4);
		tshade[k] = make_float4(ux, uy, uz, mi);
		torig[k] = j;
		const float coords[9] = { t.p0.x, t.p0.y, t.p0.z, t.p1.x, t.p1.y, t.p1.z, t.p2.x, t.p2.y, t.p2.z };
		for (float c : coords) if (std::isfinite(c) && std::fabs(c) >= 0x1p40f) ctx->mesh_in_range = false;
	}
	std::vector<float4> bmin((size_t)B), bmax((size_t)B);
	std::vector<int2> brange((size_t)B);
	for (int b = 0; b < B; ++b)
	{
		bmin[(size_t)b] = make_float4(boxes[b].min.x, boxes[b].min.y, boxes[b].min.z, 0.0f);
		bmax[(size_t)b] = make_float4(boxes[b].max.x, boxes[b].max.y, boxes[b].max.z, 0.0f);
		brange[(size_t)b] = make_int2(first[(size_t)b], first[(size_t)b + 1]);
	}
#define UP(buf, vec) do { rc = ctx->buf.upload((vec).data(), (vec).size() * sizeof((vec)[0])); if (rc) return rc; } while (0)
	// the kernel fetches small sphere tables as two 64-byte scalar loads: at least 8 entries exist
	std::vector<float4> geom_padded(geom);
	if (geom_padded.size() < 8) geom_padded.resize(8, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
	UP(sph_geom, geom_padded); UP(sph_radius, radius); UP(sph_mat_a, mat_a); UP(sph_mat_b, mat_b);
	UP(tri_geom, tgeom); UP(tri_shade, tshade); UP(mat_table, mtable); UP(tri_orig, torig);
	UP(box_min, bmin); UP(box_max, bmax); UP(box_range, brange);
	ctx->top_depth = 0;
	if (B >= ctx->box_tree_min)
	{
		std::vector<float4> tnodes;
		std::vector<int> trank;
		int tdepth = 0;
		if (wbrt_build_box_tree(bmin.data(), bmax.data(), brange.data(), B, tnodes, trank, tdepth) && tdepth + 2 <= WBRT_BVH_STACK)
		{
			UP(top_nodes, tnodes); UP(top_box, trank);
			ctx->top_depth = tdepth;
		}
	}
	// A hierarchy over the spheres pays off for large sphere counts; it needs finite, moderate
	// coordinates (the kernel's margins assume no overflow in the reference's sphere test).
	bool sphere_tree = S >= ctx->sphere_tree_min;
	for (int i = 0; i < S && sphere_tree; ++i)
	{
		const float v[4] = { spheres[i].center.x, spheres[i].center.y, spheres[i].center.z, spheres[i].radius };
		for (float x : v) if (!(std::fabs(x) < 1e8f)) sphere_tree = false;
	}
	// the margin-checked sphere scan assumes that no intermediate of the reference's sphere test can
	// overflow: coordinates and radii below 1e8 (scenes outside that range take the exact loop)
	bool spheres_moderate = true;
	for (int i = 0; i < S && spheres_moderate; ++i)
	{
		const float v[4] = { spheres[i].center.x, spheres[i].center.y, spheres[i].center.z, spheres[i].radius };
		for (float x : v) if (!(std::fabs(x) < 1e8f)) spheres_moderate = false;
	}
	ctx->sc.sph_moderate = spheres_moderate ? 1 : 0;
	// hierarchies (render build only; the counting and primary-id builds keep the loops)
	for (int variant = 0; variant < 2 && (Tg > 0 || sphere_tree); ++variant)
	{
		WbrtBvh bvh;
		wbrt_build_bvh(tgeom.data(), brange.data(), B, variant == 1, bvh);
		const size_t tri_slots = bvh.tri_index.size();
		if (sphere_tree) wbrt_append_sphere_tree(geom.data(), radius.data(), S, bvh);
		static_assert(sizeof(WbrtBvhNode) == 4 * sizeof(float4), "node layout");
		// Triangle leaves are stored as PAIRS (the kernel tests two triangles per round with packed arithmetic), 5 float4 each:
		// p0.x p0.x' p0.y p0.y' | p0.z p0.z' e0.x e0.x' | e0.y e0.y' e0.z e0.z' | e1.x e1.x' e1.y e1.y' | e1.z e1.z' pos pos'
		// (pos = position in the grouped order; pos' = -1: no second triangle, its numbers repeat the first).  A leaf
		// reference then counts pair slots: ~ref = (first pair << 4) | (pairs - 1).  Sphere leaves keep their one-sphere slots.
		std::vector<float4> bpairs;
		auto as_float = [](int v) { float f; std::memcpy(&f, &v, 4); return f; };
		auto leaf_as_pairs = [&](int ref) {
			if (ref >= 0) return ref;
			const int leaf = ~ref, first_slot = leaf >> 4, count = (leaf & 15) + 1;
			if ((size_t)first_slot >= tri_slots) return ref; // a sphere leaf, or the reference of a box without triangles
			const int base = (int)(bpairs.size() / 5), pairs = (count + 1) / 2;
			for (int k = 0; k < pairs; ++k)
			{
				const int ia = bvh.tri_index[(size_t)first_slot + 2 * k];
				const int ib = 2 * k + 1 < count ? bvh.tri_index[(size_t)first_slot + 2 * k + 1] : -1;
				const float4 *A = &tgeom[3 * (size_t)ia], *Bt = &tgeom[3 * (size_t)(ib >= 0 ? ib : ia)];
				bpairs.push_back(make_float4(A[0].x, Bt[0].x, A[0].y, Bt[0].y));
				bpairs.push_back(make_float4(A[0].z, Bt[0].z, A[0].w, Bt[0].w));
				bpairs.push_back(make_float4(A[1].x, Bt[1].x, A[1].y, Bt[1].y));
				bpairs.push_back(make_float4(A[1].z, Bt[1].z, A[1].w, Bt[1].w));
				bpairs.push_back(make_float4(A[2].x, Bt[2].x, as_float(ia), as_float(ib)));
			}
			return ~((base << 4) | (pairs - 1));
		};
		for (WbrtBvhNode &b : bvh.nodes) { b.left = leaf_as_pairs(b.left); b.right = leaf_as_pairs(b.right); }
		for (WbrtBvhRoot &r : bvh.roots) if (!r.empty) r.ref = leaf_as_pairs(r.ref);
		if (bpairs.empty()) bpairs.resize(5, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
		std::vector<float4> bnodes(bvh.nodes.size() * 4), btris(bvh.tri_index.size() * 3), broots(bvh.roots.size() * 2);
		for (size_t n = 0; n < bvh.nodes.size(); ++n) // children interleaved: the kernel handles them as packed pairs
		{
			const WbrtBvhNode &b = bvh.nodes[n];
			float left, right;
			std::memcpy(&left, &b.left, 4); std::memcpy(&right, &b.right, 4);
			bnodes[4 * n + 0] = make_float4(b.lmin[0], b.rmin[0], b.lmin[1], b.rmin[1]);
			bnodes[4 * n + 1] = make_float4(b.lmin[2], b.rmin[2], b.lmax[0], b.rmax[0]);
			bnodes[4 * n + 2] = make_float4(b.lmax[1], b.rmax[1], b.lmax[2], b.rmax[2]);
			bnodes[4 * n + 3] = make_float4(left, right, b.le, b.re);
		}
		for (size_t k = 0; k < bvh.tri_index.size(); ++k)
		{
			const size_t src = (size_t)bvh.tri_index[k];
			const int pos = bvh.tri_index[k];
			float4 last = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			if (k < tri_slots)
			{
				btris[3 * k + 0] = tgeom[3 * src + 0];
				btris[3 * k + 1] = tgeom[3 * src + 1];
				last = tgeom[3 * src + 2];
			}
			else btris[3 * k + 0] = geom[src]; // sphere slot: center.xyz, radius^2
			std::memcpy(&last.y, &pos, 4);
			btris[3 * k + 2] = last;
		}
		for (size_t b = 0; b < bvh.roots.size(); ++b)
		{
			const WbrtBvhRoot &r = bvh.roots[b];
			float ref; std::memcpy(&ref, &r.ref, 4);
			broots[2 * b] = make_float4(r.mn[0], r.mn[1], r.mn[2], ref);
			broots[2 * b + 1] = make_float4(r.mx[0], r.mx[1], r.mx[2], r.empty ? -1.0f : r.e);
		}
		UP(bvh_nodes[variant], bnodes); UP(bvh_tris[variant], btris); UP(bvh_roots[variant], broots); UP(bvh_pairs[variant], bpairs);
		if (variant == 1)
		{
			// m-box of a child as centre and half-extent, three bytes each: centre byte b -> b/127 - 1 (nearest), half-extent
			// byte b -> b/127, rounded up so that the stored box contains [lo, hi]
			auto pack = [](const float *lo, const float *hi, unsigned &centre, unsigned &half) {
				centre = half = 0;
				for (int a = 0; a < 3; ++a)
				{
					double bc = 127.0, bh = 255.0; // NaN or out of range: the widest box
					const double l = lo[a], h = hi[a];
					if (l >= -1.0 && h <= 1.0 && l <= h)
					{
						bc = std::floor((0.5 * (l + h) + 1.0) * 127.0 + 0.5);
						bc = std::min(std::max(bc, 0.0), 254.0);
						const double c = bc / 127.0 - 1.0, need = std::max(h - c, c - l);
						bh = std::ceil(need * 127.0);
						while (bh / 127.0 < need) bh += 1.0;
						bh = std::min(std::max(bh, 0.0), 255.0);
					}
					centre |= (unsigned)bc << (8 * a);
					half |= (unsigned)bh << (8 * a);
				}
			};
			std::vector<uint4> mb(bvh.nodes.size());
			for (size_t n = 0; n < bvh.nodes.size(); ++n)
			{
				const float *m = &bvh.mbox[12 * n];
				pack(m, m + 3, mb[n].x, mb[n].y);
				pack(m + 6, m + 9, mb[n].z, mb[n].w);
			}
			UP(bvh_mbox, mb);
		}
		ctx->bvh_ptr[variant][0] = (const float4 *)ctx->bvh_nodes[variant].p;
		ctx->bvh_ptr[variant][1] = (const float4 *)ctx->bvh_tris[variant].p;
		ctx->bvh_ptr[variant][2] = (const float4 *)ctx->bvh_roots[variant].p;
		ctx->bvh_depth[variant] = bvh.depth;
	}
	ctx->have_bvh = Tg > 0 || sphere_tree;
	ctx->have_sphere_tree = sphere_tree;
#undef UP
	SceneDev &sc = ctx->sc;
	sc.bvh_nodes = sc.bvh_tris = sc.bvh_roots = nullptr; // chosen per render (make_params)
	sc.bvh_mbox = nullptr;
	sc.bvh_pairs = nullptr;
	sc.bvh_rho = 0.0f;
	sc.sph_tree = 0;
	sc.sph_geom = (const float4 *)ctx->sph_geom.p; sc.sph_radius = (const float *)ctx->sph_radius.p;
	sc.sph_mat_a = (const float4 *)ctx->sph_mat_a.p; sc.sph_mat_b = (const float2 *)ctx->sph_mat_b.p;
	sc.S = S;
	ctx->scene_generation++;
	sc.tri_geom = (const float4 *)ctx->tri_geom.p; sc.tri_shade = (const float4 *)ctx->tri_shade.p;
	sc.mat_table = (const float4 *)ctx->mat_table.p; sc.tri_orig = (const int *)ctx->tri_orig.p;
	sc.T = Tg;
	sc.box_min = (const float4 *)ctx->box_min.p; sc.box_max = (const float4 *)ctx->box_max.p;
	sc.box_range = (const int2 *)ctx->box_range.p;
	sc.B = B;
	ctx->have_scene = true;
	return WBRT_OK;
}

int wbrt_upload_sky(wbrt_context *ctx, const wbrt_vec3 *texels, int32_t width, int32_t height)
{
	int rc = bind(ctx);
	if (rc) return rc;
	if (!texels || width <= 0 || height <= 0) return fail(WBRT_ERR_INVALID, "bad sky image");
	rc = wait_renders(ctx);
	if (rc) return rc;
	ctx->have_sky = false;
	size_t n = (size_t)width * height, guard = (size_t)width + 1;
	std::vector<float> buf((n + guard) * 3, 0.0f);
	std::memcpy(buf.data(), texels, n * sizeof(wbrt_vec3));
	rc = ctx->sky.upload(buf.data(), buf.size() * sizeof(float));
	if (rc) return rc;
	ctx->sc.sky = (const float *)ctx->sky.p;
	ctx->sc.skyw = width; ctx->sc.skyh = height;
	ctx->have_sky = true;
	return WBRT_OK;
}

int wbrt_upload_seeds(wbrt_context *ctx, const uint32_t *seeds, size_t n)
{
	int rc = bind(ctx);
	if (rc) return rc;
	if (!seeds || !n) return fail(WBRT_ERR_INVALID, "empty seed buffer");
	rc = wait_renders(ctx);
	if (rc) return rc;
	ctx->n_seeds = 0;
	rc = ctx->seeds.upload(seeds, n * sizeof(uint32_t));
	if (rc) return rc;
	ctx->n_seeds = n;
	return WBRT_OK;
}

int wbrt_make_seeds_device(wbrt_context *ctx, uint64_t base_seed, uint32_t shard, size_t n, void *stream)
{
	int rc = bind(ctx);
	if (rc) return rc;
	if (!n || n > 0xFFFFFFFFull) return fail(WBRT_ERR_INVALID, "bad seed count");
	if (n * sizeof(uint32_t) > ctx->seeds.bytes || !ctx->seeds.p)
	{
		// growing the buffer frees the old one: no launch may still read it
		rc = wait_renders(ctx);
		if (rc) return rc;
		ctx->n_seeds = 0;
		rc = ctx->seeds.reserve(n * sizeof(uint32_t));
		if (rc) return rc;
	}
	// a launch of this context on ANOTHER stream may still read the buffer, or an earlier generation still write it
	rc = await_inputs(ctx, (hipStream_t)stream);
	if (rc) return rc;
	HIP_TRY(wbrt_launch_make_seeds(wbrt_seed_stream_state_(base_seed, shard), (uint32_t *)ctx->seeds.p, n, (hipStream_t)stream));
	ctx->n_seeds = n;
	return seeds_generated(ctx, (hipStream_t)stream);
}

int wbrt_render_shards(wbrt_context *ctx, const wbrt_camera *camera, int32_t num_shards, const int32_t *shard_spp, uint64_t base_seed,
                       int32_t rank, int32_t world, void *device_image, void *stream)
{
	int rc = bind(ctx);
	if (rc) return rc;
	if (!camera || !device_image) return fail(WBRT_ERR_INVALID, "null argument");
	if (num_shards <= 0 || world <= 0 || rank < 0 || rank >= world) return fail(WBRT_ERR_INVALID, "bad shard split");
	if (camera->width <= 0 || camera->height <= 0 || camera->blockSize.x < 1.0f || camera->blockSize.y < 1.0f)
		return fail(WBRT_ERR_INVALID, "bad camera");
	const size_t bytes = (size_t)camera->width * camera->height * 3 * sizeof(float);
	const size_t block_px = (size_t)(int)camera->blockSize.x * (size_t)(int)camera->blockSize.y;
	{
		// an invalid camera or a context without scene or sky leaves the caller's image as it was
		RenderParams check;
		rc = make_params(ctx, camera, true, check, false);
		if (rc) return rc;
	}
	HIP_TRY(hipMemsetAsync(device_image, 0, bytes, (hipStream_t)stream));
	// This rank's shards in ONE launch where possible: 2 to 64 shards whose sample counts are equal or drop by one once
	// (wbrt_split_spp's pattern), at least 1 sample each.  Either 2, 4, ... 64 adjacent lanes per pixel, one per shard, run the
	// pixel's streams of all shards concurrently and one lane then adds the sums to the pixel in shard order (the launch lasts as
	// long as its longest SHARD stream), or one lane per pixel runs the shards in turn: the same sums in the same order as the
	// launches below either way.
	{
		std::vector<int32_t> mine;
		for (int32_t v = rank; v < num_shards; v += world) mine.push_back(v);
		const int n = (int)mine.size();
		// (scenes with a hierarchy over their boxes keep one launch per shard: their render build has no fused form)
		bool fusable = ctx->fuse_shards && n >= 2 && n <= 64 && !(ctx->top_depth > 0 && walks_hierarchies(ctx));
		// Concurrent lanes per pixel where the frame is bound by its longest pixel stream: few pixels per lane the machine
		// holds (headline frame, 0.9 M pixels: 11.12 -> 10.94 ms per 8 shards); larger frames are bound by throughput and
		// run the shards in turn in one lane (3840x2160: 332 ms concurrently, 323 ms in turn; 1920x1080 mesh: 103.7 / 97.1 ms).
		int lanes_log2 = 0;
		if ((long long)camera->width * camera->height <= (long long)ctx->shard_lanes_max_pixels)
		{
			lanes_log2 = 1;
			while ((1 << lanes_log2) < n) ++lanes_log2;
		}
		int first_spp = 0, n_full = n;
		for (int k = 0; k < n && fusable; ++k)
		{
			const int spp = shard_spp ? shard_spp[mine[(size_t)k]] : camera->samplesPerPixel;
			if (k == 0) first_spp = spp;
			if (spp < 1) fusable = false;
			else if (spp == first_spp) { if (n_full != n) fusable = false; }
			else if (spp == first_spp - 1) { if (n_full == n) n_full = k; }
			else fusable = false;
		}
		if (fusable && first_spp - (n_full < n ? 1 : 0) < 1) fusable = false;
		if (fusable)
		{
			// all seed buffers side by side
			const size_t all = block_px * (size_t)n;
			if (all * sizeof(uint32_t) > ctx->seeds.bytes || !ctx->seeds.p)
			{
				rc = wait_renders(ctx);
				if (rc) return rc;
				ctx->n_seeds = 0;
				rc = ctx->seeds.reserve(all * sizeof(uint32_t));
				if (rc) return rc;
			}
			rc = await_inputs(ctx, (hipStream_t)stream);
			if (rc) return rc;
			for (int k = 0; k < n; ++k)
				HIP_TRY(wbrt_launch_make_seeds(wbrt_seed_stream_state_(base_seed, (uint32_t)mine[(size_t)k]),
				                               (uint32_t *)ctx->seeds.p + (size_t)k * block_px, block_px, (hipStream_t)stream));
			ctx->n_seeds = all;
			rc = seeds_generated(ctx, (hipStream_t)stream);
			if (rc) return rc;
			wbrt_camera c = *camera;
			c.samplesPerPixel = first_spp;
			RenderParams p;
			rc = make_params(ctx, &c, true, p);
			if (rc) return rc;
			p.out = (float *)device_image;
			p.out_mode = WBRT_OUT_ACCUMULATE;
			p.n_shards = n; p.n_full = n_full; p.seeds_stride = (int)block_px;
			p.shard_lanes_log2 = lanes_log2;
			return timed_launch(ctx, p, (hipStream_t)stream);
		}
	}
	for (int32_t v = rank; v < num_shards; v += world)
	{
		wbrt_camera c = *camera;
		if (shard_spp) c.samplesPerPixel = shard_spp[v];
		rc = wbrt_make_seeds_device(ctx, base_seed, (uint32_t)v, block_px, stream);
		if (rc) return rc;
		rc = wbrt_render_frame_device(ctx, &c, WBRT_OUT_ACCUMULATE, device_image, stream);
		if (rc) return rc;
	}
	return WBRT_OK;
}

int wbrt_render_block(wbrt_context *ctx, const wbrt_camera *camera, wbrt_vec3 *out_block)
{
	int rc = bind(ctx);
	if (rc) return rc;
	if (!out_block) return fail(WBRT_ERR_INVALID, "null output");
	RenderParams p;
	rc = make_params(ctx, camera, false, p);
	if (rc) return rc;
	size_t bytes = (size_t)p.bsx * p.bsy * 3 * sizeof(float);
	rc = ctx->image.reserve(bytes);
	if (rc) return rc;
	p.out = (float *)ctx->image.p;
	p.out_mode = WBRT_OUT_MEAN;
	rc = timed_launch(ctx, p, ctx->stream);
	if (rc) return rc;
	HIP_TRY(hipMemcpyAsync(out_block, p.out, bytes, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return WBRT_OK;
}

int wbrt_render_blocks(wbrt_context *ctx, const wbrt_camera *camera, wbrt_block_callback callback, void *user)
{
	int rc = bind(ctx);
	if (rc) return rc;
	if (!camera || !callback) return fail(WBRT_ERR_INVALID, "null argument");
	const int bsx = (int)camera->blockSize.x, bsy = (int)camera->blockSize.y;
	if (bsx <= 0 || bsy <= 0 || camera->width <= 0 || camera->height <= 0 || camera->width % bsx || camera->height % bsy)
		return fail(WBRT_ERR_INVALID, "blockSize must divide the image (main.cpp:222-223)");
	std::vector<wbrt_vec3> block((size_t)bsx * bsy);
	wbrt_camera c = *camera;
	// main.cpp:229-231: block rows from the top of the picture down, columns left to right
	for (int y = camera->height / bsy - 1; y >= 0; --y)
		for (int x = 0; x < camera->width / bsx; ++x)
		{
			c.blockOffset.x = (float)x; c.blockOffset.y = (float)y; c.blockOffset.z = 0.0f;
			rc = wbrt_render_block(ctx, &c, block.data());
			if (rc) return rc;
			if (callback(user, x, y, bsx, bsy, block.data()) != 0) return fail(WBRT_ERR_CANCELLED, "cancelled by the block callback");
		}
	return WBRT_OK;
}

int wbrt_render_frame_device(wbrt_context *ctx, const wbrt_camera *camera, int out_mode, void *device_image, void *stream)
{
	int rc = bind(ctx);
	if (rc) return rc;
	if (!device_image) return fail(WBRT_ERR_INVALID, "null device image");
	if (out_mode < 0 || out_mode > 2) return fail(WBRT_ERR_INVALID, "bad out_mode");
	RenderParams p;
	rc = make_params(ctx, camera, true, p);
	if (rc) return rc;
	p.out = (float *)device_image;
	p.out_mode = out_mode;
	return timed_launch(ctx, p, (hipStream_t)stream);
}

int wbrt_render_frame_device_noalloc(wbrt_context *ctx, const wbrt_camera *camera, int out_mode, void *device_image, void *stream)
{
	int rc = bind(ctx);
	if (rc) return rc;
	if (!device_image) return fail(WBRT_ERR_INVALID, "null device image");
	if (out_mode < 0 || out_mode > 2) return fail(WBRT_ERR_INVALID, "bad out_mode");
	RenderParams p;
	rc = make_params(ctx, camera, true, p);
	if (rc) return rc;
	if ((long long)p.width * p.height > (long long)WBRT_NOALLOC_MAX_PIXELS)
		return fail(WBRT_ERR_INVALID, "wbrt_render_frame_device_noalloc serves frames of at most " + std::to_string(WBRT_NOALLOC_MAX_PIXELS) +
		                                  " pixels (static schedule array); use wbrt_render_frame_device");
	WbrtSchedule s;
	if (!wbrt_scratch_schedule(ctx->device, s)) return fail(WBRT_ERR_HIP, "static schedule array not initialised for this device");
	p.out = (float *)device_image;
	p.out_mode = out_mode;
	return scheduled_launch(ctx, p, (hipStream_t)stream, &s);
}

int wbrt_render_sharded_frame(wbrt_comm *comm, wbrt_context *const *ctxs, int32_t n, const wbrt_camera *camera, int32_t num_shards,
                              const int32_t *shard_spp, uint64_t base_seed, float *host_sum)
{
	if (!ctxs || n <= 0 || !camera || !host_sum) return fail(WBRT_ERR_INVALID, "null argument");
	if (n > 1 && (!comm || wbrt_comm_size(comm) != n)) return fail(WBRT_ERR_INVALID, "communicator size does not match the contexts");
	if (camera->width <= 0 || camera->height <= 0) return fail(WBRT_ERR_INVALID, "camera width/height must be positive");
	const size_t count = (size_t)camera->width * camera->height * 3, bytes = count * sizeof(float);
	std::vector<void *> images((size_t)n), streams((size_t)n);
	// every GPU's shards are enqueued before any GPU is waited for: the GPUs render concurrently from this one thread
	for (int32_t i = 0; i < n; ++i)
	{
		wbrt_context *ctx = ctxs[i];
		int rc = bind(ctx);
		if (rc) return rc;
		rc = ctx->image.reserve(bytes);
		if (rc) return rc;
		images[(size_t)i] = ctx->image.p;
		streams[(size_t)i] = ctx->stream;
		rc = wbrt_render_shards(ctx, camera, num_shards, shard_spp, base_seed, i, n, ctx->image.p, ctx->stream);
		if (rc) return rc;
	}
	if (n > 1)
	{
		// only GPU 0's buffer is read back: a reduce to it moves half of what an all-reduce moves
		int rc = wbrt_comm_reduce_sum(comm, images.data(), count, 0, streams.data());
		if (rc) return rc;
	}
	int rc = bind(ctxs[0]);
	if (rc) return rc;
	HIP_TRY(hipMemcpyAsync(host_sum, images[0], bytes, hipMemcpyDeviceToHost, ctxs[0]->stream));
	for (int32_t i = 0; i < n; ++i)
	{
		rc = bind(ctxs[i]);
		if (rc) return rc;
		HIP_TRY(hipStreamSynchronize(ctxs[i]->stream));
	}
	return WBRT_OK;
}

int wbrt_render_frame(wbrt_context *ctx, const wbrt_camera *camera, int out_mode, float *host_image)
{
	int rc = bind(ctx);
	if (rc) return rc;
	if (!host_image || !camera) return fail(WBRT_ERR_INVALID, "null argument");
	if (camera->width <= 0 || camera->height <= 0) return fail(WBRT_ERR_INVALID, "camera width/height must be positive");
	size_t bytes = (size_t)camera->width * camera->height * 3 * sizeof(float);
	rc = ctx->image.reserve(bytes);
	if (rc) return rc;
	if (out_mode == WBRT_OUT_ACCUMULATE)
		HIP_TRY(hipMemcpyAsync(ctx->image.p, host_image, bytes, hipMemcpyHostToDevice, ctx->stream));
	rc = wbrt_render_frame_device(ctx, camera, out_mode, ctx->image.p, ctx->stream);
	if (rc) return rc;
	HIP_TRY(hipMemcpyAsync(host_image, ctx->image.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return WBRT_OK;
}

int wbrt_primary_ids_frame(wbrt_context *ctx, const wbrt_camera *camera, int32_t *host_ids)
{
	int rc = bind(ctx);
	if (rc) return rc;
	if (!host_ids) return fail(WBRT_ERR_INVALID, "null output");
	RenderParams p;
	rc = make_params(ctx, camera, true, p);
	if (rc) return rc;
	size_t bytes = (size_t)p.width * p.height * sizeof(int32_t);
	rc = ctx->ids.reserve(bytes);
	if (rc) return rc;
	p.ids = (int *)ctx->ids.p;
	rc = await_inputs(ctx, ctx->stream);
	if (rc) return rc;
	HIP_TRY(wbrt_launch_ids(p, ctx->stream));
	HIP_TRY(hipMemcpyAsync(host_ids, p.ids, bytes, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return WBRT_OK;
}

int wbrt_count_events_frame(wbrt_context *ctx, const wbrt_camera *camera, uint64_t *counters)
{
	int rc = bind(ctx);
	if (rc) return rc;
	if (!counters) return fail(WBRT_ERR_INVALID, "null output");
	RenderParams p;
	rc = make_params(ctx, camera, true, p);
	if (rc) return rc;
	size_t bytes = WBRT_NUM_COUNTERS * sizeof(unsigned long long);
	rc = ctx->counters.reserve(bytes);
	if (rc) return rc;
	rc = await_inputs(ctx, ctx->stream);
	if (rc) return rc;
	HIP_TRY(hipMemsetAsync(ctx->counters.p, 0, bytes, ctx->stream));
	p.counters = (unsigned long long *)ctx->counters.p;
	HIP_TRY(wbrt_launch_count(p, ctx->stream));
	HIP_TRY(hipMemcpyAsync(counters, p.counters, bytes, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return WBRT_OK;
}

int wbrt_tonemap_device(wbrt_context *ctx, const void *device_image, float scale,
                        int32_t width, int32_t height, void *device_rgb8, void *stream)
{
	int rc = bind(ctx);
	if (rc) return rc;
	if (!device_image || !device_rgb8 || width <= 0 || height <= 0) return fail(WBRT_ERR_INVALID, "bad tonemap argument");
	HIP_TRY(wbrt_launch_tonemap((const float *)device_image, scale, width, height, (uint8_t *)device_rgb8, (hipStream_t)stream));
	return WBRT_OK;
}

int wbrt_synchronize(wbrt_context *ctx)
{
	int rc = bind(ctx);
	if (rc) return rc;
	HIP_TRY(hipDeviceSynchronize());
	return WBRT_OK;
}

int wbrt_set_schedule_reuse(wbrt_context *ctx, int reuse)
{
	if (!ctx) return fail(WBRT_ERR_INVALID, "null context");
	ctx->reuse_costs = reuse != 0;
	ctx->recorded_spp = 0;
	return WBRT_OK;
}

int wbrt_set_mesh_mode(wbrt_context *ctx, int mode)
{
	if (!ctx) return fail(WBRT_ERR_INVALID, "null context");
	if (mode < WBRT_MESH_LOOP || mode > WBRT_MESH_BVH_CAPPED) return fail(WBRT_ERR_INVALID, "bad mesh mode");
	ctx->mesh_mode = mode;
	return WBRT_OK;
}

int wbrt_set_arith_mode(wbrt_context *ctx, int mode)
{
	if (!ctx) return fail(WBRT_ERR_INVALID, "null context");
	if (mode != WBRT_ARITH_STRICT && mode != WBRT_ARITH_TOLERANT) return fail(WBRT_ERR_INVALID, "bad arithmetic mode");
	ctx->arith_mode = mode;
	return WBRT_OK;
}

int wbrt_set_timing(wbrt_context *ctx, int enable)
{
	if (!ctx) return fail(WBRT_ERR_INVALID, "null context");
	ctx->timing = enable != 0;
	return WBRT_OK;
}

int wbrt_get_timing(wbrt_context *ctx, double *total_ms, int32_t *launches, int reset)
{
	int rc = bind(ctx);
	if (rc) return rc;
	double sum = 0.0;
	for (size_t i = 0; i < ctx->events_used; ++i)
	{
		HIP_TRY(hipEventSynchronize(ctx->events[i].second));
		float ms = 0.0f;
		HIP_TRY(hipEventElapsedTime(&ms, ctx->events[i].first, ctx->events[i].second));
		sum += ms;
	}
	if (total_ms) *total_ms = sum;
	if (launches) *launches = (int32_t)ctx->events_used;
	if (reset) ctx->events_used = 0; // the pairs stay in the pool
	return WBRT_OK;
}

int wbrt_device_info(wbrt_context *ctx, char *buf, size_t buf_len)
{
	int rc = bind(ctx);
	if (rc) return rc;
	if (!buf || !buf_len) return fail(WBRT_ERR_INVALID, "null buffer");
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
	std::snprintf(buf, buf_len,
	              "device %d: %s (%s), %d CUs, %d MHz, %.1f GiB, LDS/block %zu B, wave %d",
	              ctx->device, prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000,
	              (double)prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0), prop.sharedMemPerBlock, prop.warpSize);
	return WBRT_OK;
}

// Test and experiment hook (not in the public header): thresholds the library otherwise fixes.
//   "lpt" 0/1 longest-first scheduling; "pilot_spp" samples of a pilot launch; "box_tree_min" /
//   "sphere_tree_min" primitive counts from which the hierarchies over boxes / spheres are built (take
//   effect at the next wbrt_upload_scene); "bvh_min_idle" idle lanes before a hand-over round; "pilot_depth" bounce
//   limit of a pilot launch; "strict_loop_max" triangle count up to which the proven-margin mode uses the cooperative
//   loop; "fuse_shards" 0/1: wbrt_render_shards uses one launch per rank where it can; "wg_kernel" 0 / 4 / 8: scenes without meshes render with the workgroup kernel of that many waves that
//   re-deals pixel streams between its waves (measured slower than the per-wave kernel; kept for that measurement),
//   "wg_per_cu" its workgroups per compute unit (0: 6 of 4 waves, 3 of 8).
// None of them changes a result.
int wbrt_tune_(wbrt_context *ctx, const char *key, int value)
{
	if (!ctx || !key) return fail(WBRT_ERR_INVALID, "null argument");
	const std::string k(key);
	if (k == "lpt") ctx->lpt = value != 0;
	else if (k == "pilot_spp") ctx->pilot_spp = value > 0 ? value : 4;
	else if (k == "lpt_min_spp") ctx->lpt_min_stream_spp = value;
	else if (k == "pilot_depth") ctx->pilot_depth = value;
	else if (k == "box_tree_min") ctx->box_tree_min = value;
	else if (k == "sphere_tree_min") ctx->sphere_tree_min = value;
	else if (k == "bvh_min_idle") ctx->bvh_steal_min_idle = value;
	else if (k == "strict_loop_max") ctx->strict_loop_max = value;
	else if (k == "fuse_shards") ctx->fuse_shards = value != 0;
	else if (k == "shard_lanes_max_pixels") ctx->shard_lanes_max_pixels = value;
	else if (k == "wg_kernel")
	{
		if (value != 0 && value != 4 && value != 8) return fail(WBRT_ERR_INVALID, "wg_kernel takes 0, 4 or 8 (waves per workgroup)");
		ctx->wg_kernel = value;
	}
	else if (k == "wg_per_cu") ctx->wg_per_cu = value > 0 && value <= 8 ? value : 0;
	else return fail(WBRT_ERR_INVALID, "unknown tuning key " + k);
	ctx->recorded_spp = 0;
	return WBRT_OK;
}

// Test hook (not in the public header): bytes currently allocated for one of the context's schedule buffers.
long long wbrt_buffer_bytes_(wbrt_context *ctx, const char *name)
{
	if (!ctx || !name) return -1;
	const std::string k(name);
	const DevBuf *b = k == "costs" ? &ctx->costs : k == "tile_costs" ? &ctx->tile_costs : k == "order" ? &ctx->order : k == "hist" ? &ctx->hist : nullptr;
	if (!b) return -1;
	return b->p ? (long long)b->bytes : 0;
}

// Test hook (not in the public header): loop iterations (ray segments) each pixel took.
int wbrt_pixel_costs_(wbrt_context *ctx, const wbrt_camera *camera, int32_t *host_costs)
{
	int rc = bind(ctx);
	if (rc) return rc;
	RenderParams p;
	rc = make_params(ctx, camera, true, p);
	if (rc) return rc;
	size_t bytes = WBRT_NUM_COUNTERS * sizeof(unsigned long long);
	rc = ctx->counters.reserve(bytes);
	if (rc) return rc;
	size_t cbytes = (size_t)p.width * p.height * sizeof(int32_t);
	rc = ctx->ids.reserve(cbytes + cbytes / 64 + sizeof(int32_t)); // mesh scenes append one value per 64 work indices
	if (rc) return rc;
	HIP_TRY(hipMemsetAsync(ctx->counters.p, 0, bytes, ctx->stream));
	p.counters = (unsigned long long *)ctx->counters.p;
	p.costs = (int *)ctx->ids.p;
	rc = await_inputs(ctx, ctx->stream);
	if (rc) return rc;
	HIP_TRY(wbrt_launch_render(p, ctx->stream));
	HIP_TRY(hipMemcpyAsync(host_costs, p.costs, cbytes, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return WBRT_OK;
}

// Test hook (not in the public header): runs the device's float primitives on arrays.
int wbrt_math_probe_(wbrt_context *ctx, int op, const float *a, const float *b, float *out, int32_t n)
{
	int rc = bind(ctx);
	if (rc) return rc;
	DevBuf da, db, dout;
	size_t bytes = (size_t)n * sizeof(float);
	rc = da.upload(a, bytes); if (rc) return rc;
	if (b) { rc = db.upload(b, bytes); if (rc) { da.release(); return rc; } }
	rc = dout.reserve(bytes);
	if (!rc)
	{
		hipError_t e = wbrt_launch_math_probe(op, (const float *)da.p, b ? (const float *)db.p : nullptr, (float *)dout.p, n, ctx->stream);
		if (e == hipSuccess) e = hipMemcpyAsync(out, dout.p, bytes, hipMemcpyDeviceToHost, ctx->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
		if (e != hipSuccess) rc = fail(WBRT_ERR_HIP, hipGetErrorName(e));
	}
	da.release(); db.release(); dout.release();
	return rc;
}

} // extern "C"
