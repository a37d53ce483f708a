/*
 * ocl_rt_amd.h — C ABI of the MI355X (gfx950, HIP) path tracer that is a drop-in
 * for the reference's OpenCL block-dispatch layer.
 *
 * The reference (WillBurland/OpenCL-Ray-Tracer) has no plugin interface: its
 * render path is one OpenCL kernel, `pixel_colour`, driven by a linear run of
 * OpenCL host calls inside main() (src/cpp/main.cpp:25-77,160-205,229-247).
 * Each entry point below replaces one group of those calls and says which.
 * All structs are the reference's PODs, byte for byte
 * (src/opencl/gpu_kernel.cl:3-69 == src/cpp/opencl_objects/cl_*.hpp).
 *
 * Conventions
 *   - every function returns 0 on success, a negative wbrt_status otherwise;
 *     wbrt_last_error() returns a human-readable message for the calling thread;
 *   - the library copies on upload: the caller may free host arrays right away
 *     (the reference mallocs and never frees: main.cpp:81,95,99,113,123-124,144,148);
 *   - a context is bound to one GPU and must be used from one thread at a time
 *     (the reference is single-threaded with one in-order queue: main.cpp:175);
 *   - images are row-major, 3 floats per pixel, row 0 = BOTTOM row of the picture
 *     (gpu_kernel.cl:626-627, main.cpp:250-259).
 */
#ifndef OCL_RT_AMD_H
#define OCL_RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ PODs */
typedef struct { float x, y, z; } wbrt_vec3;                    /* gpu_kernel.cl:3-6    cl_vec3.hpp:8-11    12 B */
typedef struct { wbrt_vec3 albedo; float fuzz; float ior; int32_t type; } wbrt_material; /* :15-21 cl_material.hpp:6-12 24 B */
typedef struct { wbrt_vec3 center; float radius; wbrt_material material; } wbrt_sphere;  /* :23-28 cl_sphere.hpp:7-12   40 B */
typedef struct { wbrt_vec3 p0, p1, p2; wbrt_material material; int32_t boundingBoxId; } wbrt_triangle; /* :30-35 cl_triangle.hpp:7-12 64 B */
typedef struct { int32_t id; wbrt_vec3 min, max; } wbrt_bounding_box;                     /* :37-42 cl_bounding_box.hpp:6-10 28 B */
typedef struct
{
	wbrt_vec3 origin, horizontal, vertical, lowerLeftCorner;
	wbrt_vec3 defocusDiscU, defocusDiscV;
	wbrt_vec3 blockSize;    /* block width/height in pixels, stored as floats (main.cpp:92) */
	wbrt_vec3 blockOffset;  /* block INDEX (x, y, 0) as floats (main.cpp:234) */
	int32_t width, height, samplesPerPixel, maxDepth;
	float focusDistance, aperture;
} wbrt_camera;                                                   /* gpu_kernel.cl:53-69  cl_camera.hpp:6-22  120 B */

/* material.type values (gpu_kernel.cl:282-330) */
enum { WBRT_LAMBERTIAN = 0, WBRT_METAL = 1, WBRT_DIELECTRIC = 2, WBRT_EMISSIVE = 3 };

typedef enum
{
	WBRT_OK = 0,
	WBRT_ERR_INVALID = -1,     /* bad argument / scene not uploaded / size mismatch */
	WBRT_ERR_NO_DEVICE = -2,   /* no usable HIP device */
	WBRT_ERR_HIP = -3,         /* a HIP runtime call failed (message has the hipError name) */
	WBRT_ERR_OOM = -4,
	WBRT_ERR_IO = -5,
	WBRT_ERR_CANCELLED = -6    /* a block callback asked to stop */
} wbrt_status;

/* what a render writes per pixel */
enum
{
	WBRT_OUT_MEAN = 0,        /* clamped per-sample sum / samplesPerPixel: the reference's output (gpu_kernel.cl:647) */
	WBRT_OUT_SUM = 1,         /* clamped per-sample sum, not divided (one samples-per-pixel shard) */
	WBRT_OUT_ACCUMULATE = 2   /* image += clamped per-sample sum (shards summed on the device) */
};

typedef struct wbrt_context wbrt_context;

const char *wbrt_last_error(void);

/* ------------------------------------------------ device path (HIP, gfx950) */

/* replaces cl::Device/Platform::getDefault, cl::Context, cl::Program::build,
 * cl::Kernel and cl::CommandQueue (main.cpp:27-29,38-51,175) */
int wbrt_create(int device_ordinal, wbrt_context **out);
int wbrt_destroy(wbrt_context *ctx);

/* replaces the sphere / triangle / bounding-box buffers, their count buffers and
 * blocking writes (main.cpp:168-173,185-190) and setArg 6-11 (main.cpp:200-205).
 * Any pointer may be NULL when its count is 0. */
int wbrt_upload_scene(wbrt_context *ctx, const wbrt_sphere *spheres, int32_t sphere_count,
                      const wbrt_triangle *triangles, int32_t triangle_count,
                      const wbrt_bounding_box *boxes, int32_t box_count);

/* replaces bufferHDRImageData + width/height buffers (main.cpp:163-165,180-182,195-197).
 * texels: width*height, row-major, already tone-mapped to [0,1] (utility.cpp:172-190).
 * The device copy is followed by width+1 zero guard texels: the kernel's texel index
 * reaches one row past the end when u or v is exactly 1 (gpu_kernel.cl:343-345). */
int wbrt_upload_sky(wbrt_context *ctx, const wbrt_vec3 *texels, int32_t width, int32_t height);

/* replaces bufferRandomSeeds (main.cpp:166,183,198).  seeds[i] is read by in-block
 * work-item i of EVERY block (gpu_kernel.cl:604-606); n must be >= block pixels. */
int wbrt_upload_seeds(wbrt_context *ctx, const uint32_t *seeds, size_t n);

/* replaces one iteration of the block loop: camera upload + setArg(5) +
 * enqueueNDRangeKernel + blocking enqueueReadBuffer (main.cpp:234-247).
 * out_block: blockSize.x*blockSize.y texels indexed by in-block id (row 0 = bottom). */
int wbrt_render_block(wbrt_context *ctx, const wbrt_camera *camera, wbrt_vec3 *out_block);

/* the reference's block loop with progressive delivery (main.cpp:229-267: after every block the picture is updated and
 * the viewer polls it): blocks are rendered in its order (block rows from the top down, columns left to right) and handed
 * to `callback` one by one (block: block_w*block_h texels, row 0 = bottom, valid during the call); a non-zero return
 * stops the loop (WBRT_ERR_CANCELLED). */
typedef int (*wbrt_block_callback)(void *user, int32_t block_x, int32_t block_y, int32_t block_w, int32_t block_h, const wbrt_vec3 *block);
int wbrt_render_blocks(wbrt_context *ctx, const wbrt_camera *camera, wbrt_block_callback callback, void *user);

/* replaces the whole block loop (main.cpp:229-260) with ONE launch over every block;
 * blockOffset in *camera is ignored, blockSize selects the tiling (and therefore the
 * seed indexing).  host_image: width*height texels, row 0 = bottom.  Blocking. */
int wbrt_render_frame(wbrt_context *ctx, const wbrt_camera *camera, int out_mode, float *host_image);

/* same, into DEVICE memory (width*height*3 floats) on `stream` (a hipStream_t, NULL =
 * the default stream); asynchronous.  For WBRT_OUT_ACCUMULATE the image must hold valid
 * data (e.g. zeros).  No counterpart in the reference (it reads every block back). */
int wbrt_render_frame_device(wbrt_context *ctx, const wbrt_camera *camera, int out_mode,
                             void *device_image, void *stream);

/* wbrt_render_frame_device for callers that cannot allocate (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same
 * arguments, same result bit for bit; what differs: the launch schedule's arrays (pilot costs, tile
 * costs, launch order, histogram) live in a file-scope device array of the library (8.4 MiB per GPU,
 * held until the process ends) instead of buffers the context grows on demand; every launch runs its
 * own pilot launch (no costs are carried over between launches) and is not seen by wbrt_set_timing.
 * Calls no hipMalloc/hipFree or other allocating routine.  Limit: width*height <= 2097152 pixels
 * (e.g. 1920x1080); larger frames return WBRT_ERR_INVALID.  Every such call on one GPU shares that
 * GPU's array: calls that overlap in time on one GPU (different streams or contexts) race. */
int wbrt_render_frame_device_noalloc(wbrt_context *ctx, const wbrt_camera *camera, int out_mode,
                                     void *device_image, void *stream);

/* Primary-hit primitive id per pixel for sample 0's first ray segment: sphere index,
 * sphere_count + triangle index, or -1 (miss).  No counterpart in the reference
 * (its HitRecord carries no id, gpu_kernel.cl:44-51); it exists for parity checking.
 * host_ids: width*height int32, row 0 = bottom. */
int wbrt_primary_ids_frame(wbrt_context *ctx, const wbrt_camera *camera, int32_t *host_ids);

/* Event counters of one frame render (same events, same order as wbo_counters in
 * oracle/pt_oracle.h); used to price the algorithmic FLOPs of a workload.
 * counters: 26 x uint64.  Slow (instrumented kernel); never timed. */
int wbrt_count_events_frame(wbrt_context *ctx, const wbrt_camera *camera, uint64_t *counters);

/* replaces WriteColour (colour.cpp:5-15) on the device: gamma 2, *255, truncation.
 * scale multiplies the texel first (1/samplesPerPixel for a summed image, else 1).
 * device_rgb8: width*height*3 bytes, R,G,B order, row 0 = TOP (PPM order). */
int wbrt_tonemap_device(wbrt_context *ctx, const void *device_image, float scale,
                        int32_t width, int32_t height, void *device_rgb8, void *stream);

int wbrt_synchronize(wbrt_context *ctx);

/* ------------------------------------------------ seed shards and multi-GPU (no counterpart in the reference,
 * which drives one device: main.cpp:27-29,175).  The reference has ONE random stream per pixel with a data-dependent
 * number of draws per sample (gpu_kernel.cl:606,624-645), so a frame is split into num_shards SEED SHARDS instead of
 * sample ranges: shard v = the reference kernel's semantics with its own seed buffer wbrt_make_seeds(base_seed, v)
 * and shard_spp[v] samples per pixel, output = the un-normalised clamped per-sample sum (gpu_kernel.cl:632-644 before
 * :647).  frame = sum over shards / total samples.  The shard count is fixed per workload, independent of the GPU
 * count, so the picture is the same on 1, 2, 4 or 8 GPUs (up to float summation order). */

/* shard v belongs to rank v % world; writes the shards of `rank` (ascending), returns their count (-1: bad argument) */
int32_t wbrt_shards_of_rank(int32_t num_shards, int32_t rank, int32_t world, int32_t *shards);
/* samples per shard for total_spp samples: the first total_spp % num_shards shards take one more */
int32_t wbrt_split_spp(int32_t total_spp, int32_t num_shards, int32_t *shard_spp);

/* wbrt_make_seeds + wbrt_upload_seeds without the host: the n seeds of (base_seed, shard) are generated on the
 * device into the context's seed buffer, on `stream` (bit-identical to wbrt_make_seeds).  Replaces main.cpp:148-152
 * + :166,183 for shard renders.  The generated seeds REPLACE whatever wbrt_upload_seeds put there (like a second
 * blocking write of main.cpp:183 would).  Ordering: every later launch of this context that reads the seeds, on any
 * stream, waits for the generation (as after the reference's blocking upload, a caller may rely on "generated, then
 * rendered"). */
int wbrt_make_seeds_device(wbrt_context *ctx, uint64_t base_seed, uint32_t shard, size_t n, void *stream);

/* Renders the shards of `rank` (of `world`) of one frame into device_image (width*height*3 floats, zeroed by the
 * call, then += every shard's clamped sum), all on `stream`, asynchronously, without host synchronisation: seeds are
 * generated on the device.  shard_spp: samples per shard (num_shards entries), or NULL = camera->samplesPerPixel each.
 * With world > 1 the caller sums device_image over the ranks afterwards (wbrt_comm_allreduce_sum, or a collective of
 * its own when every rank is a process).
 * Side effect: the context's seed buffer afterwards holds the seeds of this rank's shards, not what wbrt_upload_seeds
 * put there; upload again before a render that is to use other seeds.  Arguments are validated before device_image
 * is touched.
 * A rank's shards run as ONE kernel launch where their sample counts allow (2 to 64 shards, counts equal or dropping by
 * one once, as wbrt_split_spp deals them; scenes without a hierarchy over their boxes): on frames of up to 2^20 pixels 2, 4,
 * ... 64 adjacent lanes per pixel run the pixel's streams of the shards concurrently and one lane adds their sums to the
 * pixel in shard order; on larger frames one lane per pixel runs the shards in turn.  Either way: the same sums in the same
 * order as one accumulating launch per shard (which is what runs otherwise), bit for bit. */
int wbrt_render_shards(wbrt_context *ctx, const wbrt_camera *camera, int32_t num_shards, const int32_t *shard_spp,
                       uint64_t base_seed, int32_t rank, int32_t world, void *device_image, void *stream);

/* One process, n GPUs: a communicator over RCCL (ncclCommInitAll; xGMI between the GPUs of a node).  devices: the HIP
 * ordinals, each once.  n == 1 needs no RCCL.  WBRT_ERR_NO_DEVICE when RCCL cannot be loaded. */
typedef struct wbrt_comm wbrt_comm;
int wbrt_comm_create(const int32_t *devices, int32_t n, wbrt_comm **out);
int wbrt_comm_destroy(wbrt_comm *comm);
int32_t wbrt_comm_size(const wbrt_comm *comm);
/* In-place all-reduce(sum) of n float buffers of `count` elements, buffer i on device i of the communicator, enqueued
 * on streams[i] (NULL array = default streams): the exchange step of a sharded frame, once per frame
 * (count = width*height*3: 11.06 MB at 1280x720).  Asynchronous. */
int wbrt_comm_allreduce_sum(wbrt_comm *comm, void *const *device_buffers, size_t count, void *const *streams);
/* The same exchange when only ONE GPU needs the frame (it tone-maps and writes the picture, as the single device of
 * main.cpp:249-265 does): reduce(sum) to buffer `root`; the other buffers are left undefined.  Asynchronous. */
int wbrt_comm_reduce_sum(wbrt_comm *comm, void *const *device_buffers, size_t count, int32_t root, void *const *streams);
/* One sharded frame on n GPUs driven by this thread (replaces the block loop main.cpp:229-247 for multi-GPU hosts):
 * context i (on device i of the communicator) renders the shards v % n == i into its own device buffer, all GPUs
 * concurrently; one reduce(sum) to GPU 0; host_sum (width*height*3 floats, row 0 = bottom) receives the frame's clamped
 * per-sample SUM over all shards (divide by the total sample count for the picture).  Blocking.  comm may be NULL
 * when n == 1.  Every context must hold the scene and the sky. */
int wbrt_render_sharded_frame(wbrt_comm *comm, wbrt_context *const *ctxs, int32_t n, const wbrt_camera *camera,
                              int32_t num_shards, const int32_t *shard_spp, uint64_t base_seed, float *host_sum);

/* How the render kernel finds the nearest triangle of a bounding box whose slab test
 * passed (gpu_kernel.cl:374-391).  No counterpart in the reference, which always loops.
 *   WBRT_MESH_LOOP        every triangle of the box is tested, as the reference does;
 *   WBRT_MESH_BVH         (default) a hierarchy of boxes built on upload is walked; a subtree is
 *                         skipped only if a forward error bound of the reference's float triangle
 *                         test proves that the test accepts none of its triangles for this ray
 *                         (margin per ray and node from the distance to the node, the largest
 *                         |e0||e1| below it with the test's own |det| >= 1e-5 cutoff, and a lower
 *                         bound of |det| from the orientations below it): same result as the loop;
 *   WBRT_MESH_BVH_CAPPED  opt-in: the walk on plain surface-area trees with every margin capped at
 *                         2e-4 of the ray's distance to the mesh (about 1.6x faster on the f1_car
 *                         scene).  NOT proven equal to the loop: it can differ for a ray within about
 *                         half a degree of a triangle's plane that passes that triangle at more than
 *                         the margin, where the reference's own accept/reject decision is rounding
 *                         noise.
 * In both BVH modes scenes with 48 or more bounding boxes find the boxes a ray enters through a
 * hierarchy over the boxes (conservative above, the reference's slab test gpu_kernel.cl:534-555 at
 * the leaves: the same boxes as its loop), and scenes with 48 or more spheres also walk a hierarchy over the spheres
 * (margins proven from the reference's float sphere test: same result as its loop over all
 * spheres, gpu_kernel.cl:361-371); WBRT_MESH_LOOP loops over all spheres as well.  In WBRT_MESH_BVH
 * meshes of at most 256 triangles use the loop.
 * Primary ids and event counters always use the loops.  Takes effect at the next render. */
enum { WBRT_MESH_LOOP = 0, WBRT_MESH_BVH = 1, WBRT_MESH_BVH_CAPPED = 2 };
int wbrt_set_mesh_mode(wbrt_context *ctx, int mode);

/* Arithmetic of the render launches (wbrt_render_block / _frame / _frame_device / _shards).  No counterpart in the
 * reference, whose OpenCL build passes no options (main.cpp:45) and so leaves contraction and the precision of / and sqrt
 * to the device.
 *   WBRT_ARITH_STRICT    (default) IEEE binary32 without contraction, correctly rounded / and sqrt, the reference's
 *                        evaluation order (gpu_kernel.cl:108-257,455-555): images bit-identical to a non-contracting
 *                        host build of the reference kernel;
 *   WBRT_ARITH_TOLERANT  the same algorithm and the same random streams with fused multiply-adds, the hardware's 1-ulp
 *                        reciprocal / square root and no error-margin bookkeeping: faster; a sample's path can differ
 *                        from the reference's where a decision falls within rounding distance, so images agree within
 *                        a tolerance, not bit for bit.  Measured: README scene against the reference kernel, 320x180 x
 *                        250 spp: mean absolute error 2e-4, PSNR 59 dB (bar: 1e-3 / 40 dB); main.cpp scenes with and
 *                        without the mesh below 1e-3; scenes with dozens of small spheres, where paths bounce
 *                        chaotically, 1.4e-3 / 49 dB against the strict image (any change of arithmetic, a contracting
 *                        OpenCL build of the reference included, moves a fixed share of such paths).
 * wbrt_primary_ids_frame and wbrt_count_events_frame are always strict.  Takes effect at the next render. */
enum { WBRT_ARITH_STRICT = 0, WBRT_ARITH_TOLERANT = 1 };
int wbrt_set_arith_mode(wbrt_context *ctx, int mode);

/* Launch schedule.  Waves are launched costliest pixel streams first.  The costs come from the
 * previous render launch of the same view (camera vectors, aperture, image and block size, block
 * offset, depth, scene, mesh mode): every launch records the ray segments each pixel stream took.
 * A launch without such a predecessor is preceded by a pilot launch with 4 samples per pixel.
 * Seeds and sample count may differ between the launches (seed shards of one frame share the
 * schedule).  The schedule never changes a result.  reuse = 0 makes every launch run its own pilot
 * (default 1).  No counterpart in the reference (one NDRange per block, main.cpp:240). */
int wbrt_set_schedule_reuse(wbrt_context *ctx, int reuse);

/* kernel-only timing: when enabled every render launch is bracketed by HIP events on
 * its own stream.  wbrt_get_timing synchronises and returns the sum over launches since
 * the last reset. */
int wbrt_set_timing(wbrt_context *ctx, int enable);
int wbrt_get_timing(wbrt_context *ctx, double *total_ms, int32_t *launches, int reset);

/* device facts for the harness log (replaces PrintOpenCLInfo, utility.cpp:3-61) */
int wbrt_device_info(wbrt_context *ctx, char *buf, size_t buf_len);

/* ------------------------------------- host helpers (reference host functions) */

/* CalculateCamera (cl_camera.cpp:3-45); width/height/spp/depth are runtime values
 * here (macros in globals.hpp:8-14).  aperture is a defocus ANGLE in degrees. */
void wbrt_calculate_camera(wbrt_camera *camera, wbrt_vec3 look_from, wbrt_vec3 look_at, wbrt_vec3 v_up,
                           float v_fov, float aspect_ratio, float focus_distance, float aperture,
                           int32_t width, int32_t height, int32_t samples_per_pixel, int32_t max_depth);

/* IMAGE_HEIGHT = (int)(IMAGE_WIDTH / ASPECT_RATIO) in double (globals.hpp:8-10) */
int32_t wbrt_image_height(int32_t width, double aspect_ratio);

/* GetIdealBlockSize(width*height, target_blocks) (utility.cpp:192-236, main.cpp:65) */
wbrt_vec3 wbrt_ideal_block_size(int32_t width, int32_t height, int32_t target_blocks);

/* deterministic replacement for srand(time)/rand() seeding (main.cpp:18,148-152):
 * seeds[i] = low 32 bits of (r*0x5DEECE66D + 0xB) with r a 31-bit SplitMix64 draw */
void wbrt_make_seeds(uint64_t base_seed, uint32_t shard, uint32_t *seeds, size_t n);

/* WriteColour (colour.cpp:5-15) for a whole image: rgb8 R,G,B order, row 0 = TOP */
void wbrt_tonemap_host(const float *image, float scale, int32_t width, int32_t height, uint8_t *rgb8);

/* headless writers replacing the Win32 window + output.bmp (main.cpp:262-265) */
int wbrt_write_ppm(const char *path, const uint8_t *rgb8, int32_t width, int32_t height);
int wbrt_write_pfm(const char *path, const float *image, int32_t width, int32_t height);
/* linear float picture as an uncompressed OpenEXR scanline file (channels B,G,R float32);
 * image: width*height texels, row 0 = bottom, as the render entry points deliver it */
int wbrt_write_exr(const char *path, const float *image, int32_t width, int32_t height);
/* generateBitmapImage (bitmap_io.cpp:3-29): 24-bit bottom-up BMP from top-down RGB8 */
int wbrt_write_bmp(const char *path, const uint8_t *rgb8, int32_t width, int32_t height);

/* OBJ loader: GetNumOfVertices/GetNumOfFaces/ReadMeshData (utility.cpp:63-170) with the
 * bounding-box initialisation of main.cpp:114-115.  Call with triangles == NULL to get
 * the counts; then with arrays of those sizes. */
int wbrt_read_mesh(const char *path, wbrt_vec3 scale, wbrt_vec3 translate, wbrt_material material,
                   int32_t bounding_box_id, wbrt_triangle *triangles, int32_t *triangle_count,
                   int32_t *vertex_count, wbrt_vec3 *box_min, wbrt_vec3 *box_max);

/* Order-preserving bounding-box splitter: the T triangles of ONE mesh (the reference loads a whole OBJ into one box,
 * main.cpp:111-129) are cut into target_boxes contiguous runs of the array; run k gets boundingBoxId = first_box_id + k
 * and boxes_out[k] = its bounds widened by pad x its largest extent (maxima starting at FLT_MIN like main.cpp:114-115
 * when flt_min_quirk != 0).  The reference's loops then meet the triangles in the same order as with one box
 * (gpu_kernel.cl:374-391), and a ray tests only the runs whose box it enters.  Returns the boxes written or -1. */
int32_t wbrt_split_mesh_boxes(wbrt_triangle *triangles, int32_t triangle_count, int32_t first_box_id, int32_t target_boxes,
                              float pad, int32_t flt_min_quirk, wbrt_bounding_box *boxes_out, int32_t max_boxes);

/* sky tone map of ReadHdrImageData (utility.cpp:172-190): c' = 1 + 1/pow(-1-c, 5) */
void wbrt_sky_tonemap(const float *rgb_in, size_t n_texels, wbrt_vec3 *out);
/* Radiance .hdr reader (lib/hdrloader.cpp:29-191); *rgb is malloc'd, free with wbrt_free */
int wbrt_load_hdr(const char *path, int32_t *width, int32_t *height, float **rgb);
void wbrt_free(void *p);

/* ------------------------------------------------ scene fixtures of the harness */
/* README.md:32-38 five-sphere scene; returns the count written (<= max) */
int32_t wbrt_scene_readme(wbrt_sphere *spheres, int32_t max);
/* main.cpp:98-107 eight-sphere scene */
int32_t wbrt_scene_main_cpp(wbrt_sphere *spheres, int32_t max);
/* main.cpp:81-92 camera (fov 60, focus 2, aperture 1.2 deg) and the README-scene camera
 * (not recorded by the reference: look from origin down -z, fov 90, no defocus) */
void wbrt_camera_main_cpp(wbrt_camera *camera, int32_t width, int32_t spp, int32_t depth);
void wbrt_camera_readme(wbrt_camera *camera, int32_t width, int32_t spp, int32_t depth);
/* synthetic equirectangular sky (the reference's skybox.hdr is not in its checkout):
 * per-row blend white -> (0.5,0.7,1.0), already in [0,1] */
void wbrt_sky_gradient(wbrt_vec3 *texels, int32_t width, int32_t height);
/* divergence-stress scene: n random Lambertian/metal spheres from base_seed */
int32_t wbrt_scene_stress_spheres(wbrt_sphere *spheres, int32_t n, uint64_t base_seed);
/* procedural height-field mesh, (2*nx*nz) triangles split into boxes_x*boxes_z bounding boxes */
int32_t wbrt_scene_stress_mesh(wbrt_triangle *triangles, wbrt_bounding_box *boxes,
                               int32_t nx, int32_t nz, int32_t boxes_x, int32_t boxes_z, uint64_t base_seed);

#ifdef __cplusplus
}
#endif
#endif
