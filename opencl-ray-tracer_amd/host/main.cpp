// Headless host driver: the reference's main() (src/cpp/main.cpp:13-282) with the OpenCL
// block-dispatch layer replaced by the wbrt_* C ABI and the Win32 window + output.bmp
// replaced by PPM / PFM / BMP files.  Same flow: device setup, block size, camera, scene
// (spheres + optional OBJ mesh in one bounding box), sky, seeds, uploads, then the block
// loop in the reference's order (block rows top to bottom, columns left to right), each
// block read back, tone-mapped with WriteColour and the picture rewritten after every block.
#include "wb_compat.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static void usage()
{
	std::printf(
	    "wbrt_render [--scene readme|maincpp] [--width N] [--spp N] [--depth N] [--blocks N]\n"
	    "            [--mesh file.obj] [--sky file.hdr] [--seed N] [--out name] [--whole-frame] [--device N]\n"
	    "            [--gpus N --shards V]   the frame as V seed shards of spp/V samples split over N GPUs (RCCL all-reduce)\n"
	    "  writes <name>.ppm (8-bit, gamma 2), <name>.pfm (linear float) and <name>.bmp\n");
}

#define CHECK(call)                                                              \
	do {                                                                         \
		int rc_ = (call);                                                        \
		if (rc_ != 0) {                                                          \
			std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, wbrt_last_error()); \
			return 1;                                                            \
		}                                                                        \
	} while (0)

int main(int argc, char **argv)
{
	RenderConfig cfg;
	std::string scene = "maincpp", mesh, sky_path, out = "output";
	unsigned long long seed = 0x5EED;
	bool whole_frame = false;
	int device = 0, gpus = 1, shards = 0;
	for (int i = 1; i < argc; ++i)
	{
		std::string a = argv[i];
		auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
		if (a == "--scene") scene = next();
		else if (a == "--width") cfg.image_width = std::atoi(next());
		else if (a == "--spp") cfg.samples_per_pixel = std::atoi(next());
		else if (a == "--depth") cfg.max_depth = std::atoi(next());
		else if (a == "--blocks") cfg.target_block_num = std::atoi(next());
		else if (a == "--mesh") mesh = next();
		else if (a == "--sky") sky_path = next();
		else if (a == "--seed") seed = std::strtoull(next(), nullptr, 0);
		else if (a == "--out") out = next();
		else if (a == "--whole-frame") whole_frame = true;
		else if (a == "--device") device = std::atoi(next());
		else if (a == "--gpus") gpus = std::atoi(next());
		else if (a == "--shards") shards = std::atoi(next());
		else { usage(); return a == "--help" || a == "-h" ? 0 : 2; }
	}
	const int W = cfg.image_width, H = cfg.image_height();

	// device setup (main.cpp:25-51)
	wbrt_context *ctx = nullptr;
	CHECK(wbrt_create(device, &ctx));
	char info[512];
	CHECK(wbrt_device_info(ctx, info, sizeof info));
	std::printf("\n === Device information ===\n%s\n === Done ===\n", info);
	auto begin_setup = std::chrono::steady_clock::now();

	// block size (main.cpp:63-78)
	std::printf("Blocks to use: %d\n", cfg.target_block_num);
	cl_vec3 blockSize = GetIdealBlockSize(cfg);
	if (blockSize.x <= 0 || blockSize.y <= 0) { std::fprintf(stderr, "no block size for %dx%d\n", W, H); return 1; }

	// camera + spheres (main.cpp:80-107, or the README scene)
	cl_camera camera;
	std::vector<cl_sphere> spheres(8);
	if (scene == "readme")
	{
		wbrt_camera_readme(&camera, W, cfg.samples_per_pixel, cfg.max_depth);
		spheres.resize((size_t)wbrt_scene_readme(spheres.data(), 8));
	}
	else
	{
		CalculateCamera(&camera, CreateVec3(-1.3f, 0.2f, 0.5f), CreateVec3(0.2f, 0.0f, -1.5f), CreateVec3(0.0f, 1.0f, 0.0f),
		                60.0f, (float)cfg.aspect_ratio, 2.0f, 1.2f, cfg);
		spheres.resize((size_t)wbrt_scene_main_cpp(spheres.data(), 8));
	}
	camera.blockSize = blockSize;

	// mesh in one bounding box (main.cpp:109-129)
	std::vector<cl_triangle> triangles;
	std::vector<cl_bounding_box> boxes;
	if (!mesh.empty())
	{
		std::printf("Generating triangles from mesh...\n");
		cl_material m = CreateMaterial(CreateVec3(0.8f, 0.6f, 0.2f), 0.5f, 0.0f, 1);
		int32_t nt = 0, nv = 0;
		cl_vec3 mn, mx;
		CHECK(wbrt_read_mesh(mesh.c_str(), CreateVec3(0.15f, 0.15f, 0.15f), CreateVec3(0.0f, -0.5f, -1.0f), m, 0, nullptr, &nt, &nv, nullptr, nullptr));
		triangles.resize((size_t)nt);
		CHECK(wbrt_read_mesh(mesh.c_str(), CreateVec3(0.15f, 0.15f, 0.15f), CreateVec3(0.0f, -0.5f, -1.0f), m, 0, triangles.data(), &nt, &nv, &mn, &mx));
		boxes.push_back(CreateBoundingBox(0, mn, mx));
		std::printf("%d vertices, %d triangles\n", nv, nt);
	}

	// sky (main.cpp:131-145); a synthetic gradient stands in when no .hdr is given
	std::vector<cl_vec3> sky;
	int32_t skyw = 64, skyh = 32;
	if (!sky_path.empty())
	{
		float *rgb = nullptr;
		CHECK(wbrt_load_hdr(sky_path.c_str(), &skyw, &skyh, &rgb));
		std::printf("HDR image loaded, size: %d x %d\n", skyw, skyh);
		sky.resize((size_t)skyw * skyh);
		wbrt_sky_tonemap(rgb, sky.size(), sky.data());
		wbrt_free(rgb);
	}
	else
	{
		sky.resize((size_t)skyw * skyh);
		wbrt_sky_gradient(sky.data(), skyw, skyh);
	}

	// seeds (main.cpp:147-152), one per in-block work-item
	std::printf("Generating random numbers...\n");
	const size_t block_px = (size_t)blockSize.x * (size_t)blockSize.y;
	std::vector<uint32_t> seeds(block_px);
	wbrt_make_seeds(seed, 0, seeds.data(), seeds.size());

	// Multi-GPU / sharded frame (no counterpart in the reference): the frame is `shards` seed shards of
	// samples_per_pixel/shards samples each, GPU g renders the shards v % gpus == g, one all-reduce sums them.
	if (shards > 0 || gpus > 1)
	{
		if (shards <= 0) shards = 8;
		if (gpus <= 0) { std::fprintf(stderr, "--gpus must be positive\n"); return 2; }
		std::vector<wbrt_context *> ctxs((size_t)gpus, nullptr);
		std::vector<int32_t> devices((size_t)gpus);
		ctxs[0] = ctx;
		devices[0] = device;
		for (int g = 1; g < gpus; ++g)
		{
			devices[(size_t)g] = device + g;
			CHECK(wbrt_create(device + g, &ctxs[(size_t)g]));
		}
		std::printf("Writing to GPU memory (%d GPUs)...\n", gpus);
		for (int g = 0; g < gpus; ++g)
		{
			CHECK(wbrt_upload_scene(ctxs[(size_t)g], spheres.data(), (int32_t)spheres.size(), triangles.data(), (int32_t)triangles.size(),
			                        boxes.data(), (int32_t)boxes.size()));
			CHECK(wbrt_upload_sky(ctxs[(size_t)g], sky.data(), skyw, skyh));
		}
		wbrt_comm *comm = nullptr;
		CHECK(wbrt_comm_create(devices.data(), gpus, &comm));
		std::vector<int32_t> shard_spp((size_t)shards);
		wbrt_split_spp(cfg.samples_per_pixel, shards, shard_spp.data());
		std::vector<float> sum((size_t)W * H * 3, 0.0f), image((size_t)W * H * 3);
		std::vector<uint8_t> rgb8((size_t)W * H * 3, 0);
		std::printf("Running kernel: %d seed shards of %d..%d spp over %d GPUs...\n", shards, shard_spp.back(), shard_spp.front(), gpus);
		CHECK(wbrt_render_sharded_frame(comm, ctxs.data(), gpus, &camera, shards, shard_spp.data(), seed, sum.data())); // warm-up: schedules, RCCL channels
		auto t0 = std::chrono::steady_clock::now();
		CHECK(wbrt_render_sharded_frame(comm, ctxs.data(), gpus, &camera, shards, shard_spp.data(), seed, sum.data()));
		const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		std::printf(" === Done in %f s (%.1f Mpaths/s on %d GPUs incl. all-reduce and read-back) ===\n", secs,
		            (double)W * H * cfg.samples_per_pixel / secs / 1e6, gpus);
		const float total = (float)cfg.samples_per_pixel;
		for (size_t i = 0; i < sum.size(); ++i) image[i] = sum[i] / total; // gpu_kernel.cl:647 over all shards
		wbrt_tonemap_host(sum.data(), 1.0f / total, W, H, rgb8.data());
		CHECK(wbrt_write_ppm((out + ".ppm").c_str(), rgb8.data(), W, H));
		CHECK(wbrt_write_pfm((out + ".pfm").c_str(), image.data(), W, H));
		CHECK(wbrt_write_exr((out + ".exr").c_str(), image.data(), W, H));
		CHECK(wbrt_write_bmp((out + ".bmp").c_str(), rgb8.data(), W, H));
		wbrt_comm_destroy(comm);
		for (int g = 0; g < gpus; ++g) wbrt_destroy(ctxs[(size_t)g]);
		return 0;
	}

	// uploads (main.cpp:160-205)
	std::printf("Writing to GPU memory...\n");
	CHECK(wbrt_upload_scene(ctx, spheres.data(), (int32_t)spheres.size(), triangles.data(), (int32_t)triangles.size(),
	                        boxes.data(), (int32_t)boxes.size()));
	CHECK(wbrt_upload_sky(ctx, sky.data(), skyw, skyh));
	CHECK(wbrt_upload_seeds(ctx, seeds.data(), seeds.size()));
	auto end_setup = std::chrono::steady_clock::now();
	std::printf(" === Done in %f s ===\n", std::chrono::duration<float>(end_setup - begin_setup).count());

	// render (main.cpp:217-277)
	std::printf("Running kernel...\n");
	const int blockNumX = W / (int)blockSize.x, blockNumY = H / (int)blockSize.y;
	std::printf("Blocks to render: %d (%d x %d)\nBlock size: %d x %d\n", blockNumX * blockNumY, blockNumX, blockNumY,
	            (int)blockSize.x, (int)blockSize.y);
	std::vector<float> image((size_t)W * H * 3, 0.0f);
	std::vector<uint8_t> rgb8((size_t)W * H * 3, 0);
	std::vector<cl_vec3> block(block_px);
	auto begin_render = std::chrono::steady_clock::now();
	if (whole_frame)
	{
		CHECK(wbrt_render_frame(ctx, &camera, WBRT_OUT_MEAN, image.data()));
		wbrt_tonemap_host(image.data(), 1.0f, W, H, rgb8.data());
	}
	else
	{
		int blockNum = 0;
		for (int y = blockNumY - 1; y >= 0; y--)
			for (int x = 0; x < blockNumX; x++)
			{
				camera.blockOffset = CreateVec3((float)x, (float)y, 0);
				CHECK(wbrt_render_block(ctx, &camera, block.data()));
				for (int i = 0; i < (int)blockSize.y; i++)
					std::memcpy(&image[3 * ((size_t)(i + (int)blockSize.y * y) * W + (size_t)((int)blockSize.x * x))],
					            &block[(size_t)i * (int)blockSize.x], sizeof(cl_vec3) * (size_t)blockSize.x);
				wbrt_tonemap_host(image.data(), 1.0f, W, H, rgb8.data());
				CHECK(wbrt_write_ppm((out + ".ppm").c_str(), rgb8.data(), W, H)); // picture rewritten per block (main.cpp:262-265)
				std::printf("Rendered block %d / %d (%d%%)\r", ++blockNum, blockNumX * blockNumY,
				            (int)((float)blockNum / (float)(blockNumX * blockNumY) * 100.0f));
				std::fflush(stdout);
			}
	}
	CHECK(wbrt_synchronize(ctx));
	auto end_render = std::chrono::steady_clock::now();
	const double secs = std::chrono::duration<double>(end_render - begin_render).count();
	std::printf("\n === Done in %f s (%.1f Mpaths/s incl. read-back and file writes) ===\n\n", secs,
	            (double)W * H * cfg.samples_per_pixel / secs / 1e6);
	CHECK(wbrt_write_ppm((out + ".ppm").c_str(), rgb8.data(), W, H));
	CHECK(wbrt_write_pfm((out + ".pfm").c_str(), image.data(), W, H));
	CHECK(wbrt_write_exr((out + ".exr").c_str(), (const float *)image.data(), W, H));
	CHECK(wbrt_write_bmp((out + ".bmp").c_str(), rgb8.data(), W, H));
	wbrt_destroy(ctx);
	return 0;
}
