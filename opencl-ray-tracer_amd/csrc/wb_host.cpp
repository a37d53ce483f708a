// Host-side helpers of the C ABI: restatements of the reference's host functions that
// prepare inputs for, or consume outputs of, the render path.  Pure CPU code; compiled
// with -ffp-contract=off so float results do not depend on the host's FMA support.
//
// Reference functions followed (file:line cited per function):
//   src/cpp/opencl_objects/cl_camera.cpp, src/cpp/utility.cpp, src/cpp/colour.cpp,
//   src/cpp/bitmap_io.cpp, src/cpp/lib/hdrloader.cpp, src/cpp/main.cpp (scene literals).
#include "../../include/ocl_rt_amd.h"

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_error;

inline wbrt_vec3 V(float x, float y, float z) { wbrt_vec3 r = { x, y, z }; return r; }
inline wbrt_vec3 sub(wbrt_vec3 a, wbrt_vec3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
inline wbrt_vec3 mul(wbrt_vec3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
inline wbrt_vec3 div(wbrt_vec3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
inline wbrt_vec3 cross(wbrt_vec3 a, wbrt_vec3 b)
{
	return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// cl_vec3.cpp:21-35: sqrt is the C double function applied to a float; rounding the
// double result to float equals the correctly rounded float square root.
inline float length(wbrt_vec3 a) { return (float)std::sqrt((double)(a.x * a.x + a.y * a.y + a.z * a.z)); }
inline wbrt_vec3 unit(wbrt_vec3 a) { return div(a, length(a)); }

inline wbrt_material M(float r, float g, float b, float fuzz, float ior, int type)
{
	wbrt_material m; m.albedo = V(r, g, b); m.fuzz = fuzz; m.ior = ior; m.type = type; return m;
}
inline wbrt_sphere S(float x, float y, float z, float radius, wbrt_material m)
{
	wbrt_sphere s; s.center = V(x, y, z); s.radius = radius; s.material = m; return s;
}

struct SplitMix64
{
	uint64_t s;
	explicit SplitMix64(uint64_t seed) : s(seed) {}
	uint64_t next()
	{
		uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
		return z ^ (z >> 31);
	}
	// uniform float in [0,1) with 24 random bits
	float unit() { return (float)(next() >> 40) * (1.0f / 16777216.0f); }
};

} // namespace

void wbrt_set_error_(const char *msg) { g_error = msg ? msg : ""; }

extern "C" {

const char *wbrt_last_error(void) { return g_error.c_str(); }

// cl_camera.cpp:3-45.  tan() there is the C double function applied to a float.
void wbrt_calculate_camera(wbrt_camera *camera, wbrt_vec3 look_from, wbrt_vec3 look_at, wbrt_vec3 v_up,
                           float v_fov, float aspect_ratio, float focus_distance, float aperture,
                           int32_t width, int32_t height, int32_t samples_per_pixel, int32_t max_depth)
{
	const float pi = (float)3.141592654;
	aperture *= pi / 180.0f;
	float theta = v_fov * pi / 180.0f;
	float h = (float)std::tan((double)(theta / 2));
	float viewport_h = 2.0f * h * focus_distance;
	float viewport_w = aspect_ratio * viewport_h;

	wbrt_vec3 w = unit(sub(look_from, look_at));
	wbrt_vec3 u = unit(cross(v_up, w));
	wbrt_vec3 v = cross(w, u);

	wbrt_vec3 vu = mul(u, viewport_w);
	wbrt_vec3 vv = mul(v, viewport_h);
	wbrt_vec3 llc = sub(sub(sub(look_from, div(vu, 2.0f)), div(vv, 2.0f)), mul(w, focus_distance));

	// tan() returns double there, so the product with focusDistance is formed in double (cl_camera.cpp:27)
	float defocus_radius = (float)((double)focus_distance * std::tan((double)(aperture / 2.0f)));

	camera->width = width;
	camera->height = height;
	camera->blockOffset = V(0, 0, 0);
	camera->samplesPerPixel = samples_per_pixel;
	camera->maxDepth = max_depth;
	camera->origin = look_from;
	camera->horizontal = vu;
	camera->vertical = vv;
	camera->lowerLeftCorner = llc;
	camera->defocusDiscU = mul(u, defocus_radius);
	camera->defocusDiscV = mul(v, defocus_radius);
	camera->focusDistance = focus_distance;
	camera->aperture = aperture;
	camera->blockSize = V((float)width, (float)height, 0); // one block until the caller tiles (main.cpp:92)
}

// globals.hpp:8-10
int32_t wbrt_image_height(int32_t width, double aspect_ratio) { return (int32_t)(width / aspect_ratio); }

// utility.cpp:192-236: among factor pairs (fw | width, fh | height) with fw*fh <= ceil(pixels/target),
// the largest product, then the most square pair.
wbrt_vec3 wbrt_ideal_block_size(int32_t width, int32_t height, int32_t target_blocks)
{
	if (width <= 0 || height <= 0 || target_blocks <= 0) return V(0, 0, 0);
	int max_product = (int)std::ceil((float)(width * height) / (float)target_blocks);
	int best_w = 0, best_h = 0;
	long best_p = -1;
	int best_d = 0;
	for (int fw = 1; fw <= width; ++fw)
	{
		if (width % fw) continue;
		for (int fh = 1; fh <= height; ++fh)
		{
			if (height % fh) continue;
			long p = (long)fw * fh;
			if (p > max_product) continue;
			int d = std::abs(fw - fh);
			if (p > best_p || (p == best_p && d < best_d)) { best_p = p; best_d = d; best_w = fw; best_h = fh; }
		}
	}
	return V((float)best_w, (float)best_h, 0);
}

// main.cpp:148-152 with rand() replaced by a SplitMix64 stream so renders are reproducible.
// The reference's expression masks with 0xFFFFFFFF (>> binds tighter than &), i.e. it keeps
// the low 32 bits of r*0x5DEECE66D + 0xB.
uint64_t wbrt_seed_stream_state_(uint64_t base_seed, uint32_t shard) { return base_seed ^ (0xD1B54A32D192ED03ULL * (uint64_t)(shard + 1)); }

void wbrt_make_seeds(uint64_t base_seed, uint32_t shard, uint32_t *seeds, size_t n)
{
	SplitMix64 g(wbrt_seed_stream_state_(base_seed, shard));
	for (size_t i = 0; i < n; ++i)
	{
		uint64_t r = g.next() >> 33; // 31-bit, like a POSIX rand()
		seeds[i] = (uint32_t)((r * 0x5DEECE66DULL + 0xBULL) & 0xFFFFFFFFULL);
	}
}

// Seed shards of a frame (no counterpart in the reference, which has one device): shard v of num_shards belongs to
// rank v % world; when total_spp is not a multiple of num_shards the first total_spp % num_shards shards take one
// sample more.
int32_t wbrt_shards_of_rank(int32_t num_shards, int32_t rank, int32_t world, int32_t *shards)
{
	if (num_shards < 0 || world <= 0 || rank < 0 || rank >= world) return -1;
	int32_t n = 0;
	for (int32_t v = rank; v < num_shards; v += world) { if (shards) shards[n] = v; ++n; }
	return n;
}

int32_t wbrt_split_spp(int32_t total_spp, int32_t num_shards, int32_t *shard_spp)
{
	if (total_spp < 0 || num_shards <= 0 || !shard_spp) return -1;
	const int32_t q = total_spp / num_shards, r = total_spp % num_shards;
	for (int32_t v = 0; v < num_shards; ++v) shard_spp[v] = q + (v < r ? 1 : 0);
	return 0;
}

// colour.cpp:5-15: sqrt is the C double function, so the product with 255.0f is formed in
// double and rounded to float once; then a truncating cast.
void wbrt_tonemap_host(const float *image, float scale, int32_t width, int32_t height, uint8_t *rgb8)
{
	for (int32_t row = 0; row < height; ++row)
	{
		const float *src = image + (size_t)(height - 1 - row) * width * 3;
		uint8_t *dst = rgb8 + (size_t)row * width * 3;
		for (int32_t i = 0; i < width * 3; ++i)
		{
			float c = src[i] * scale;
			float g = (float)(std::sqrt((double)c) * (double)255.0f);
			dst[i] = (unsigned char)g;
		}
	}
}

int wbrt_write_ppm(const char *path, const uint8_t *rgb8, int32_t width, int32_t height)
{
	FILE *f = std::fopen(path, "wb");
	if (!f) { g_error = std::string("cannot open ") + path; return WBRT_ERR_IO; }
	std::fprintf(f, "P6\n%d %d\n255\n", width, height);
	size_t n = (size_t)width * height * 3;
	bool ok = std::fwrite(rgb8, 1, n, f) == n;
	std::fclose(f);
	if (!ok) { g_error = std::string("short write ") + path; return WBRT_ERR_IO; }
	return WBRT_OK;
}

// Portable float map: "PF", little-endian (negative scale), rows bottom-to-top — which is
// the render's native row order.
int wbrt_write_pfm(const char *path, const float *image, int32_t width, int32_t height)
{
	FILE *f = std::fopen(path, "wb");
	if (!f) { g_error = std::string("cannot open ") + path; return WBRT_ERR_IO; }
	std::fprintf(f, "PF\n%d %d\n-1.0\n", width, height);
	size_t n = (size_t)width * height * 3;
	bool ok = std::fwrite(image, sizeof(float), n, f) == n;
	std::fclose(f);
	if (!ok) { g_error = std::string("short write ") + path; return WBRT_ERR_IO; }
	return WBRT_OK;
}

// OpenEXR 2 scanline file, no compression, three 32-bit float channels (B, G, R: the format
// orders channels by name), increasing-y line order with row 0 of the file = TOP of the picture
// (the render's rows are bottom-to-top, so they are written in reverse).  Linear values, no
// tone map: the headless float output next to the 8-bit PPM/BMP.
int wbrt_write_exr(const char *path, const float *image, int32_t width, int32_t height)
{
	if (!image || width <= 0 || height <= 0) { g_error = "bad image"; return WBRT_ERR_INVALID; }
	std::vector<unsigned char> head;
	auto put = [&](const void *p, size_t n) { const unsigned char *b = (const unsigned char *)p; head.insert(head.end(), b, b + n); };
	auto put_str = [&](const char *z) { put(z, std::strlen(z) + 1); };
	auto put_i32 = [&](int32_t v) { put(&v, 4); };
	auto put_f32 = [&](float v) { put(&v, 4); };
	auto attr = [&](const char *name, const char *type, int32_t size) { put_str(name); put_str(type); put_i32(size); };
	const uint32_t magic = 20000630u, version = 2u;
	put(&magic, 4); put(&version, 4);
	attr("channels", "chlist", 3 * 18 + 1);
	for (const char *c : { "B", "G", "R" })
	{
		put_str(c);            // name
		put_i32(2);            // pixel type FLOAT
		put_i32(0);            // pLinear + 3 reserved bytes
		put_i32(1); put_i32(1); // x, y sampling
	}
	head.push_back(0);
	attr("compression", "compression", 1); head.push_back(0);
	attr("dataWindow", "box2i", 16); put_i32(0); put_i32(0); put_i32(width - 1); put_i32(height - 1);
	attr("displayWindow", "box2i", 16); put_i32(0); put_i32(0); put_i32(width - 1); put_i32(height - 1);
	attr("lineOrder", "lineOrder", 1); head.push_back(0);
	attr("pixelAspectRatio", "float", 4); put_f32(1.0f);
	attr("screenWindowCenter", "v2f", 8); put_f32(0.0f); put_f32(0.0f);
	attr("screenWindowWidth", "float", 4); put_f32(1.0f);
	head.push_back(0); // end of header
	const uint64_t line_bytes = 8 + (uint64_t)width * 12;
	const uint64_t first = head.size() + (uint64_t)height * 8;
	for (int32_t y = 0; y < height; ++y) { uint64_t off = first + (uint64_t)y * line_bytes; put(&off, 8); }
	FILE *f = std::fopen(path, "wb");
	if (!f) { g_error = std::string("cannot open ") + path; return WBRT_ERR_IO; }
	bool ok = std::fwrite(head.data(), 1, head.size(), f) == head.size();
	std::vector<float> line((size_t)width * 3);
	for (int32_t y = 0; y < height && ok; ++y)
	{
		const float *src = image + (size_t)(height - 1 - y) * width * 3;
		for (int32_t x = 0; x < width; ++x)
		{
			line[(size_t)x] = src[3 * x + 2];
			line[(size_t)width + x] = src[3 * x + 1];
			line[(size_t)2 * width + x] = src[3 * x + 0];
		}
		const int32_t hdr[2] = { y, width * 12 };
		ok = std::fwrite(hdr, 4, 2, f) == 2 && std::fwrite(line.data(), 4, line.size(), f) == line.size();
	}
	std::fclose(f);
	if (!ok) { g_error = std::string("short write ") + path; return WBRT_ERR_IO; }
	return WBRT_OK;
}

// bitmap_io.cpp:3-84: 14-byte file header + 40-byte info header (only size, width, height,
// planes = 1 and bpp = 24 are filled), rows bottom-up, B,G,R, padded to 4 bytes.
int wbrt_write_bmp(const char *path, const uint8_t *rgb8, int32_t width, int32_t height)
{
	FILE *f = std::fopen(path, "wb");
	if (!f) { g_error = std::string("cannot open ") + path; return WBRT_ERR_IO; }
	int row_bytes = width * 3;
	int pad = (4 - row_bytes % 4) % 4;
	int stride = row_bytes + pad;
	uint32_t file_size = 14 + 40 + (uint32_t)stride * (uint32_t)height;
	unsigned char fh[14] = { 0 }, ih[40] = { 0 };
	fh[0] = 'B'; fh[1] = 'M';
	for (int i = 0; i < 4; ++i) fh[2 + i] = (unsigned char)(file_size >> (8 * i));
	fh[10] = 14 + 40;
	ih[0] = 40;
	for (int i = 0; i < 4; ++i) { ih[4 + i] = (unsigned char)(width >> (8 * i)); ih[8 + i] = (unsigned char)(height >> (8 * i)); }
	ih[12] = 1;
	ih[14] = 24;
	std::fwrite(fh, 1, 14, f);
	std::fwrite(ih, 1, 40, f);
	std::vector<unsigned char> row((size_t)stride, 0);
	for (int32_t y = height - 1; y >= 0; --y) // file rows go bottom-up; rgb8 row 0 is the top
	{
		const uint8_t *src = rgb8 + (size_t)y * row_bytes;
		for (int32_t x = 0; x < width; ++x)
		{
			row[3 * x + 0] = src[3 * x + 2];
			row[3 * x + 1] = src[3 * x + 1];
			row[3 * x + 2] = src[3 * x + 0];
		}
		std::fwrite(row.data(), 1, (size_t)stride, f);
	}
	std::fclose(f);
	return WBRT_OK;
}

// utility.cpp:63-170.  Two passes like the reference: count "v"/"f" lines, then read.
// Vertex = file value * scale + translate, in float.  Bounding box starts at
// (+FLT_MAX) / (+FLT_MIN): main.cpp:114-115 uses numeric_limits<float>::min() for the
// maximum, so an axis whose coordinates are all negative keeps max = 1.17549435e-38.
// Face tokens: the text after "f " is split on ' ' and '/'; with 9 tokens the vertex
// indices are tokens 0,3,6, otherwise tokens 1,5,9 (utility.cpp:147-157).
int wbrt_read_mesh(const char *path, wbrt_vec3 scale, wbrt_vec3 translate, wbrt_material material,
                   int32_t bounding_box_id, wbrt_triangle *triangles, int32_t *triangle_count,
                   int32_t *vertex_count, wbrt_vec3 *box_min, wbrt_vec3 *box_max)
{
	FILE *f = std::fopen(path, "rb");
	if (!f) { g_error = std::string("cannot open ") + path; return WBRT_ERR_IO; }
	std::vector<std::string> lines;
	{
		std::string cur;
		int ch;
		while ((ch = std::fgetc(f)) != EOF)
		{
			if (ch == '\n') { lines.push_back(cur); cur.clear(); }
			else cur.push_back((char)ch);
		}
		if (!cur.empty()) lines.push_back(cur);
	}
	std::fclose(f);

	auto first_token = [](const std::string &l) {
		size_t a = l.find_first_not_of(" \t\r\v\f");
		if (a == std::string::npos) return std::string();
		size_t b = l.find_first_of(" \t\r\v\f", a);
		return l.substr(a, b == std::string::npos ? std::string::npos : b - a);
	};

	int32_t nv = 0, nf = 0;
	for (const std::string &l : lines)
	{
		std::string t = first_token(l);
		if (t == "v") ++nv;
		else if (t == "f") ++nf;
	}
	if (vertex_count) *vertex_count = nv;
	if (!triangles)
	{
		if (triangle_count) *triangle_count = nf;
		return WBRT_OK;
	}
	if (!triangle_count || *triangle_count < nf) { g_error = "triangle array too small"; return WBRT_ERR_INVALID; }

	std::vector<wbrt_vec3> verts((size_t)nv);
	wbrt_vec3 mn = V(FLT_MAX, FLT_MAX, FLT_MAX);
	wbrt_vec3 mx = V(FLT_MIN, FLT_MIN, FLT_MIN);
	int32_t cv = 0, cf = 0;
	for (const std::string &l : lines)
	{
		std::string t = first_token(l);
		if (t == "v")
		{
			size_t pos = l.find('v') + 1;
			const char *p = l.c_str() + pos;
			char *end = nullptr;
			float c[3] = { 0, 0, 0 };
			for (int k = 0; k < 3; ++k) { c[k] = std::strtof(p, &end); if (end == p) break; p = end; }
			wbrt_vec3 w = V(c[0] * scale.x + translate.x, c[1] * scale.y + translate.y, c[2] * scale.z + translate.z);
			verts[(size_t)cv] = w;
			if (w.x < mn.x) mn.x = w.x;
			if (w.y < mn.y) mn.y = w.y;
			if (w.z < mn.z) mn.z = w.z;
			if (w.x > mx.x) mx.x = w.x;
			if (w.y > mx.y) mx.y = w.y;
			if (w.z > mx.z) mx.z = w.z;
			++cv;
		}
		else if (t == "f")
		{
			std::string body = l.size() > 2 ? l.substr(2) : std::string();
			std::vector<std::string> tok;
			std::string cur;
			for (char ch : body)
			{
				if (ch == ' ' || ch == '/') { tok.push_back(cur); cur.clear(); }
				else cur.push_back(ch);
			}
			if (!cur.empty()) tok.push_back(cur); // std::getline drops a trailing empty field
			size_t start = 1, change = 4;
			if (tok.size() == 9) { start = 0; change = 3; }
			int idx[3];
			for (size_t k = 0; k < 3; ++k)
			{
				size_t at = start + change * k;
				if (at >= tok.size()) { g_error = "malformed face line: " + l; return WBRT_ERR_INVALID; }
				idx[k] = std::atoi(tok[at].c_str());
				if (idx[k] < 1 || idx[k] > nv) { g_error = "face index out of range: " + l; return WBRT_ERR_INVALID; }
			}
			wbrt_triangle tr;
			tr.p0 = verts[(size_t)idx[0] - 1];
			tr.p1 = verts[(size_t)idx[1] - 1];
			tr.p2 = verts[(size_t)idx[2] - 1];
			tr.material = material;
			tr.boundingBoxId = bounding_box_id;
			triangles[cf++] = tr;
		}
	}
	*triangle_count = cf;
	if (box_min) *box_min = mn;
	if (box_max) *box_max = mx;
	return WBRT_OK;
}

// Splits one mesh into many bounding boxes (the reference puts a whole OBJ into ONE box, main.cpp:111-129, and then tests
// every triangle of that box for every ray that enters it, gpu_kernel.cl:374-391; its kernel accepts any number of boxes).
// Order-preserving: box k takes a CONTIGUOUS run of the triangle array, so the order in which the reference's loops meet
// the triangles (boxes ascending, triangles ascending inside a box) is the array order, exactly as with one box, and ties
// go to the same triangle (first one tested wins, gpu_kernel.cl:520).  boundingBoxId becomes first_box_id + k.  Boxes are
// the runs' bounds widened by `pad` times their largest extent (a hit the float test accepts a rounding error outside its
// triangle must still find its box entered); with flt_min_quirk the maxima start at FLT_MIN like main.cpp:114-115.
// Returns the number of boxes written, or -1.
int32_t wbrt_split_mesh_boxes(wbrt_triangle *triangles, int32_t triangle_count, int32_t first_box_id, int32_t target_boxes,
                              float pad, int32_t flt_min_quirk, wbrt_bounding_box *boxes_out, int32_t max_boxes)
{
	if (!triangles || triangle_count <= 0 || target_boxes <= 0 || !boxes_out || !(pad >= 0.0f)) return -1;
	const int32_t n = std::min(target_boxes, triangle_count);
	if (n > max_boxes) return -1;
	for (int32_t k = 0; k < n; ++k)
	{
		const int32_t a = (int32_t)((int64_t)triangle_count * k / n), b = (int32_t)((int64_t)triangle_count * (k + 1) / n);
		wbrt_vec3 mn = V(FLT_MAX, FLT_MAX, FLT_MAX);
		wbrt_vec3 mx = flt_min_quirk ? V(FLT_MIN, FLT_MIN, FLT_MIN) : V(-FLT_MAX, -FLT_MAX, -FLT_MAX);
		for (int32_t j = a; j < b; ++j)
		{
			triangles[j].boundingBoxId = first_box_id + k;
			const wbrt_vec3 *p[3] = { &triangles[j].p0, &triangles[j].p1, &triangles[j].p2 };
			for (const wbrt_vec3 *v : p)
			{
				mn.x = std::min(mn.x, v->x); mn.y = std::min(mn.y, v->y); mn.z = std::min(mn.z, v->z);
				mx.x = std::max(mx.x, v->x); mx.y = std::max(mx.y, v->y); mx.z = std::max(mx.z, v->z);
			}
		}
		const float extent = std::max(std::max(mx.x - mn.x, mx.y - mn.y), mx.z - mn.z);
		const float m = pad * (extent > 0.0f && extent < FLT_MAX ? extent : 0.0f);
		boxes_out[k].id = first_box_id + k;
		boxes_out[k].min = V(mn.x - m, mn.y - m, mn.z - m);
		boxes_out[k].max = V(mx.x + m, mx.y + m, mx.z + m);
	}
	return n;
}

// utility.cpp:172-190: exposure exponent 2*2+1 = 5, evaluated in double, stored as float.
void wbrt_sky_tonemap(const float *rgb_in, size_t n_texels, wbrt_vec3 *out)
{
	for (size_t i = 0; i < n_texels; ++i)
	{
		float c[3];
		for (int k = 0; k < 3; ++k)
		{
			float x = rgb_in[3 * i + k];
			c[k] = (float)(1 + (1 / std::pow((double)(-1 - x), 5.0)));
		}
		out[i] = V(c[0], c[1], c[2]);
	}
}

namespace {
struct Rgbe { unsigned char c[4]; };

// hdrloader.cpp:161-191: flat RGBE with (1,1,1,n) repeat markers
bool read_flat(Rgbe *scan, int len, FILE *f)
{
	int rshift = 0;
	Rgbe *begin = scan;
	while (len > 0)
	{
		int r = std::fgetc(f), g = std::fgetc(f), b = std::fgetc(f), e = std::fgetc(f);
		if (e == EOF) return false;
		scan->c[0] = (unsigned char)r; scan->c[1] = (unsigned char)g; scan->c[2] = (unsigned char)b; scan->c[3] = (unsigned char)e;
		if (r == 1 && g == 1 && b == 1)
		{
			for (int i = e << rshift; i > 0 && len > 0; --i)
			{
				if (scan == begin) return false;
				*scan = *(scan - 1);
				++scan; --len;
			}
			rshift += 8;
		}
		else { ++scan; --len; rshift = 0; }
	}
	return true;
}

// hdrloader.cpp:118-159: adaptive run-length scanline, one component plane at a time
bool read_scanline(Rgbe *scan, int len, FILE *f)
{
	if (len < 8 || len > 0x7fff) return read_flat(scan, len, f);
	int i = std::fgetc(f);
	if (i != 2) { std::ungetc(i, f); return read_flat(scan, len, f); }
	scan[0].c[1] = (unsigned char)std::fgetc(f);
	scan[0].c[2] = (unsigned char)std::fgetc(f);
	i = std::fgetc(f);
	if (scan[0].c[1] != 2 || (scan[0].c[2] & 128))
	{
		scan[0].c[0] = 2;
		scan[0].c[3] = (unsigned char)i;
		return read_flat(scan + 1, len - 1, f);
	}
	for (int comp = 0; comp < 4; ++comp)
	{
		for (int j = 0; j < len;)
		{
			int code = std::fgetc(f);
			if (code == EOF) return false;
			if (code > 128)
			{
				code &= 127;
				int val = std::fgetc(f);
				while (code-- && j < len) scan[j++].c[comp] = (unsigned char)val;
			}
			else
			{
				while (code-- && j < len) scan[j++].c[comp] = (unsigned char)std::fgetc(f);
			}
		}
	}
	return !std::feof(f);
}
} // namespace

// hdrloader.cpp:29-116: "#?RADIANCE" magic, header up to a blank line, "-Y h +X w",
// scanlines top to bottom; component = mantissa/256 * 2^(e-128).
int wbrt_load_hdr(const char *path, int32_t *width, int32_t *height, float **rgb)
{
	FILE *f = std::fopen(path, "rb");
	if (!f) { g_error = std::string("cannot open ") + path; return WBRT_ERR_IO; }
	char magic[10];
	if (std::fread(magic, 10, 1, f) != 1 || std::memcmp(magic, "#?RADIANCE", 10) != 0)
	{
		std::fclose(f);
		g_error = "not a Radiance file";
		return WBRT_ERR_IO;
	}
	std::fseek(f, 1, SEEK_CUR);
	int c = 0, oldc = 0;
	for (;;)
	{
		oldc = c;
		c = std::fgetc(f);
		if (c == EOF) { std::fclose(f); g_error = "truncated header"; return WBRT_ERR_IO; }
		if (c == 0xa && oldc == 0xa) break;
	}
	char reso[200];
	int n = 0;
	for (;;)
	{
		c = std::fgetc(f);
		if (c == EOF || n >= 199) { std::fclose(f); g_error = "bad resolution line"; return WBRT_ERR_IO; }
		reso[n++] = (char)c;
		if (c == 0xa) break;
	}
	reso[n] = 0;
	int w = 0, h = 0;
	if (std::sscanf(reso, "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0)
	{
		std::fclose(f);
		g_error = "bad resolution line";
		return WBRT_ERR_IO;
	}
	float *cols = (float *)std::calloc((size_t)w * h * 3, sizeof(float));
	std::vector<Rgbe> scan((size_t)w);
	if (!cols) { std::fclose(f); return WBRT_ERR_OOM; }
	float *out = cols;
	for (int y = h - 1; y >= 0; --y)
	{
		if (!read_scanline(scan.data(), w, f)) break;
		for (int x = 0; x < w; ++x)
		{
			int expo = scan[(size_t)x].c[3] - 128;
			float d = std::pow(2.0f, (float)expo);
			for (int k = 0; k < 3; ++k) out[k] = (scan[(size_t)x].c[k] / 256.0f) * d;
			out += 3;
		}
	}
	std::fclose(f);
	*width = w; *height = h; *rgb = cols;
	return WBRT_OK;
}

void wbrt_free(void *p) { std::free(p); }

// ------------------------------------------------------------ scene fixtures

// README.md:32-38 (types: Lambertian 0, metal 1)
int32_t wbrt_scene_readme(wbrt_sphere *s, int32_t max)
{
	if (max < 5) return 0;
	s[0] = S(0.0f, -100.5f, -1.0f, 100.0f, M(0.0f, 0.8f, 0.7f, 0.0f, 0.0f, WBRT_LAMBERTIAN));
	s[1] = S(0.0f, 0.5f, -1.0f, 0.5f, M(1.0f, 0.5f, 0.3f, 0.0f, 0.0f, WBRT_LAMBERTIAN));
	s[2] = S(-0.9f, 0.0f, -1.0f, 0.5f, M(0.8f, 0.5f, 0.5f, 0.1f, 0.0f, WBRT_METAL));
	s[3] = S(0.9f, 0.0f, -1.0f, 0.5f, M(0.8f, 0.6f, 0.2f, 0.5f, 0.0f, WBRT_METAL));
	s[4] = S(0.0f, -0.3f, -1.0f, 0.2f, M(0.8f, 0.8f, 0.8f, 0.0f, 0.0f, WBRT_METAL));
	return 5;
}

// main.cpp:98-107
int32_t wbrt_scene_main_cpp(wbrt_sphere *s, int32_t max)
{
	if (max < 8) return 0;
	s[0] = S(0.0f, -100.5f, -1.0f, 100.0f, M(0.3f, 0.5f, 0.4f, 0.0f, 0.0f, 0));
	s[1] = S(1.6f, 0.0f, -1.3f, 0.5f, M(0.7f, 0.3f, 0.9f, 0.0f, 0.0f, 0));
	s[2] = S(-0.5f, 0.0f, -2.0f, 0.5f, M(0.8f, 0.5f, 0.5f, 0.2f, 0.0f, 1));
	s[3] = S(0.6f, 0.1f, -1.9f, 0.6f, M(0.8f, 0.8f, 0.8f, 0.0f, 0.0f, 1));
	s[4] = S(0.2f, -0.35f, -0.4f, 0.15f, M(0.8f, 0.8f, 0.8f, 0.0f, 1.5f, 2));
	s[5] = S(-0.4f, -0.4f, -0.6f, 0.1f, M(1.0f, 1.0f, 1.0f, 0.0f, 0.0f, 3));
	s[6] = S(-0.2f, -0.1f, 0.6f, 0.4f, M(0.5f, 0.5f, 0.8f, 0.0f, 0.0f, 1));
	s[7] = S(-1.5f, -0.1f, -5.0f, 0.5f, M(0.5f, 0.8f, 0.5f, 0.0f, 0.0f, 1));
	return 8;
}

// main.cpp:81-92
void wbrt_camera_main_cpp(wbrt_camera *camera, int32_t width, int32_t spp, int32_t depth)
{
	wbrt_calculate_camera(camera, V(-1.3f, 0.2f, 0.5f), V(0.2f, 0.0f, -1.5f), V(0.0f, 1.0f, 0.0f),
	                      60.0f, (float)1.7777777, 2.0f, 1.2f, width, wbrt_image_height(width, 1.7777777), spp, depth);
}

void wbrt_camera_readme(wbrt_camera *camera, int32_t width, int32_t spp, int32_t depth)
{
	wbrt_calculate_camera(camera, V(0.0f, 0.0f, 0.0f), V(0.0f, 0.0f, -1.0f), V(0.0f, 1.0f, 0.0f),
	                      90.0f, (float)1.7777777, 1.0f, 0.0f, width, wbrt_image_height(width, 1.7777777), spp, depth);
}

// Row 0 of an equirectangular map is straight up (v = 0.5 - asin(y)/pi = 0 for y = 1).
void wbrt_sky_gradient(wbrt_vec3 *texels, int32_t width, int32_t height)
{
	for (int32_t y = 0; y < height; ++y)
	{
		float t = height > 1 ? 1.0f - (float)y / (float)(height - 1) : 1.0f; // 1 at the zenith
		wbrt_vec3 c = V((1.0f - t) * 1.0f + t * 0.5f, (1.0f - t) * 1.0f + t * 0.7f, (1.0f - t) * 1.0f + t * 1.0f);
		for (int32_t x = 0; x < width; ++x) texels[(size_t)y * width + x] = c;
	}
}

int32_t wbrt_scene_stress_spheres(wbrt_sphere *s, int32_t n, uint64_t base_seed)
{
	if (n < 1) return 0;
	SplitMix64 g(base_seed);
	s[0] = S(0.0f, -1000.0f, 0.0f, 1000.0f, M(0.5f, 0.5f, 0.5f, 0.0f, 0.0f, WBRT_LAMBERTIAN));
	for (int32_t i = 1; i < n; ++i)
	{
		float r = 0.1f + 0.3f * g.unit();
		float x = -15.0f + 30.0f * g.unit();
		float z = -15.0f + 30.0f * g.unit();
		int type = g.unit() < 0.5f ? WBRT_LAMBERTIAN : WBRT_METAL;
		float ar = 0.1f + 0.9f * g.unit(), ag = 0.1f + 0.9f * g.unit(), ab = 0.1f + 0.9f * g.unit();
		float fuzz = type == WBRT_METAL ? 0.5f * g.unit() : 0.0f;
		s[i] = S(x, r, z, r, M(ar, ag, ab, fuzz, 0.0f, type));
	}
	return n;
}

// Height field over x,z in [-6,6] x [-14,-2], y = 0.6 + bumps, nx*nz cells of 2 triangles;
// cells are grouped into boxes_x*boxes_z bounding boxes, triangles listed box by box.
int32_t wbrt_scene_stress_mesh(wbrt_triangle *tris, wbrt_bounding_box *boxes,
                               int32_t nx, int32_t nz, int32_t boxes_x, int32_t boxes_z, uint64_t base_seed)
{
	if (nx < 1 || nz < 1 || boxes_x < 1 || boxes_z < 1) return 0;
	SplitMix64 g(base_seed ^ 0xA5A5A5A5ULL);
	std::vector<float> hgt((size_t)(nx + 1) * (nz + 1));
	for (int32_t k = 0; k <= nz; ++k)
		for (int32_t i = 0; i <= nx; ++i)
		{
			float fx = (float)i / (float)nx, fz = (float)k / (float)nz;
			hgt[(size_t)k * (nx + 1) + i] = 0.6f + 0.25f * std::sin(9.0f * fx) * std::cos(7.0f * fz) + 0.03f * g.unit();
		}
	auto P = [&](int32_t i, int32_t k) {
		float x = -6.0f + 12.0f * (float)i / (float)nx;
		float z = -14.0f + 12.0f * (float)k / (float)nz;
		return V(x, hgt[(size_t)k * (nx + 1) + i], z);
	};
	wbrt_material mat = M(0.7f, 0.6f, 0.5f, 0.3f, 0.0f, WBRT_METAL);
	wbrt_material mat2 = M(0.4f, 0.6f, 0.3f, 0.0f, 0.0f, WBRT_LAMBERTIAN);
	int32_t count = 0;
	for (int32_t bz = 0; bz < boxes_z; ++bz)
		for (int32_t bx = 0; bx < boxes_x; ++bx)
		{
			int32_t b = bz * boxes_x + bx;
			int32_t i0 = (int32_t)((int64_t)nx * bx / boxes_x), i1 = (int32_t)((int64_t)nx * (bx + 1) / boxes_x);
			int32_t k0 = (int32_t)((int64_t)nz * bz / boxes_z), k1 = (int32_t)((int64_t)nz * (bz + 1) / boxes_z);
			wbrt_vec3 mn = V(FLT_MAX, FLT_MAX, FLT_MAX), mx = V(-FLT_MAX, -FLT_MAX, -FLT_MAX);
			for (int32_t k = k0; k < k1; ++k)
				for (int32_t i = i0; i < i1; ++i)
				{
					wbrt_vec3 a = P(i, k), bq = P(i + 1, k), c = P(i, k + 1), d = P(i + 1, k + 1);
					wbrt_vec3 corner[4] = { a, bq, c, d };
					for (const wbrt_vec3 &p : corner)
					{
						if (p.x < mn.x) mn.x = p.x;
						if (p.y < mn.y) mn.y = p.y;
						if (p.z < mn.z) mn.z = p.z;
						if (p.x > mx.x) mx.x = p.x;
						if (p.y > mx.y) mx.y = p.y;
						if (p.z > mx.z) mx.z = p.z;
					}
					const wbrt_material &m = ((i + k) & 1) ? mat : mat2;
					wbrt_triangle t0; t0.p0 = a; t0.p1 = c; t0.p2 = bq; t0.material = m; t0.boundingBoxId = b;
					wbrt_triangle t1; t1.p0 = bq; t1.p1 = c; t1.p2 = d; t1.material = m; t1.boundingBoxId = b;
					tris[count++] = t0;
					tris[count++] = t1;
				}
			boxes[b].id = b;
			boxes[b].min = mn;
			boxes[b].max = mx;
		}
	return count;
}

} // extern "C"
