/*
 * TEST INFRASTRUCTURE ONLY — never linked into the product library.
 *
 * C wrappers around the reference's own HOST functions (src/cpp/opencl_objects/*.cpp,
 * utility.cpp, colour.cpp, bitmap_io.cpp, lib/hdrloader.cpp), compiled from where they lie
 * under /root/reference by oracle/Makefile into oracle/_ref/libwbref_host.so.  The product's
 * host helpers (wbrt_calculate_camera, wbrt_ideal_block_size, wbrt_read_mesh, wbrt_load_hdr,
 * wbrt_sky_tonemap, wbrt_tonemap_host, wbrt_write_bmp) are pinned against these.
 * Everything here calls the reference; nothing is restated.  The reference's image size,
 * samples and depth are macros (globals.hpp: 1280x720, 5 spp, depth 32).
 */
#include "utility.hpp"
#include "colour.hpp"
#include "bitmap_io.hpp"
#include "opencl_objects/cl_bounding_box.hpp"
#include "opencl_objects/cl_camera.hpp"
#include "opencl_objects/cl_sphere.hpp"

#include <cstring>
#include <limits>

unsigned char image[IMAGE_HEIGHT][IMAGE_WIDTH][BYTES_PER_PIXEL]; // the global colour.cpp writes into

extern "C" {

void wbrefh_config(int *out) { out[0] = IMAGE_WIDTH; out[1] = IMAGE_HEIGHT; out[2] = SAMPLES_PER_PIXEL; out[3] = MAX_DEPTH; out[4] = TARGET_BLOCK_NUM; }

void wbrefh_calculate_camera(void *camera_out, const float *from, const float *at, const float *up,
                             float vfov, float aspect, float focus, float aperture)
{
	cl_camera cam;
	std::memset(&cam, 0, sizeof cam);
	CalculateCamera(&cam, cl_vec3{ from[0], from[1], from[2] }, cl_vec3{ at[0], at[1], at[2] }, cl_vec3{ up[0], up[1], up[2] },
	                vfov, aspect, focus, aperture);
	std::memcpy(camera_out, &cam, sizeof cam);
}

void wbrefh_ideal_block_size(int global_size, int local_size, float *out)
{
	cl_vec3 v = GetIdealBlockSize(global_size, local_size);
	out[0] = v.x; out[1] = v.y; out[2] = v.z;
}

/* main.cpp:112-129: counts, then ReadMeshData with the box initialised as main() does */
int wbrefh_read_mesh(const char *path, const float *scale, const float *translate, const void *material,
                     void *triangles_out, int max_triangles, int *num_vertices, int *num_triangles, float *box_min, float *box_max)
{
	std::ifstream f(path);
	if (!f) return -1;
	int nv = GetNumOfVertices(f); f.clear(); f.seekg(0, std::ios::beg);
	int nt = GetNumOfFaces(f);    f.clear(); f.seekg(0, std::ios::beg);
	*num_vertices = nv; *num_triangles = nt;
	if (!triangles_out) return 0;
	if (nt > max_triangles) return -2;
	cl_vec3 mn = CreateVec3(std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max());
	cl_vec3 mx = CreateVec3(std::numeric_limits<float>::min(), std::numeric_limits<float>::min(), std::numeric_limits<float>::min());
	cl_vec3 *verts = (cl_vec3 *)malloc(sizeof(cl_vec3) * (size_t)(nv > 0 ? nv : 1));
	cl_material m;
	std::memcpy(&m, material, sizeof m);
	ReadMeshData(f, verts, (cl_triangle *)triangles_out, mn, mx, cl_vec3{ scale[0], scale[1], scale[2] },
	             cl_vec3{ translate[0], translate[1], translate[2] }, m);
	free(verts);
	box_min[0] = mn.x; box_min[1] = mn.y; box_min[2] = mn.z;
	box_max[0] = mx.x; box_max[1] = mx.y; box_max[2] = mx.z;
	return 0;
}

/* HDRLoader::load + ReadHdrImageData (main.cpp:131-145); raw = decoded floats, toned = after the sky tone map */
int wbrefh_load_hdr(const char *path, int *w, int *h, float *raw, float *toned, int max_texels)
{
	HDRLoaderResult res;
	if (!HDRLoader::load(path, res)) return -1;
	*w = res.width; *h = res.height;
	if (!raw) return 0;
	if (res.width * res.height > max_texels) return -2;
	std::memcpy(raw, res.cols, sizeof(float) * 3 * (size_t)res.width * res.height);
	ReadHdrImageData((cl_vec3 *)toned, res);
	return 0;
}

/* WriteColour (colour.cpp:5-15) into the global image, then read the three bytes back (B,G,R order) */
void wbrefh_write_colour(int row, int col, const float *rgb, unsigned char *bgr_out)
{
	WriteColour(row, col, cl_vec3{ rgb[0], rgb[1], rgb[2] });
	bgr_out[0] = image[row][col][0]; bgr_out[1] = image[row][col][1]; bgr_out[2] = image[row][col][2];
}

/* generateBitmapImage (bitmap_io.cpp:3-29) of a caller-supplied B,G,R image, rows bottom-up */
void wbrefh_write_bmp(unsigned char *bgr_image, int height, int width, const char *path)
{
	generateBitmapImage(bgr_image, height, width, (char *)path);
}

} // extern "C"
