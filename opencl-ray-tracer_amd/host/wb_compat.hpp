// C++ surface with the reference's own names (src/cpp/opencl_objects/cl_*.hpp, utility.hpp,
// colour.hpp) over the C ABI in include/ocl_rt_amd.h, so a host written against the
// reference's headers compiles against this library with its OpenCL calls swapped for
// wbrt_* calls.  Image size, samples and depth are runtime values here (macros in the
// reference's globals.hpp:8-14); they live in RenderConfig.
#ifndef WB_COMPAT_HPP
#define WB_COMPAT_HPP

#include "../../include/ocl_rt_amd.h"

typedef wbrt_vec3 cl_vec3;                 // cl_vec3.hpp:8-11
typedef wbrt_material cl_material;         // cl_material.hpp:6-12
typedef wbrt_sphere cl_sphere;             // cl_sphere.hpp:7-12
typedef wbrt_triangle cl_triangle;         // cl_triangle.hpp:7-12
typedef wbrt_bounding_box cl_bounding_box; // cl_bounding_box.hpp:6-10
typedef wbrt_camera cl_camera;             // cl_camera.hpp:6-22

struct RenderConfig // globals.hpp:8-14 as runtime values
{
	double aspect_ratio = 1.7777777;
	int image_width = 1280;
	int target_block_num = 4;
	int samples_per_pixel = 5;
	int max_depth = 32;
	int image_height() const { return wbrt_image_height(image_width, aspect_ratio); }
};

inline cl_vec3 CreateVec3(float x, float y, float z) { cl_vec3 v = { x, y, z }; return v; }            // cl_vec3.cpp:3-10
inline cl_vec3 operator+(cl_vec3 a, cl_vec3 b) { return CreateVec3(a.x + b.x, a.y + b.y, a.z + b.z); } // cl_vec3.cpp:37-43
inline cl_vec3 operator-(cl_vec3 a, cl_vec3 b) { return CreateVec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline cl_vec3 operator*(cl_vec3 a, float b) { return CreateVec3(a.x * b, a.y * b, a.z * b); }
inline cl_vec3 operator/(cl_vec3 a, float b) { return CreateVec3(a.x / b, a.y / b, a.z / b); }

inline cl_material CreateMaterial(cl_vec3 albedo, float fuzz, float ior, int type) // cl_material.cpp
{
	cl_material m; m.albedo = albedo; m.fuzz = fuzz; m.ior = ior; m.type = type; return m;
}
inline cl_sphere CreateSphere(cl_vec3 center, float radius, cl_material material) // cl_sphere.cpp
{
	cl_sphere s; s.center = center; s.radius = radius; s.material = material; return s;
}
inline cl_triangle CreateTriangle(cl_vec3 p0, cl_vec3 p1, cl_vec3 p2, cl_material material, int boundingBoxId) // cl_triangle.cpp
{
	cl_triangle t; t.p0 = p0; t.p1 = p1; t.p2 = p2; t.material = material; t.boundingBoxId = boundingBoxId; return t;
}
inline cl_bounding_box CreateBoundingBox(int id, cl_vec3 min, cl_vec3 max) // cl_bounding_box.cpp
{
	cl_bounding_box b; b.id = id; b.min = min; b.max = max; return b;
}

// cl_camera.cpp:3-45
inline void CalculateCamera(cl_camera *camera, cl_vec3 lookFrom, cl_vec3 lookAt, cl_vec3 vUp, float vFov,
                            float aspectRatio, float focusDistance, float aperture, const RenderConfig &cfg)
{
	wbrt_calculate_camera(camera, lookFrom, lookAt, vUp, vFov, aspectRatio, focusDistance, aperture,
	                      cfg.image_width, cfg.image_height(), cfg.samples_per_pixel, cfg.max_depth);
}

// utility.cpp:229-236
inline cl_vec3 GetIdealBlockSize(const RenderConfig &cfg)
{
	return wbrt_ideal_block_size(cfg.image_width, cfg.image_height(), cfg.target_block_num);
}

#endif
