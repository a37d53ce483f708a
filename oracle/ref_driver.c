/*
 * TEST INFRASTRUCTURE ONLY — never linked into the product library.
 *
 * Host driver appended (same translation unit) behind the reference's kernel
 * file when oracle/Makefile builds oracle/_ref/libwbref.so.  It plays the role
 * the OpenCL runtime plays for the reference: it sets the work-item id and
 * calls the kernel entry point once per work-item (the NDRange of
 * main.cpp:240), optionally on several host threads.
 *
 * It also exposes thin pointer-argument wrappers around individual reference
 * functions so that per-function known-answer vectors can be generated.
 * Everything here calls the reference's own functions; nothing is restated.
 */
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

__thread int wbref_gid = 0;

/* ---- struct sizes, so the Python side can check its ctypes mirrors ---- */
void wbref_sizes(int *out)
{
	out[0] = (int)sizeof(Vec3);
	out[1] = (int)sizeof(Material);
	out[2] = (int)sizeof(Sphere);
	out[3] = (int)sizeof(Triangle);
	out[4] = (int)sizeof(BoundingBox);
	out[5] = (int)sizeof(Camera);
	out[6] = (int)sizeof(Ray);
	out[7] = (int)sizeof(HitRecord);
}

/* ---- NDRange emulation ---- */
typedef struct
{
	Vec3 *rgb;
	Vec3 *sky;
	int skyw, skyh;
	unsigned int *seeds;
	Camera *cam;
	Sphere *spheres;
	int S;
	Triangle *tris;
	int T;
	BoundingBox *bbs;
	int B;
	int first, count;
	atomic_int next;
} wbref_job;

#define WBREF_CHUNK 64

static void *wbref_worker(void *arg)
{
	wbref_job *job = (wbref_job *)arg;
	for (;;)
	{
		int begin = atomic_fetch_add(&job->next, WBREF_CHUNK);
		if (begin >= job->count) break;
		int end = begin + WBREF_CHUNK;
		if (end > job->count) end = job->count;
		for (int k = begin; k < end; k++)
		{
			wbref_gid = job->first + k;
			pixel_colour(job->rgb, job->sky, &job->skyw, &job->skyh, job->seeds, job->cam,
			             job->spheres, &job->S, job->tris, &job->T, job->bbs, &job->B);
		}
	}
	return NULL;
}

/* Runs work-items [first, first+count) of one block launch.  rgb and seeds are
 * indexed by the in-block work-item id exactly as the kernel does. */
int wbref_run_items(void *rgb, void *sky, int skyw, int skyh, unsigned int *seeds, void *cam,
                    void *spheres, int S, void *tris, int T, void *bbs, int B,
                    int first, int count, int nthreads)
{
	wbref_job job;
	job.rgb = (Vec3 *)rgb; job.sky = (Vec3 *)sky; job.skyw = skyw; job.skyh = skyh;
	job.seeds = seeds; job.cam = (Camera *)cam;
	job.spheres = (Sphere *)spheres; job.S = S;
	job.tris = (Triangle *)tris; job.T = T;
	job.bbs = (BoundingBox *)bbs; job.B = B;
	job.first = first; job.count = count;
	atomic_init(&job.next, 0);
	if (nthreads <= 1)
	{
		wbref_worker(&job);
		return 0;
	}
	pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
	if (!th) return -1;
	int started = 0;
	for (int i = 0; i < nthreads; i++)
	{
		if (pthread_create(&th[i], NULL, wbref_worker, &job) != 0) break;
		started++;
	}
	if (started == 0) wbref_worker(&job);
	for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
	free(th);
	return 0;
}

/* ---- per-function wrappers (pointer arguments, for known-answer vectors) ---- */
unsigned long long wbref_next_seed(unsigned long long s) { return NextSeed(s); }

float wbref_rand_float(unsigned long long *s)
{
	ulong t = *s;
	float r = RandFloatFromSeed(&t);
	*s = t;
	return r;
}

int wbref_hit_sphere(const void *sphere, const void *ray, float tmin, float tmax, void *hit_out)
{
	HitRecord h;
	memset(&h, 0, sizeof h);
	int r = HitSphere(*(const Sphere *)sphere, *(const Ray *)ray, tmin, tmax, &h) ? 1 : 0;
	memcpy(hit_out, &h, sizeof h);
	return r;
}

int wbref_hit_triangle(const void *tri, const void *ray, float tmin, float tmax, void *hit_out)
{
	HitRecord h;
	memset(&h, 0, sizeof h);
	int r = HitTriangle(*(const Triangle *)tri, *(const Ray *)ray, tmin, tmax, &h) ? 1 : 0;
	memcpy(hit_out, &h, sizeof h);
	return r;
}

int wbref_hit_bbox(const void *bb, const void *ray)
{
	const BoundingBox *b = (const BoundingBox *)bb;
	return HitBoundingBox(b->min, b->max, *(const Ray *)ray) ? 1 : 0;
}

int wbref_hit_anything(void *hit_out, const void *ray, float tmin, void *spheres, int S,
                       void *tris, int T, void *bbs, int B)
{
	HitRecord h;
	memset(&h, 0, sizeof h);
	int r = HitAnything(&h, *(const Ray *)ray, tmin, INFINITY, (Sphere *)spheres, S,
	                    (Triangle *)tris, T, (BoundingBox *)bbs, B) ? 1 : 0;
	memcpy(hit_out, &h, sizeof h);
	return r;
}

void wbref_get_ray(const void *cam, float u, float v, unsigned long long *s, void *ray_out)
{
	ulong t = *s;
	Ray r = GetRay(*(const Camera *)cam, u, v, &t);
	*s = t;
	memcpy(ray_out, &r, sizeof r);
}

/* kind: 0 Lambertian, 1 metal, 2 dielectric.  Returns the scatter function's result. */
int wbref_scatter(int kind, const void *ray_in, const void *hit, unsigned long long *s,
                  void *atten_out, void *ray_out)
{
	ulong t = *s;
	HitRecord h = *(const HitRecord *)hit;
	Ray in = *(const Ray *)ray_in;
	Ray out;
	Vec3 att;
	memset(&out, 0, sizeof out);
	memset(&att, 0, sizeof att);
	bool ok = false;
	if (kind == 0) ok = LambertianScatter(in, &h, &att, &out, &t);
	else if (kind == 1) ok = MetalScatter(in, &h, &att, &out, &t);
	else if (kind == 2) ok = TransparentScatter(in, &h, &att, &out, &t);
	*s = t;
	memcpy(atten_out, &att, sizeof att);
	memcpy(ray_out, &out, sizeof out);
	return ok ? 1 : 0;
}

void wbref_ray_colour(const void *ray, int max_depth, void *sky, int skyw, int skyh,
                      void *spheres, int S, void *tris, int T, void *bbs, int B,
                      unsigned long long *s, void *rgb_out)
{
	ulong t = *s;
	Vec3 c = RayColour(*(const Ray *)ray, max_depth, (Vec3 *)sky, skyw, skyh, (Sphere *)spheres, S,
	                   (Triangle *)tris, T, (BoundingBox *)bbs, B, &t);
	*s = t;
	memcpy(rgb_out, &c, sizeof c);
}

void wbref_to_uv(const void *dir, float *uv_out)
{
	float *uv = Vec3ToUV(*(const Vec3 *)dir);
	uv_out[0] = uv[0];
	uv_out[1] = uv[1];
}
