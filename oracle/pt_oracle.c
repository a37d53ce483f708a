/*
 * TEST INFRASTRUCTURE ONLY — see pt_oracle.h.
 *
 * CPU restatement of the reference path tracer (src/opencl/gpu_kernel.cl).
 * Structure differs from the reference (the closest hit is kept as (t, id) and
 * the hit point / normal are derived once per segment), the arithmetic does not:
 * every float operation below is written in the reference's evaluation order and
 * must be compiled without FMA contraction.  Each function cites the lines it
 * follows.
 */
#define _GNU_SOURCE
#include "pt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

typedef wbo_vec3 v3;

/* ------------------------------------------------------------------ vectors */
/* gpu_kernel.cl:108-180: componentwise, dot = (x*x' + y*y') + z*z' */
static inline v3 v_add(v3 a, v3 b) { v3 r = { a.x + b.x, a.y + b.y, a.z + b.z }; return r; }
static inline v3 v_sub(v3 a, v3 b) { v3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
static inline v3 v_mul(v3 a, v3 b) { v3 r = { a.x * b.x, a.y * b.y, a.z * b.z }; return r; }
static inline v3 v_scale(v3 a, float s) { v3 r = { a.x * s, a.y * s, a.z * s }; return r; }
static inline v3 v_div(v3 a, float s) { v3 r = { a.x / s, a.y / s, a.z / s }; return r; }
static inline float v_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float v_len2(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
static inline v3 v_cross(v3 a, v3 b)
{
	v3 r = { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
	return r;
}
static inline v3 v_unit(v3 a) { return v_div(a, sqrtf(v_len2(a))); } /* :177-180 */

/* ---------------------------------------------------------------------- RNG */
/* gpu_kernel.cl:582-585 */
uint64_t wbo_next_seed(uint64_t s)
{
	return (s * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
}

/* gpu_kernel.cl:587-591: the divisor literal 0xFFFFFFFF converts to 2^32 in float */
float wbo_rand_float(uint64_t *s)
{
	*s = wbo_next_seed(*s);
	return (float)(*s >> 16) / 4294967296.0f;
}

typedef struct
{
	uint64_t seed;
	wbo_counters *c; /* may be NULL */
	int math_mode;
} lane;

static inline float draw(lane *L)
{
	if (L->c) L->c->rng_draws++;
	return wbo_rand_float(&L->seed);
}

/* ----------------------------------------------- portable transcendental math */
/* The reference calls OpenCL's atan2/asin/pow, whose last bits are implementation
 * defined.  WBO_MATH_LIBM uses the host libm (what oracle/_ref uses).
 * WBO_MATH_PORTABLE evaluates fixed double-precision sequences (+,-,*,/,sqrt only,
 * no contraction) that the GPU kernel evaluates identically, so GPU and oracle
 * agree bit for bit; both modes agree with each other except for rare last-bit
 * differences. */
static const double ATAN_K8[9] = {
	0x0.0p+0,               0x1.fd5ba9aac2f6ep-4, 0x1.f5b75f92c80ddp-3,
	0x1.6f61941e4def1p-2,   0x1.dac670561bb4fp-2, 0x1.1e00babdefeb4p-1,
	0x1.4978fa3269ee1p-1,   0x1.700a7c5784634p-1, 0x1.921fb54442d18p-1
};
#define PT_PI    0x1.921fb54442d18p+1
#define PT_PI_2  0x1.921fb54442d18p+0

/* atan2 for doubles that came from floats; result rounded to float by the caller */
static double atan2_core(double y, double x)
{
	if (x != x || y != y) return x + y;
	double ax = fabs(x), ay = fabs(y);
	double hi = ax > ay ? ax : ay;
	double lo = ax > ay ? ay : ax;
	double a = 0.0;
	if (hi > 0.0)
	{
		if (hi == INFINITY) a = (lo == INFINITY) ? 1.0 : 0.0;
		else a = lo / hi;
	}
	int k = (int)(a * 8.0 + 0.5);
	double c = (double)k * 0.125;
	double t = (a - c) / (1.0 + a * c);
	double t2 = t * t;
	double p = t2 * (1.0 / 9.0);
	p = t2 * (p - 1.0 / 7.0);
	p = t2 * (p + 1.0 / 5.0);
	p = t2 * (p - 1.0 / 3.0);
	p = t + t * p;
	double r = ATAN_K8[k] + p;
	if (ay > ax) r = PT_PI_2 - r;
	if (signbit(x)) r = PT_PI - r;
	if (signbit(y)) r = -r;
	return r;
}

float wbo_atan2f_portable(float y, float x) { return (float)atan2_core((double)y, (double)x); }

float wbo_asinf_portable(float y)
{
	double d = (double)y;
	if (!(fabs(d) <= 1.0)) return NAN;
	double c = sqrt((1.0 - d) * (1.0 + d));
	return (float)atan2_core(d, c);
}

float wbo_pow5f_portable(float x)
{
	double d = (double)x;
	double d2 = d * d;
	double d4 = d2 * d2;
	return (float)(d4 * d);
}

static inline float m_atan2(int mode, float y, float x) { return mode == WBO_MATH_LIBM ? atan2f(y, x) : wbo_atan2f_portable(y, x); }
static inline float m_asin(int mode, float y) { return mode == WBO_MATH_LIBM ? asinf(y) : wbo_asinf_portable(y); }
static inline float m_pow5(int mode, float x) { return mode == WBO_MATH_LIBM ? powf(x, 5.0f) : wbo_pow5f_portable(x); }

/* ------------------------------------------------------------- intersection */
/* gpu_kernel.cl:455-478: half-b quadratic; returns 1 and *t on acceptance */
static int sphere_root(const wbo_sphere *s, v3 o, v3 d, float tmin, float tmax, float *t, wbo_counters *c)
{
	if (c) c->sphere_tests++;
	v3 oc = v_sub(o, s->center);
	float a = v_len2(d);
	float hb = v_dot(oc, d);
	float cc = v_len2(oc) - s->radius * s->radius;
	float disc = hb * hb - a * cc;
	if (disc < 0) return 0;
	if (c) c->sphere_disc_ge0++;
	float sq = sqrtf(disc);
	float root = (-hb - sq) / a;
	if (root < tmin || tmax < root)
	{
		if (c) c->sphere_second_root++;
		root = (-hb + sq) / a;
		if (root < tmin || tmax < root) return 0;
	}
	if (c) c->sphere_accept++;
	*t = root;
	return 1;
}

/* gpu_kernel.cl:489-520: Moeller-Trumbore, two-sided, strict t bounds */
static int triangle_t(const wbo_triangle *tr, v3 o, v3 d, float tmin, float tmax, float *t, wbo_counters *c)
{
	if (c) c->tri_tests++;
	v3 e0 = v_sub(tr->p1, tr->p0);
	v3 e1 = v_sub(tr->p2, tr->p0);
	v3 h = v_cross(d, e1);
	float a = v_dot(e0, h);
	if (a > -0.00001f && a < 0.00001f) { if (c) c->tri_exit_parallel++; return 0; }
	float f = 1.0f / a;
	v3 s = v_sub(o, tr->p0);
	float u = f * v_dot(s, h);
	if (u < 0.0f || u > 1.0f) { if (c) c->tri_exit_u++; return 0; }
	v3 q = v_cross(s, e0);
	float v = f * v_dot(d, q);
	if (v < 0.0f || u + v > 1.0f) { if (c) c->tri_exit_v++; return 0; }
	if (c) c->tri_reach_t++;
	float t0 = f * v_dot(e1, q);
	if (t0 > tmin && t0 < tmax)
	{
		if (c) c->tri_accept++;
		*t = t0;
		return 1;
	}
	return 0;
}

/* gpu_kernel.cl:534-555 with invDirection = 1/d per component (:229-232) */
static int bbox_hit(v3 mn, v3 mx, v3 o, v3 d)
{
	float ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
	float tx1 = (mn.x - o.x) * ix, tx2 = (mx.x - o.x) * ix;
	float tmin = fminf(tx1, tx2), tmax = fmaxf(tx1, tx2);
	float ty1 = (mn.y - o.y) * iy, ty2 = (mx.y - o.y) * iy;
	tmin = fmaxf(tmin, fminf(ty1, ty2));
	tmax = fminf(tmax, fmaxf(ty1, ty2));
	float tz1 = (mn.z - o.z) * iz, tz2 = (mx.z - o.z) * iz;
	tmin = fmaxf(tmin, fminf(tz1, tz2));
	tmax = fminf(tmax, fmaxf(tz1, tz2));
	return tmax > fmaxf(tmin, 0.0f);
}

/* gpu_kernel.cl:358-394: spheres first, then every triangle whose boundingBoxId
 * equals the *index* of each bounding box the ray hits.  The tMax argument of the
 * reference is ignored there (closestSoFar starts at INFINITY), so it is here. */
static int closest_hit(const wbo_scene *sc, v3 o, v3 d, float tmin, float *t_out, wbo_counters *c)
{
	float closest = INFINITY;
	int id = -1;
	float t;
	for (int i = 0; i < sc->S; i++)
		if (sphere_root(&sc->spheres[i], o, d, tmin, closest, &t, c)) { closest = t; id = i; }
	for (int b = 0; b < sc->B; b++)
	{
		if (c) c->bbox_tests++;
		if (!bbox_hit(sc->bbs[b].min, sc->bbs[b].max, o, d)) continue;
		if (c) c->bbox_hits++;
		for (int j = 0; j < sc->T; j++)
		{
			if (sc->tris[j].boundingBoxId != b) continue;
			if (triangle_t(&sc->tris[j], o, d, tmin, closest, &t, c)) { closest = t; id = sc->S + j; }
		}
	}
	*t_out = closest;
	return id;
}

typedef struct { v3 p, normal; int front; const wbo_material *m; } surface;

/* Hit point, outward normal and face flag for the winning primitive:
 * gpu_kernel.cl:480-484 (sphere), :522-526 (triangle), :352-356 */
static void surface_at(const wbo_scene *sc, int id, v3 o, v3 d, float t, surface *sf)
{
	v3 p = v_add(o, v_scale(d, t));
	v3 outward;
	if (id < sc->S)
	{
		const wbo_sphere *s = &sc->spheres[id];
		outward = v_div(v_sub(p, s->center), s->radius);
		sf->m = &s->material;
	}
	else
	{
		const wbo_triangle *tr = &sc->tris[id - sc->S];
		v3 e0 = v_sub(tr->p1, tr->p0);
		v3 e1 = v_sub(tr->p2, tr->p0);
		outward = v_unit(v_cross(e0, e1));
		sf->m = &tr->material;
	}
	sf->p = p;
	sf->front = v_dot(d, outward) < 0;
	sf->normal = sf->front ? outward : v_scale(outward, -1);
}

/* ------------------------------------------------------------------ sampling */
/* gpu_kernel.cl:182-196 */
static v3 rand_in_unit_sphere(lane *L)
{
	for (;;)
	{
		if (L->c) L->c->sphere_tries++;
		float x = draw(L) * 2.0f - 1.0f;
		float y = draw(L) * 2.0f - 1.0f;
		float z = draw(L) * 2.0f - 1.0f;
		v3 p = { x, y, z };
		if (v_len2(p) >= 1.0f) continue;
		return p;
	}
}

/* gpu_kernel.cl:243-257 */
static v3 rand_in_unit_disk(lane *L)
{
	for (;;)
	{
		if (L->c) L->c->disk_tries++;
		float x = draw(L) * 2.0f - 1.0f;
		float y = draw(L) * 2.0f - 1.0f;
		v3 p = { x, y, 0.0f };
		if (v_len2(p) >= 1.0f) continue;
		return p;
	}
}

/* gpu_kernel.cl:209-212 */
static v3 reflect(v3 v, v3 n) { return v_sub(v, v_scale(n, 2.0f * v_dot(v, n))); }

/* gpu_kernel.cl:214-220 */
static v3 refract(v3 uv, v3 n, float eta)
{
	float cos_t = fminf(v_dot(v_scale(uv, -1.0f), n), 1.0f);
	v3 perp = v_add(v_scale(uv, eta), v_scale(n, cos_t * eta));
	v3 par = v_scale(n, -sqrtf(fabsf(1.0f - v_len2(perp))));
	return v_add(perp, par);
}

/* gpu_kernel.cl:222-227 */
static float reflectance(int mode, float cosine, float ref_idx)
{
	float r0 = (1 - ref_idx) / (1 + ref_idx);
	r0 = r0 * r0;
	return r0 + (1 - r0) * m_pow5(mode, 1 - cosine);
}

/* gpu_kernel.cl:398-451.  Returns the scatter function's boolean. */
static int scatter(int kind, lane *L, v3 din, const surface *sf, v3 *atten, v3 *dout)
{
	const wbo_material *m = sf->m;
	if (kind == 0)
	{
		if (L->c) L->c->lambert++;
		v3 dir = v_add(sf->normal, v_unit(rand_in_unit_sphere(L)));
		const float eps = 1e-8;
		if (fabsf(dir.x) < eps && fabsf(dir.y) < eps && fabsf(dir.z) < eps) dir = sf->normal;
		*dout = dir;
		*atten = m->albedo;
		return 1;
	}
	if (kind == 1)
	{
		if (L->c) L->c->metal++;
		v3 refl = reflect(v_unit(din), sf->normal);
		if (m->fuzz > 0.0f)
		{
			if (L->c) L->c->metal_fuzz++;
			refl = v_add(refl, v_scale(rand_in_unit_sphere(L), m->fuzz));
		}
		*dout = refl;
		*atten = m->albedo;
		return v_dot(refl, sf->normal) > 0;
	}
	/* dielectric */
	if (L->c) L->c->dielectric++;
	v3 one = { 1.0f, 1.0f, 1.0f };
	*atten = one;
	float ratio = sf->front ? (1.0f / m->ior) : m->ior;
	v3 ud = v_unit(din);
	float cos_t = fminf(v_dot(v_scale(ud, -1), sf->normal), 1.0f);
	float sin_t = sqrtf(1.0f - cos_t * cos_t);
	int cannot = ratio * sin_t > 1.0f;
	int refl;
	if (cannot) refl = 1;
	else
	{
		if (L->c) L->c->dielectric_draw++;
		float fr = reflectance(L->math_mode, cos_t, ratio);
		refl = fr > draw(L);
	}
	*dout = refl ? reflect(ud, sf->normal) : refract(ud, sf->normal, ratio);
	return 1;
}

/* gpu_kernel.cl:234-241 + :342-345; index made safe for non-finite directions
 * (the reference would read out of bounds there). */
static void to_uv(int mode, v3 dir, float *uv)
{
	v3 n = v_unit(dir);
	uv[0] = 0.5f + m_atan2(mode, n.z, n.x) / (2.0f * (float)M_PI);
	uv[1] = 0.5f - m_asin(mode, n.y) / (float)M_PI;
}

static v3 sky_lookup(const wbo_scene *sc, int mode, v3 dir)
{
	float uv[2];
	to_uv(mode, dir, uv);
	float fx = uv[0] * sc->skyw;
	float fy = uv[1] * sc->skyh;
	int x = (fx >= 0.0f && fx <= (float)sc->skyw) ? (int)fx : 0;
	int y = (fy >= 0.0f && fy <= (float)sc->skyh) ? (int)fy : 0;
	return sc->sky[y * sc->skyw + x];
}

/* gpu_kernel.cl:266-348 */
static v3 ray_colour(const wbo_scene *sc, lane *L, v3 o, v3 d, int max_depth)
{
	v3 black = { 0, 0, 0 };
	v3 through = { 1.0f, 1.0f, 1.0f };
	int depth = 0;
	while (depth < max_depth)
	{
		float t;
		if (L->c) L->c->segments++;
		int id = closest_hit(sc, o, d, 0.001f, &t, L->c);
		if (id < 0) break;
		surface sf;
		surface_at(sc, id, o, d, t, &sf);
		int type = sf.m->type;
		if (type == 0 || type == 1 || type == 2)
		{
			v3 att, nd;
			if (scatter(type, L, d, &sf, &att, &nd))
			{
				o = sf.p;
				d = nd;
				through = v_mul(through, att);
				depth++;
				continue;
			}
			if (L->c) L->c->absorbed++;
			return type == 2 ? sf.m->albedo : black;
		}
		if (type == 3)
		{
			if (L->c) L->c->emissive++;
			return v_mul(through, sf.m->albedo);
		}
		depth++; /* unknown material: same ray again (:331-332) */
	}
	if (depth == max_depth)
	{
		if (L->c) L->c->depth_exhausted++;
		return black;
	}
	if (L->c) L->c->miss++;
	return v_mul(through, sky_lookup(sc, L->math_mode, d));
}

/* gpu_kernel.cl:559-578 */
static void camera_ray(const wbo_camera *cam, float u, float v, lane *L, v3 *o, v3 *d)
{
	v3 org = cam->origin;
	if (!(cam->aperture <= 0))
	{
		v3 p = rand_in_unit_disk(L);
		org = v_add(cam->origin, v_add(v_scale(cam->defocusDiscU, p.x), v_scale(cam->defocusDiscV, p.y)));
	}
	*d = v_sub(v_add(cam->lowerLeftCorner, v_add(v_scale(cam->horizontal, u), v_scale(cam->vertical, v))), org);
	*o = org;
}

/* gpu_kernel.cl:626-627 */
static void pixel_uv(const wbo_camera *cam, int gid, lane *L, float *u, float *v)
{
	int bsx = (int)cam->blockSize.x;
	int bsy = (int)cam->blockSize.y;
	float r0 = draw(L);
	*u = ((float)(gid % bsx) + (bsx * cam->blockOffset.x) + r0) / cam->width;
	float r1 = draw(L);
	*v = ((float)(gid / bsx) + (bsy * cam->blockOffset.y) + r1) / cam->height;
}

/* gpu_kernel.cl:595-648 for one work-item */
static void work_item(const wbo_scene *sc, const wbo_camera *cam, const uint32_t *seeds, int gid,
                      float *rgb, int out_mode, int math_mode, wbo_counters *c)
{
	lane L = { (uint64_t)seeds[gid], c, math_mode };
	v3 sum = { 0.0f, 0.0f, 0.0f };
	for (int i = 0; i < cam->samplesPerPixel; i++)
	{
		float u, v;
		v3 o, d;
		if (c) c->paths++;
		pixel_uv(cam, gid, &L, &u, &v);
		camera_ray(cam, u, v, &L, &o, &d);
		v3 col = ray_colour(sc, &L, o, d, cam->maxDepth);
		if (col.x > 1.0f) col.x = 1.0f;
		if (col.y > 1.0f) col.y = 1.0f;
		if (col.z > 1.0f) col.z = 1.0f;
		if (col.x < 0.0f) col.x = 0.0f;
		if (col.y < 0.0f) col.y = 0.0f;
		if (col.z < 0.0f) col.z = 0.0f;
		if (isnan(col.x)) col.x = sum.x; /* :640-642 quirk: adds the running sum */
		if (isnan(col.y)) col.y = sum.y;
		if (isnan(col.z)) col.z = sum.z;
		sum = v_add(sum, col);
	}
	if (out_mode == WBO_OUT_MEAN) sum = v_div(sum, cam->samplesPerPixel);
	rgb[3 * gid + 0] = sum.x;
	rgb[3 * gid + 1] = sum.y;
	rgb[3 * gid + 2] = sum.z;
}

/* ------------------------------------------------------------------ threading */
typedef struct
{
	const wbo_scene *sc;
	const wbo_camera *cam;
	const uint32_t *seeds;
	float *rgb;
	int32_t *ids;
	int first, count, out_mode, math_mode;
	int want_counters;
	atomic_int next;
	pthread_mutex_t mu;
	wbo_counters total;
} job_t;

static void add_counters(wbo_counters *dst, const wbo_counters *src)
{
	uint64_t *d = (uint64_t *)dst;
	const uint64_t *s = (const uint64_t *)src;
	for (int i = 0; i < WBO_NUM_COUNTERS; i++) d[i] += s[i];
}

static int primary_id(const wbo_scene *sc, const wbo_camera *cam, const uint32_t *seeds, int gid)
{
	lane L = { (uint64_t)seeds[gid], NULL, WBO_MATH_PORTABLE };
	float u, v, t;
	v3 o, d;
	pixel_uv(cam, gid, &L, &u, &v);
	camera_ray(cam, u, v, &L, &o, &d);
	return closest_hit(sc, o, d, 0.001f, &t, NULL);
}

static void *worker(void *arg)
{
	job_t *J = (job_t *)arg;
	wbo_counters local;
	memset(&local, 0, sizeof local);
	wbo_counters *c = J->want_counters ? &local : NULL;
	for (;;)
	{
		int begin = atomic_fetch_add(&J->next, 64);
		if (begin >= J->count) break;
		int end = begin + 64 < J->count ? begin + 64 : J->count;
		for (int k = begin; k < end; k++)
		{
			int gid = J->first + k;
			if (J->ids) J->ids[gid] = primary_id(J->sc, J->cam, J->seeds, gid);
			else work_item(J->sc, J->cam, J->seeds, gid, J->rgb, J->out_mode, J->math_mode, c);
		}
	}
	if (c)
	{
		pthread_mutex_lock(&J->mu);
		add_counters(&J->total, &local);
		pthread_mutex_unlock(&J->mu);
	}
	return NULL;
}

static int run_job(job_t *J, int nthreads)
{
	atomic_init(&J->next, 0);
	pthread_mutex_init(&J->mu, NULL);
	memset(&J->total, 0, sizeof J->total);
	if (nthreads <= 1) worker(J);
	else
	{
		pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
		if (!th) return -1;
		int started = 0;
		for (int i = 0; i < nthreads; i++)
		{
			if (pthread_create(&th[i], NULL, worker, J) != 0) break;
			started++;
		}
		if (!started) worker(J);
		for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
		free(th);
	}
	pthread_mutex_destroy(&J->mu);
	return 0;
}

int wbo_render_items(const wbo_scene *sc, const wbo_camera *cam, const uint32_t *seeds,
                     int first, int count, float *rgb, int out_mode, int math_mode,
                     wbo_counters *counters, int nthreads)
{
	job_t J;
	memset(&J, 0, sizeof J);
	J.sc = sc; J.cam = cam; J.seeds = seeds; J.rgb = rgb; J.ids = NULL;
	J.first = first; J.count = count; J.out_mode = out_mode; J.math_mode = math_mode;
	J.want_counters = counters != NULL;
	int rc = run_job(&J, nthreads);
	if (counters) add_counters(counters, &J.total);
	return rc;
}

int wbo_primary_ids(const wbo_scene *sc, const wbo_camera *cam, const uint32_t *seeds,
                    int first, int count, int32_t *ids)
{
	job_t J;
	memset(&J, 0, sizeof J);
	J.sc = sc; J.cam = cam; J.seeds = seeds; J.ids = ids;
	J.first = first; J.count = count;
	return run_job(&J, 1);
}

/* Block loop of main.cpp:222-260.  blockSize comes from the camera (floats). */
static int frame_loop(const wbo_scene *sc, const wbo_camera *cam, const uint32_t *seeds,
                      float *image, int32_t *ids, int out_mode, int math_mode,
                      wbo_counters *counters, int nthreads)
{
	int bw = (int)cam->blockSize.x, bh = (int)cam->blockSize.y;
	if (bw <= 0 || bh <= 0) return -2;
	int nbx = cam->width / bw, nby = cam->height / bh;
	int n = bw * bh;
	float *blk = image ? (float *)malloc(sizeof(float) * 3 * (size_t)n) : NULL;
	int32_t *bid = ids ? (int32_t *)malloc(sizeof(int32_t) * (size_t)n) : NULL;
	if ((image && !blk) || (ids && !bid)) { free(blk); free(bid); return -1; }
	for (int by = nby - 1; by >= 0; by--)
		for (int bx = 0; bx < nbx; bx++)
		{
			wbo_camera c = *cam;
			c.blockOffset.x = (float)bx; c.blockOffset.y = (float)by; c.blockOffset.z = 0.0f;
			if (image)
			{
				wbo_render_items(sc, &c, seeds, 0, n, blk, out_mode, math_mode, counters, nthreads);
				for (int i = 0; i < bh; i++)
					memcpy(&image[3 * ((size_t)(i + bh * by) * cam->width + (size_t)bw * bx)],
					       &blk[3 * (size_t)i * bw], sizeof(float) * 3 * (size_t)bw);
			}
			else
			{
				wbo_primary_ids(sc, &c, seeds, 0, n, bid);
				for (int i = 0; i < bh; i++)
					memcpy(&ids[(size_t)(i + bh * by) * cam->width + (size_t)bw * bx],
					       &bid[(size_t)i * bw], sizeof(int32_t) * (size_t)bw);
			}
		}
	free(blk);
	free(bid);
	return 0;
}

int wbo_render_frame(const wbo_scene *sc, const wbo_camera *cam, const uint32_t *seeds,
                     float *image, int out_mode, int math_mode, wbo_counters *counters, int nthreads)
{
	return frame_loop(sc, cam, seeds, image, NULL, out_mode, math_mode, counters, nthreads);
}

int wbo_primary_ids_frame(const wbo_scene *sc, const wbo_camera *cam, const uint32_t *seeds, int32_t *ids)
{
	return frame_loop(sc, cam, seeds, NULL, ids, 0, 0, NULL, 1);
}

/* ------------------------------------------------------- KAT entry points */
int wbo_closest_hit(const wbo_scene *sc, const wbo_ray *ray, float *t_out, wbo_counters *c)
{
	return closest_hit(sc, ray->origin, ray->direction, 0.001f, t_out, c);
}

int wbo_hit_sphere(const wbo_sphere *s, const wbo_ray *r, float tmin, float tmax, float *t_out)
{
	return sphere_root(s, r->origin, r->direction, tmin, tmax, t_out, NULL);
}

int wbo_hit_triangle(const wbo_triangle *t, const wbo_ray *r, float tmin, float tmax, float *t_out)
{
	return triangle_t(t, r->origin, r->direction, tmin, tmax, t_out, NULL);
}

int wbo_hit_bbox(const wbo_bounding_box *b, const wbo_ray *r)
{
	return bbox_hit(b->min, b->max, r->origin, r->direction);
}

void wbo_get_ray(const wbo_camera *cam, float u, float v, uint64_t *s, wbo_ray *out)
{
	lane L = { *s, NULL, WBO_MATH_PORTABLE };
	camera_ray(cam, u, v, &L, &out->origin, &out->direction);
	*s = L.seed;
}

int wbo_scatter(int kind, const wbo_ray *in, const wbo_vec3 *p, const wbo_vec3 *normal, int front_face,
                const wbo_material *m, uint64_t *s, int math_mode, wbo_vec3 *atten, wbo_ray *out)
{
	lane L = { *s, NULL, math_mode };
	surface sf = { *p, *normal, front_face, m };
	int ok = scatter(kind, &L, in->direction, &sf, atten, &out->direction);
	out->origin = *p;
	*s = L.seed;
	return ok;
}

void wbo_ray_colour(const wbo_scene *sc, const wbo_ray *ray, int max_depth, uint64_t *s, int math_mode, wbo_vec3 *rgb)
{
	lane L = { *s, NULL, math_mode };
	*rgb = ray_colour(sc, &L, ray->origin, ray->direction, max_depth);
	*s = L.seed;
}

void wbo_to_uv(const wbo_vec3 *dir, int math_mode, float *uv) { to_uv(math_mode, *dir, uv); }

/* colour.cpp:5-15: gamma 2, *255, truncating cast.  RGB order here. */
void wbo_tonemap_rgb8(const float *image, int n_pixels, uint8_t *rgb8)
{
	for (int i = 0; i < 3 * n_pixels; i++)
	{
		float g = sqrtf(image[i]) * 255.0f;
		rgb8[i] = (unsigned char)g;
	}
}
