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

struct F3 { float x, y, z; };

__device__ __forceinline__ F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ F3 operator/(F3 a, float s) { return f3(a.x / s, a.y / s, a.z / s); }
// (x*x' + y*y') + z*z'  (gpu_kernel.cl:153-166)
__device__ __forceinline__ float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross(F3 a, F3 b)
{
	return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ F3 unit(F3 a) { return a / sqrtf(dot(a, a)); }
__device__ __forceinline__ F3 ld3(const float *p) { return f3(p[0], p[1], p[2]); }

// Wave-uniform read of read-only scene data (same index in every lane): going through the
// constant address space makes it a scalar load into SGPRs instead of a 64-lane vector load of
// one address.  Only for buffers no kernel writes.
typedef float wbrt_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 uniform_ld(const float4 *p, int i)
{
#if __HIP_DEVICE_COMPILE__
	const wbrt_v4f v = ((const __attribute__((address_space(4))) wbrt_v4f *)p)[i];
	return make_float4(v.x, v.y, v.z, v.w);
#else
	return p[i]; // host pass of the compiler only; never executed
#endif
}

// ---------------------------------------------------------------- counters
template <bool ON> struct Counters;
template <> struct Counters<false>
{
	__device__ __forceinline__ void add(int, unsigned) {}
	__device__ __forceinline__ void flush(unsigned long long *) {}
};
template <> struct Counters<true>
{
	unsigned v[WBRT_NUM_COUNTERS];
	unsigned long long tot[WBRT_NUM_COUNTERS];
	__device__ Counters()
	{
		for (int i = 0; i < WBRT_NUM_COUNTERS; ++i) { v[i] = 0; tot[i] = 0; }
	}
	__device__ __forceinline__ void add(int slot, unsigned n)
	{
		v[slot] += n;
		if (v[slot] > 0x7fffffffu) { tot[slot] += v[slot]; v[slot] = 0; }
	}
	__device__ void flush(unsigned long long *g)
	{
		for (int i = 0; i < WBRT_NUM_COUNTERS; ++i)
		{
			unsigned long long x = tot[i] + v[i];
			if (x) atomicAdd(&g[i], x);
		}
	}
};

// --------------------------------------------------------------------- RNG
// gpu_kernel.cl:582-591: 48-bit LCG s' = (s*0x5DEECE66D + 0xB) mod 2^48; the float is
// bits 47..16 of the new state converted with round-to-nearest and divided by 2^32.
// The 48-bit state s is held shifted left by 16 in a 64-bit pair: hi = bits 47..16 of s (exactly the
// integer the float is made of), lo = bits 15..0 of s in its upper half.  (s << 16) * M mod 2^64 is
// (s * M mod 2^48) << 16, so with M = 0xDEECE66D + 5*2^32 one step is
//   lo*0xDEECE66D + (0xB << 16)  (64-bit)  +  ((lo*5 + hi*0xDEECE66D) << 32).
// Scaling by a power of two is exact, so callers fold the 2^-32 into a fused multiply-add
// (see unit()/bipolar()): fma(top, 2^-31, -1) is bit-identical to (top*2^-32)*2 - 1.
struct Rng
{
	unsigned lo, hi;
	__device__ __forceinline__ void seed(unsigned s32) { lo = s32 << 16; hi = s32 >> 16; }
	// advances the state, returns float(state >> 16) (an integer value in [0, 2^32])
	__device__ __forceinline__ float next_top()
	{
		const unsigned long long p = (unsigned long long)lo * 0xDEECE66Du + 0xB0000u;
		hi = (unsigned)(p >> 32) + (lo << 2) + lo + hi * 0xDEECE66Du;
		lo = (unsigned)p;
		return (float)hi;
	}
	// the reference's draw in [0,1]
	__device__ __forceinline__ float next() { return next_top() * 2.3283064365386963e-10f; }
	// draw*2 - 1  (gpu_kernel.cl:187-189, :248-249)
	__device__ __forceinline__ float bipolar() { return __builtin_fmaf(next_top(), 4.6566128730773926e-10f, -1.0f); }
};

// ---------------------------------------------- portable transcendental math
// atan2 / asin / x^5 as fixed double-precision sequences of + - * / sqrt (no contraction).
// OpenCL leaves the last bits of these builtins to the implementation; this sequence is
// evaluated identically on the host, so a sky texel index never depends on the device.
__device__ const double ATAN_K8[9] = {
	0x0.0p+0,             0x1.fd5ba9aac2f6ep-4, 0x1.f5b75f92c80ddp-3,
	0x1.6f61941e4def1p-2, 0x1.dac670561bb4fp-2, 0x1.1e00babdefeb4p-1,
	0x1.4978fa3269ee1p-1, 0x1.700a7c5784634p-1, 0x1.921fb54442d18p-1
};
#define PT_PI   0x1.921fb54442d18p+1
#define PT_PI_2 0x1.921fb54442d18p+0

__device__ __noinline__ double atan2_core(double y, double x)
{
	if (x != x || y != y) return x + y;
	double ax = fabs(x), ay = fabs(y);
	double hi = ax > ay ? ax : ay;
	double lo = ax > ay ? ay : ax;
	double a = 0.0;
	if (hi > 0.0)
	{
		if (hi == (double)INFINITY) a = (lo == (double)INFINITY) ? 1.0 : 0.0;
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
	if (__builtin_signbit(x)) r = PT_PI - r;
	if (__builtin_signbit(y)) r = -r;
	return r;
}

__device__ __forceinline__ float pt_atan2f(float y, float x) { return (float)atan2_core((double)y, (double)x); }

__device__ __forceinline__ float pt_asinf(float y)
{
	double d = (double)y;
	if (!(fabs(d) <= 1.0)) return NAN;
	double c = sqrt((1.0 - d) * (1.0 + d));
	return (float)atan2_core(d, c);
}

__device__ __forceinline__ float pt_pow5f(float x)
{
	double d = (double)x;
	double d2 = d * d;
	double d4 = d2 * d2;
	return (float)(d4 * d);
}

// gpu_kernel.cl:234-241 and :342-347.  Non-finite coordinates fall back to texel 0
// (the reference would read out of bounds).
__device__ __forceinline__ F3 sky_lookup(const SceneDev &sc, F3 dir)
{
	F3 n = unit(dir);
	float u = 0.5f + pt_atan2f(n.z, n.x) / (2.0f * 3.14159274101257324f);
	float v = 0.5f - pt_asinf(n.y) / 3.14159274101257324f;
	float fx = u * (float)sc.skyw;
	float fy = v * (float)sc.skyh;
	int x = (fx >= 0.0f && fx <= (float)sc.skyw) ? (int)fx : 0;
	int y = (fy >= 0.0f && fy <= (float)sc.skyh) ? (int)fy : 0;
	return ld3(sc.sky + 3 * (size_t)(y * sc.skyw + x));
}


// IEEE-correct num/den for three numerators that share one reciprocal refinement.
// This is the hardware's own correctly-rounded division sequence (v_rcp_f32 + one Newton
// step, then quotient + two residual corrections) without the operand scaling, which is
// the identity while every operand sits well inside the normal range; outside it (or for
// zero / non-finite operands) the compiler's full IEEE division is used instead.
__device__ __forceinline__ F3 div3(F3 num, float den)
{
	// all four magnitudes inside [2^-40, 2^40]; a NaN operand slips through min/max and yields NaN from
	// the sequence below as it does from a division
	const float ad = fabsf(den), ax = fabsf(num.x), ay = fabsf(num.y), az = fabsf(num.z);
	const float smallest = fminf(fminf(fminf(ax, ay), az), ad);
	const float largest = fmaxf(fmaxf(fmaxf(ax, ay), az), ad);
	const bool safe = smallest >= 0x1p-40f && largest <= 0x1p40f;
	float y = __builtin_amdgcn_rcpf(den);
	const float e = __builtin_fmaf(-den, y, 1.0f);
	y = __builtin_fmaf(e, y, y);
	F3 q;
	{
		float q0 = num.x * y; float r = __builtin_fmaf(-den, q0, num.x); q0 = __builtin_fmaf(r, y, q0);
		r = __builtin_fmaf(-den, q0, num.x); q.x = __builtin_fmaf(r, y, q0);
	}
	{
		float q0 = num.y * y; float r = __builtin_fmaf(-den, q0, num.y); q0 = __builtin_fmaf(r, y, q0);
		r = __builtin_fmaf(-den, q0, num.y); q.y = __builtin_fmaf(r, y, q0);
	}
	{
		float q0 = num.z * y; float r = __builtin_fmaf(-den, q0, num.z); q0 = __builtin_fmaf(r, y, q0);
		r = __builtin_fmaf(-den, q0, num.z); q.z = __builtin_fmaf(r, y, q0);
	}
	if (!safe) q = num / den;
	return q;
}
__device__ __forceinline__ F3 unit_fast(F3 a) { return div3(a, sqrtf(dot(a, a))); }

// IEEE-correct x/den for 0 <= x < 2^24 and an integer-valued 1 <= den <= 2^24 whose
// reciprocal rcp (within 1 ulp) is precomputed: same quotient + two residual corrections as
// div3; every intermediate stays in the normal range for such operands.
__device__ __forceinline__ float div_by_const(float x, float den, float rcp)
{
	float q = x * rcp;
	float r = __builtin_fmaf(-den, q, x);
	q = __builtin_fmaf(r, rcp, q);
	r = __builtin_fmaf(-den, q, x);
	q = __builtin_fmaf(r, rcp, q);
	const bool safe = x < 16777216.0f && (x == 0.0f || x > 0x1p-60f);
	return safe ? q : x / den;
}

// atan(a) for a in [0,1]: odd polynomial a*P(a^2), absolute error < 1.1e-7.
__device__ __forceinline__ float atan01(float a)
{
	const float s = a * a;
	float p = 2.834064187e-03f;
	p = __builtin_fmaf(p, s, -1.600502990e-02f);
	p = __builtin_fmaf(p, s, 4.258760810e-02f);
	p = __builtin_fmaf(p, s, -7.495445758e-02f);
	p = __builtin_fmaf(p, s, 1.063675433e-01f);
	p = __builtin_fmaf(p, s, -1.420257092e-01f);
	p = __builtin_fmaf(p, s, 1.999248415e-01f);
	p = __builtin_fmaf(p, s, -3.333306611e-01f);
	p = __builtin_fmaf(p, s, 1.0f);
	return a * p;
}

// Sky texel index without the exact (double precision) path.  The texel is an integer, so
// an approximation of the texel coordinates decides it whenever they are farther from a
// texel boundary than the approximation's error bound; otherwise (rare) it reports failure
// and the caller evaluates the defining sequence in sky_lookup().  Bounds: |u_fast - u| <=
// 3.4e-7 and |v_fast - v| <= 4.2e-7 + 6.4e-8/cos(elevation) against the defining float
// pipeline (its own roundings included); the bands below use 1e-6 and 1e-6 + 2e-7/cos.
__device__ __forceinline__ bool sky_texel_fast(const SceneDev &sc, F3 d, int &idx)
{
	const float PI = 3.14159274101257324f, PI_2 = 1.57079637050628662f;
	const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
	const float h2 = d.x * d.x + d.z * d.z;
	const float l2 = h2 + d.y * d.y;
	// azimuth atan2(z, x)
	float hi = fmaxf(ax, az), lo = fminf(ax, az);
	float r = atan01(lo * __builtin_amdgcn_rcpf(hi));
	if (az > ax) r = PI_2 - r;
	if (d.x < 0.0f) r = PI - r;
	r = copysignf(r, d.z); // d.z != 0 whenever the result is used
	const float fx = __builtin_fmaf(r, 0.159154943091895346f, 0.5f) * (float)sc.skyw;
	// elevation atan2(y, sqrt(x^2+z^2)) == asin(y/|d|)
	const float hd = __builtin_amdgcn_sqrtf(h2);
	hi = fmaxf(ay, hd); lo = fminf(ay, hd);
	float e = atan01(lo * __builtin_amdgcn_rcpf(hi));
	if (ay > hd) e = PI_2 - e;
	e = copysignf(e, d.y); // e == 0 when d.y is a zero of either sign
	const float fy = __builtin_fmaf(-e, 0.318309886183790691f, 0.5f) * (float)sc.skyh;
	const float inv_cos = __builtin_amdgcn_sqrtf(l2) * __builtin_amdgcn_rcpf(hd);
	const float bx = 1e-6f * (float)sc.skyw;
	const float by = __builtin_fmaf(2e-7f, inv_cos, 1e-6f) * (float)sc.skyh;
	const float x0 = floorf(fx - bx), x1 = floorf(fx + bx);
	const float y0 = floorf(fy - by), y1 = floorf(fy + by);
	const bool ok = x0 == x1 && y0 == y1 && x0 >= 0.0f && y0 >= 0.0f && x1 <= (float)(sc.skyw - 1) &&
	                y1 <= (float)(sc.skyh - 1) && d.x != 0.0f && d.z != 0.0f && l2 > 1e-20f && l2 < 1e20f;
	idx = (int)y0 * sc.skyw + (int)x0;
	return ok;
}

// Closest sphere without per-sphere exact square roots and divisions.  All roots of one
// segment share the divisor a = |d|^2 > 0 and correctly rounded division is monotonic, so
// roots are ordered like their numerators n = -hb -+ sqrt(disc) unless two numerators are
// within rounding distance.  The loop works on approximate numerators (hardware sqrt, 1 ulp)
// with explicit error margins: every decision the reference takes (first root below tMin?
// second root below tMin? closer than the best so far, later sphere winning ties) is taken
// only when the margins make it certain; otherwise the lane reports failure and the caller
// runs the exact loop.  The winner's root is then evaluated exactly, once.
// Straight-line form: every sphere yields an interval [lo, hi] around the approximate numerator
// of the root the reference would take (lo = +inf when it certainly takes none); the loop keeps the
// smallest and second smallest lower bound.  The candidate with the smallest lower bound is the
// reference's winner whenever its lower bound plus twice the largest margin of the segment lies below every
// other lower bound.  A root whose comparison with tMin falls inside the margin gets lo = -1, which fails that check.
// Bounds are positive floats, +inf or negative, so they are ordered as signed integers.
// Requires |o| < 1e8, 1e-20 < a < 1e16 and sphere data below 1e8 (no overflow, no NaN in disc).
struct SphereScan
{
	float a, T_lo, T_hi;
	float m_all, bhb, bsd;
	int min1, min2, id;
	__device__ __forceinline__ void begin(float a_)
	{
		a = a_;
		const float T = 0.001f * a;
		T_lo = T * (1.0f - 4e-7f); T_hi = T * (1.0f + 4e-7f);
		min1 = min2 = 0x7f800000; m_all = 0.0f; bhb = bsd = 0.0f; id = -1;
	}
	__device__ __forceinline__ void test(F3 o, F3 d, float gx, float gy, float gz, float gw, int i)
	{
		F3 oc = o - f3(gx, gy, gz);
		float hb = dot(oc, d);
		float cc = dot(oc, oc) - gw;
		float disc = hb * hb - a * cc;
		if (!__ballot(!(disc < 0.0f))) return; // no lane of the wave reaches this sphere
		// hardware square root: 1 ulp, (flushed) tiny arguments are covered by the absolute slack; NaN for
		// a miss, and every comparison below is written so that NaN means "no root here"
		const float sq = __builtin_amdgcn_sqrtf(disc);
		const float n1 = -hb - sq, n2 = -hb + sq;
		// |approximate - exact numerator| (1.5 ulp of the root, roundings of the sums) + the distance
		// below which two quotients may round to the same float: at most 5e-7 (|hb| + sq)
		const float m = __builtin_fmaf(5e-7f, fabsf(hb) + sq, 1e-14f);
		const float lo1 = n1 - m, hi1 = n1 + m, lo2 = n2 - m, hi2 = n2 + m;
		const bool f_ok = lo1 > T_hi, f_bad = !(hi1 >= T_lo);   // first root certainly >= tMin / < tMin
		const bool s_ok = lo2 > T_hi, s_bad = !(hi2 >= T_lo);
		// lower bound of the numerator of the root the reference takes: +inf when it certainly takes none,
		// -1 (which fails the final check) when a comparison with tMin is inside the margin
		float lo = (f_bad & s_bad) ? INFINITY : -1.0f;
		lo = (f_bad & s_ok) ? lo2 : lo;
		lo = f_ok ? lo1 : lo;
		m_all = fmaxf(m_all, m);                                // the largest margin of the segment (a miss: NaN, ignored)
		const int ilo = __float_as_int(lo);
		const bool less = ilo < min1;
		min2 = min(min2, max(min1, ilo));
		min1 = min(min1, ilo);
		bhb = less ? hb : bhb;
		bsd = less ? disc : bsd;
		id = less ? i : id;
	}
	__device__ __forceinline__ bool finish(bool in_range, float &t_out, int &id_out)
	{
		id_out = id;
		t_out = INFINITY;
		if (id >= 0)
		{
			// the reference's own choice of root (gpu_kernel.cl:468-478), which the margins made certain
			const float sq = sqrtf(bsd);
			t_out = (-bhb - sq) / a;
			if (t_out < 0.001f) t_out = (-bhb + sq) / a;
		}
		// the winner's numerator lies in [min1, min1 + 2 margins]; every other candidate's at or above min2
		const bool apart = min1 >= 0 && __builtin_fmaf(2.0f, m_all, __int_as_float(min1)) < __int_as_float(min2);
		return in_range & ((id < 0) | apart);
	}
};

__device__ __forceinline__ bool spheres_fast(const SceneDev &sc, F3 o, F3 d, float a, float &t_out, int &id_out)
{
	SphereScan s;
	s.begin(a);
	const bool in_range = sc.sph_moderate && a > 1e-20f && a < 1e16f && fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) < 1e8f;
	// the scalar loads of the next two spheres are in flight while one sphere is tested
	const int last = sc.S - 1;
	float4 g_next = sc.S > 0 ? uniform_ld(sc.sph_geom, 0) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	float4 g_next2 = sc.S > 0 ? uniform_ld(sc.sph_geom, 1 < last ? 1 : last) : g_next;
	for (int i = 0; i < sc.S; ++i)
	{
		const float4 g = g_next;
		g_next = g_next2;
		g_next2 = uniform_ld(sc.sph_geom, i + 2 < last ? i + 2 : last);
		s.test(o, d, g.x, g.y, g.z, g.w, i);
	}
	return s.finish(in_range, t_out, id_out);
}

// ------------------------------------------------------------ intersection
// gpu_kernel.cl:358-394 + :455-478 + :489-520 + :534-555.  Returns the primitive id
// (sphere index, S + grouped triangle index, -1) and the accepted t.
template <bool STATS>
__device__ __forceinline__ void spheres_exact(const SceneDev &sc, F3 o, F3 d, float a, float &closest, int &id, Counters<STATS> &ctr)
{
	const float tmin = 0.001f;
	closest = INFINITY;
	id = -1;
	for (int i = 0; i < sc.S; ++i)
	{
		const float4 g = sc.sph_geom[i];
		F3 oc = o - f3(g.x, g.y, g.z);
		float hb = dot(oc, d);
		float cc = dot(oc, oc) - g.w;
		float disc = hb * hb - a * cc;
		if (disc < 0) continue;
		if (STATS) ctr.add(C_SPHERE_DISC_GE0, 1);
		float sq = sqrtf(disc);
		float root = (-hb - sq) / a;
		if (root < tmin || closest < root)
		{
			if (STATS) ctr.add(C_SPHERE_SECOND_ROOT, 1);
			root = (-hb + sq) / a;
			if (root < tmin || closest < root) continue;
		}
		if (STATS) ctr.add(C_SPHERE_ACCEPT, 1);
		closest = root;
		id = i;
	}
}

// Wave-private LDS staging area for triangle tiles: two buffers of up to 64 triangles, each
// triangle three float4 {p0.xyz e0.x | e0.yz e1.xy | e1.z - - -}, component-major so that the
// staging writes of a wave are contiguous and the per-triangle reads are broadcasts.
struct TriTiles { float4 v[2][3][64]; };

// gpu_kernel.cl:374-391 + :489-520 + :534-555.  Triangles of a hit box are streamed through
// LDS: every active lane of the wave loads ONE triangle of the next tile from global memory
// (coalesced 16-byte loads, issued before the current tile is processed), the wave then walks
// the staged tile with broadcast LDS reads.  No lane waits on a per-triangle memory access.
template <bool STATS>
__device__ __forceinline__ void boxes_and_triangles(const SceneDev &sc, TriTiles &tiles, F3 o, F3 d,
                                                    float &closest, int &id, Counters<STATS> &ctr)
{
	const float tmin = 0.001f;
	if (STATS) ctr.add(C_BBOX_TESTS, (unsigned)sc.B);
	const float ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
	// lanes executing this call (live pixel streams of the wave); each stages one triangle per tile
	const unsigned long long active = __ballot(true);
	const int n_active = __popcll(active);
	const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(active >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)active, 0u));
	for (int b = 0; b < sc.B; ++b)
	{
		const float4 mn = uniform_ld(sc.box_min, b), mx = uniform_ld(sc.box_max, b);
		float tx1 = (mn.x - o.x) * ix, tx2 = (mx.x - o.x) * ix;
		float lo = fminf(tx1, tx2), hi = fmaxf(tx1, tx2);
		float ty1 = (mn.y - o.y) * iy, ty2 = (mx.y - o.y) * iy;
		lo = fmaxf(lo, fminf(ty1, ty2));
		hi = fminf(hi, fmaxf(ty1, ty2));
		float tz1 = (mn.z - o.z) * iz, tz2 = (mx.z - o.z) * iz;
		lo = fmaxf(lo, fminf(tz1, tz2));
		hi = fminf(hi, fmaxf(tz1, tz2));
		const bool hit_box = hi > fmaxf(lo, 0.0f);
		if (STATS && hit_box) ctr.add(C_BBOX_HITS, 1);
		if (!__ballot(hit_box)) continue; // no lane of the wave needs this box
		const int2 range = sc.box_range[b];
		int buf = 0;
		float4 n0 = make_float4(0, 0, 0, 0), n1 = n0, n2 = n0;
		if (range.x + rank < range.y)
		{
			const float4 *src = sc.tri_geom + 3 * (size_t)(range.x + rank);
			n0 = src[0]; n1 = src[1]; n2 = src[2];
		}
		for (int base = range.x; base < range.y; base += n_active)
		{
			tiles.v[buf][0][rank] = n0;
			tiles.v[buf][1][rank] = n1;
			tiles.v[buf][2][rank] = n2;
			__builtin_amdgcn_wave_barrier(); // keep the staging stores ahead of the tile reads
			const int next = base + n_active + rank;
			if (next < range.y)
			{
				const float4 *src = sc.tri_geom + 3 * (size_t)next;
				n0 = src[0]; n1 = src[1]; n2 = src[2];
			}
			const int count = range.y - base < n_active ? range.y - base : n_active;
			if (hit_box)
			{
				for (int j = 0; j < count; ++j)
				{
					const float4 g0 = tiles.v[buf][0][j];
					const float4 g1 = tiles.v[buf][1][j];
					const float4 g2 = tiles.v[buf][2][j];
					const F3 p0 = f3(g0.x, g0.y, g0.z), e0 = f3(g0.w, g1.x, g1.y), e1 = f3(g1.z, g1.w, g2.x);
					if (STATS) ctr.add(C_TRI_TESTS, 1);
					F3 h = cross(d, e1);
					float det = dot(e0, h);
					if (det > -0.00001f && det < 0.00001f) { if (STATS) ctr.add(C_TRI_EXIT_PARALLEL, 1); continue; }
					float f = 1.0f / det;
					F3 s = o - p0;
					float u = f * dot(s, h);
					if (u < 0.0f || u > 1.0f) { if (STATS) ctr.add(C_TRI_EXIT_U, 1); continue; }
					F3 q = cross(s, e0);
					float v = f * dot(d, q);
					if (v < 0.0f || u + v > 1.0f) { if (STATS) ctr.add(C_TRI_EXIT_V, 1); continue; }
					if (STATS) ctr.add(C_TRI_REACH_T, 1);
					float t0 = f * dot(e1, q);
					if (t0 > tmin && t0 < closest)
					{
						if (STATS) ctr.add(C_TRI_ACCEPT, 1);
						closest = t0;
						id = sc.S + base + j;
					}
				}
			}
			__builtin_amdgcn_wave_barrier();
			buf ^= 1;
		}
	}
}

// Triangle phase for the render build: every lane of the wave (helper lanes whose own pixel
// stream is finished included) loads ONE triangle of the current 64-triangle tile straight
// into registers (coalesced 16-byte loads, next tile prefetched), then the rays of the lanes
// that hit the box are broadcast one at a time (v_readlane -> scalar operands) and each lane
// tests its triangle against that ray.  All 64 lanes work whatever the number of rays, and a
// tile is fetched once per wave, not once per ray.  A hit posts (t, triangle index) to the
// ray's slot in LDS with a 64-bit atomic min: the reference's sequential rule — accept
// tMin < t0 < closest, first triangle wins ties (gpu_kernel.cl:382-387,520) — selects exactly
// the lexicographic minimum of (t0, index) among candidates below the sphere result.
__device__ __forceinline__ float lane_bcast(float v, int lane)
{
	return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__device__ __forceinline__ void triangles_cooperative(const SceneDev &sc, unsigned long long *best, bool trace,
                                                      F3 o, F3 d, float &closest, int &id)
{
	const float tmin = 0.001f;
	const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
	best[lane] = ((unsigned long long)__float_as_uint(closest) << 32) | 0xFFFFFFFFull;
	__builtin_amdgcn_wave_barrier();
	const float ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
	for (int b = 0; b < sc.B; ++b)
	{
		const float4 mn = uniform_ld(sc.box_min, b), mx = uniform_ld(sc.box_max, b);
		float tx1 = (mn.x - o.x) * ix, tx2 = (mx.x - o.x) * ix;
		float lo = fminf(tx1, tx2), hi = fmaxf(tx1, tx2);
		float ty1 = (mn.y - o.y) * iy, ty2 = (mx.y - o.y) * iy;
		lo = fmaxf(lo, fminf(ty1, ty2));
		hi = fminf(hi, fmaxf(ty1, ty2));
		float tz1 = (mn.z - o.z) * iz, tz2 = (mx.z - o.z) * iz;
		lo = fmaxf(lo, fminf(tz1, tz2));
		hi = fminf(hi, fmaxf(tz1, tz2));
		const unsigned long long rays = __ballot(trace && hi > fmaxf(lo, 0.0f));
		if (!rays) continue;
		const int2 range = sc.box_range[b];
		float4 n0 = make_float4(0, 0, 0, 0), n1 = n0, n2 = n0;
		if (range.x + lane < range.y)
		{
			const float4 *src = sc.tri_geom + 3 * (size_t)(range.x + lane);
			n0 = src[0]; n1 = src[1]; n2 = src[2];
		}
		for (int base = range.x; base < range.y; base += 64)
		{
			const float4 g0 = n0, g1 = n1, g2 = n2;
			const int tri = base + lane;
			const bool have = tri < range.y;
			if (tri + 64 < range.y)
			{
				const float4 *src = sc.tri_geom + 3 * (size_t)(tri + 64);
				n0 = src[0]; n1 = src[1]; n2 = src[2];
			}
			const F3 p0 = f3(g0.x, g0.y, g0.z), e0 = f3(g0.w, g1.x, g1.y), e1 = f3(g1.z, g1.w, g2.x);
			for (unsigned long long m = rays; m; m &= m - 1)
			{
				const int r = __builtin_ctzll(m);
				const F3 ro = f3(lane_bcast(o.x, r), lane_bcast(o.y, r), lane_bcast(o.z, r));
				const F3 rd = f3(lane_bcast(d.x, r), lane_bcast(d.y, r), lane_bcast(d.z, r));
				const float rc = lane_bcast(closest, r);
				if (!have) continue;
				F3 h = cross(rd, e1);
				float det = dot(e0, h);
				if (det > -0.00001f && det < 0.00001f) continue;
				float f = 1.0f / det;
				F3 s = ro - p0;
				float u = f * dot(s, h);
				if (u < 0.0f || u > 1.0f) continue;
				F3 q = cross(s, e0);
				float v = f * dot(rd, q);
				if (v < 0.0f || u + v > 1.0f) continue;
				float t0 = f * dot(e1, q);
				if (t0 > tmin && t0 < rc)
					atomicMin(&best[r], ((unsigned long long)__float_as_uint(t0) << 32) | (unsigned)tri);
			}
		}
	}
	__builtin_amdgcn_wave_barrier();
	const unsigned long long k = best[lane];
	if (trace && (unsigned)k != 0xFFFFFFFFu)
	{
		closest = __uint_as_float((unsigned)(k >> 32));
		id = sc.S + (int)(unsigned)k;
	}
}

// Walk of the triangle hierarchies (built on upload, see rt_bvh.cpp) for the render build.  A box of the scene is entered only when the reference's own slab test passes
// (gpu_kernel.cl:534-555), exactly like its triangle loop; inside, the triangle test is the
// reference's, and the winner is the lexicographic minimum of (t0, loop position) among
// accepted hits below the sphere result — what the sequential loop with its strict
// "t0 < closest" keeps (gpu_kernel.cl:382-387,520).  Node boxes are widened per ray by a bound
// on how far outside its triangle the float test can still accept a hit, so skipping a node
// never drops a triangle the reference would have accepted.

// o_near = o + margin, o_far = o - margin: the box [mn - margin, mx + margin] seen from o
__device__ __forceinline__ bool bvh_box(F3 o_near, F3 o_far, F3 inv, float mnx, float mny, float mnz, float mxx, float mxy, float mxz,
                                        float best, float &lo)
{
	const float tx1 = (mnx - o_near.x) * inv.x, tx2 = (mxx - o_far.x) * inv.x;
	const float ty1 = (mny - o_near.y) * inv.y, ty2 = (mxy - o_far.y) * inv.y;
	const float tz1 = (mnz - o_near.z) * inv.z, tz2 = (mxz - o_far.z) * inv.z;
	lo = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fminf(tz1, tz2));
	const float hi = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fmaxf(tz1, tz2));
	// written so that a NaN anywhere means "visit"
	return !(lo > hi) & !(hi < 0.0f) & !(lo > best);
}

// The walk is wave-cooperative.  Rays of a wave need very different amounts
// of traversal (many miss every box, some cross the whole mesh), so a per-lane walk leaves
// most lanes idle.  Here every lane of the wave is a WORKER with its own depth-first stack;
// a worker without work takes (1) the next box its own ray hit, else (2) the OLDEST entry —
// the largest unexplored subtree — from the stack of a worker that has spare entries.  A
// worker that took foreign work loads that ray from LDS.  Hits are posted to the ray's slot in
// LDS with a 64-bit atomic min on (t0 bits, loop position), which is the reference's
// "first triangle wins ties" rule whatever the order of visits; the slot's t0 also prunes
// subtrees for every worker on that ray.  A worker's stack only ever holds nodes of ONE ray and
// one root-to-leaf path of siblings, so its depth is bounded by the tree depth.
// Threads per workgroup of the render kernels: one wave.  Nothing is shared between the waves of
// a workgroup, and single waves free their LDS and fill freed slots soonest.
#define WBRT_RENDER_BLOCK 64
#define WBRT_LDS_SPHERES 16 // sphere tables up to this size are copied to LDS (<= WBRT_RENDER_BLOCK)

struct StealLds
{
	float ray[7][WBRT_RENDER_BLOCK];              // o.xyz d.xyz, sphere result (upper bound for accepted t0)
	unsigned long long best[WBRT_RENDER_BLOCK];   // (t0 bits << 32) | loop position
	int2 mail[WBRT_RENDER_BLOCK];                 // hand-over: node, ray slot | box << 8
};

struct BvhRay
{
	float4 o;
	F3 d, inv;
	float rc, dlen, slop, e_limit, rho_pad;
	F3 o_near, o_far; // o +- rho_pad: capped-margin mode, one margin per ray and box
	float a;        // dot(d, d), as the sphere test computes it
	bool spheres;   // the hierarchy being walked is the one over the spheres
};

__device__ __forceinline__ void bvh_load_ray(const StealLds &w, int slot, BvhRay &r)
{
	r.o = make_float4(w.ray[0][slot], w.ray[1][slot], w.ray[2][slot], 0.0f);
	r.d = f3(w.ray[3][slot], w.ray[4][slot], w.ray[5][slot]);
	r.rc = w.ray[6][slot];
	// the walk's own box tests tolerate the 1-ulp hardware reciprocal and square root (their
	// error is far inside the box margins); the reference's slab test does not use these
	r.inv = f3(__builtin_amdgcn_rcpf(r.d.x), __builtin_amdgcn_rcpf(r.d.y), __builtin_amdgcn_rcpf(r.d.z));
	r.a = dot(r.d, r.d);
	r.dlen = __builtin_amdgcn_sqrtf(r.a) * 1.00001f;
	r.spheres = false;
}

// Lower bound of |d . m| over a node's m-box (m = (e0 x e1)/(|e0||e1|) of the triangles below it, bytes b -> b/127 - 1,
// rounded outwards by the builder): |det| of the reference's test is at least that times |e0||e1|.  dk = d / 127.
__device__ __forceinline__ float mbox_lower(unsigned lo3, unsigned hi3, F3 d, F3 dk)
{
	const float ax = __builtin_fmaf((float)(lo3 & 255u), dk.x, -d.x), bx = __builtin_fmaf((float)(hi3 & 255u), dk.x, -d.x);
	const float ay = __builtin_fmaf((float)((lo3 >> 8) & 255u), dk.y, -d.y), by = __builtin_fmaf((float)((hi3 >> 8) & 255u), dk.y, -d.y);
	const float az = __builtin_fmaf((float)((lo3 >> 16) & 255u), dk.z, -d.z), bz = __builtin_fmaf((float)((hi3 >> 16) & 255u), dk.z, -d.z);
	const float mn = fminf(ax, bx) + fminf(ay, by) + fminf(az, bz);
	const float mx = fmaxf(ax, bx) + fmaxf(ay, by) + fmaxf(az, bz);
	return fmaxf(fmaxf(mn, -mx), 0.0f);
}

// Proven box margin of one child for one ray (triangle hierarchies, WBRT_MESH_BVH).  With u = 2^-24, s = o - p0,
// K = |d||e0||e1| and det the FLOAT determinant of gpu_kernel.cl:494, forward error analysis of gpu_kernel.cl:489-520
// (every rounding of the cross products, dot products, the reciprocal and the three final products, Cramer's identity
// s det = Nu e0 + Nv e1 - Nt d for the exact numerators) bounds the distance between the ray point o + t0 d of an
// ACCEPTED hit and the triangle by 26.5 u |s| K / |det|, plus a few u (|s| + |e0| + |e1|) that the slop covers.
// Two bounds of K / |det| hold for every triangle below the child:
//   |det| >= 1e-5 (the test's own cutoff)                        ->  K/|det| <= 1e5 |d| E,  E = max |e0||e1|;
//   |det| >= |e0||e1| (L - 6u|d|),  L = mbox_lower(child, d)     ->  K/|det| <= 1.02 |d| / L  once L > 2e-5 |d|.
// |s| <= distance from o to the far corner of the child's box.  Constants: 26.5 u x 1.25 = 2.0e-6 (the factor covers
// the second-order terms and the roundings of this function itself).
__device__ __forceinline__ float far_corner(float ox, float oy, float oz, float mnx, float mny, float mnz, float mxx, float mxy, float mxz)
{
	const float fx = fmaxf(fabsf(mnx - ox), fabsf(mxx - ox)), fy = fmaxf(fabsf(mny - oy), fabsf(mxy - oy)), fz = fmaxf(fabsf(mnz - oz), fabsf(mxz - oz));
	return __builtin_amdgcn_sqrtf(__builtin_fmaf(fx, fx, __builtin_fmaf(fy, fy, fz * fz))) * 1.00001f;
}
// the det bound's factor of one child, or +inf when its m-box gives none
__device__ __forceinline__ float det_factor(unsigned lo3, unsigned hi3, F3 d, F3 dk, float dlen)
{
	const float L = mbox_lower(lo3, hi3, d, dk) - 2e-6f * dlen;
	return L > 2e-5f * dlen ? 2.05e-6f * __builtin_amdgcn_rcpf(L) : INFINITY;
}

// margins of ray r against the hierarchy of `box`; returns its root reference (INT_MIN: no triangles)
#define BVH_NONE ((int)0x80000000)
#define BVH_BOX_TAG 0x40000000
__device__ __forceinline__ int bvh_enter_box(const SceneDev &sc, int box, BvhRay &r)
{
	const float4 r0 = sc.bvh_roots[2 * box], r1 = sc.bvh_roots[2 * box + 1];
	const float far = fmaxf(fabsf(r.o.x - r0.x), fabsf(r.o.x - r1.x)) + fmaxf(fabsf(r.o.y - r0.y), fabsf(r.o.y - r1.y)) +
	                  fmaxf(fabsf(r.o.z - r0.z), fabsf(r.o.z - r1.z));
	const float diag = (r1.x - r0.x) + (r1.y - r0.y) + (r1.z - r0.z);
	const float R = (far + diag) * 1.000001f;
	r.slop = 1e-6f * (R + fabsf(r.o.x) + fabsf(r.o.y) + fabsf(r.o.z));
	r.spheres = sc.sph_tree && box == sc.B;
	if (r.spheres)
	{
		// box margins of this hierarchy are computed per child from the child's own box
		r.rho_pad = 0.0f;
		r.e_limit = INFINITY;
		return __float_as_int(r0.w);
	}
	r.e_limit = 5.99f * __builtin_amdgcn_rcpf(r.dlen);
	r.rho_pad = sc.bvh_rho * R + r.slop;
	r.o_near = f3(r.o.x + r.rho_pad, r.o.y + r.rho_pad, r.o.z + r.rho_pad);
	r.o_far = f3(r.o.x - r.rho_pad, r.o.y - r.rho_pad, r.o.z - r.rho_pad);
	return r1.w < 0.0f ? BVH_NONE : __float_as_int(r0.w);
}

template <bool BOXTREE>
__device__ __forceinline__ void triangles_steal(const SceneDev &sc, int (*stack)[WBRT_RENDER_BLOCK], StealLds &w, bool trace, bool spheres,
                                                F3 o, F3 d, float &closest, int &id, int &rounds)
{
	const float tmin = 0.001f;
	const int col = threadIdx.x, base = threadIdx.x & ~63, lane = threadIdx.x & 63;
	w.ray[0][col] = o.x; w.ray[1][col] = o.y; w.ray[2][col] = o.z;
	w.ray[3][col] = d.x; w.ray[4][col] = d.y; w.ray[5][col] = d.z;
	w.ray[6][col] = closest;
	w.best[col] = ((unsigned long long)__float_as_uint(closest) << 32) | 0xFFFFFFFFull;
	__builtin_amdgcn_wave_barrier();
	BvhRay r;
	bvh_load_ray(w, col, r);
	const F3 inv_d = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); // gpu_kernel.cl:536: the reference's invDirection
	const unsigned long long below = (1ull << lane) - 1ull;
	// Boxes whose slab test passes for the lane's own ray are queued on the lane's stack as
	// BVH_BOX_TAG | box; an entry turns into that box's root when a worker takes it.  A stack
	// has room for bvh_box_slots of them on top of one root-to-leaf path; a lane that hit more
	// boxes queues the rest in a further pass.
	int start = 0; // first box this lane has not queued yet
	int node = BVH_NONE, sp = 0, bot = 0, cur = col, cur_box = 0;
	// the sphere hierarchy (if the lane's ray walks it) is queued first: it is the largest job,
	// and thieves take the oldest entry
	if (spheres) stack[sp++][col] = BVH_BOX_TAG | sc.B;
	for (;;)
	{
		int resume = sc.B;
		if (BOXTREE)
		{
			// Many boxes: the lane's ray walks the hierarchy over the boxes (left child first, so leaves
			// come in increasing rank; `start` and `resume` count ranks here).  A node is skipped only when
			// the ray misses its box widened by far more than the rounding of the reference's slab test
			// (relative 2e-7 per plane distance); at a leaf that test itself decides.  The walk's own
			// stack is the top of the lane's column, above the queued boxes.
			const float pad = 2e-6f * (fabsf(o.x) + fabsf(o.y) + fabsf(o.z));
			const F3 rel = f3(2e-6f, 2e-6f, 2e-6f);
			int tnode = trace ? 0 : BVH_NONE, tp = WBRT_BVH_STACK;
			while (__ballot(tnode != BVH_NONE))
			{
				if (tnode == BVH_NONE) continue;
				if (tnode >= 0)
				{
					const float4 q0 = sc.top_nodes[4 * (size_t)tnode], q1 = sc.top_nodes[4 * (size_t)tnode + 1];
					const float4 q2 = sc.top_nodes[4 * (size_t)tnode + 2], q3 = sc.top_nodes[4 * (size_t)tnode + 3];
					// margin of a child: pad + 2e-6 * (largest coordinate magnitude of its box)
					const float ml = pad + rel.x * (fmaxf(fabsf(q0.x), fabsf(q0.w)) + fmaxf(fabsf(q0.y), fabsf(q1.x)) + fmaxf(fabsf(q0.z), fabsf(q1.y)));
					const float mr = pad + rel.x * (fmaxf(fabsf(q1.z), fabsf(q2.y)) + fmaxf(fabsf(q1.w), fabsf(q2.z)) + fmaxf(fabsf(q2.x), fabsf(q2.w)));
					float unused;
					const bool vl = bvh_box(f3(o.x + ml, o.y + ml, o.z + ml), f3(o.x - ml, o.y - ml, o.z - ml), r.inv,
					                        q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, INFINITY, unused);
					const bool vr = bvh_box(f3(o.x + mr, o.y + mr, o.z + mr), f3(o.x - mr, o.y - mr, o.z - mr), r.inv,
					                        q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, INFINITY, unused);
					const int left = __float_as_int(q3.x), right = __float_as_int(q3.y);
					if (vl & vr) stack[--tp][col] = right;
					tnode = vl ? left : vr ? right : BVH_NONE;
				}
				else
				{
					const int rank = ~tnode;
					tnode = BVH_NONE;
					if (rank >= start)
					{
						const int b = sc.top_box[rank];
						const float4 mn = sc.box_min[b], mx = sc.box_max[b];
						float tx1 = (mn.x - o.x) * inv_d.x, tx2 = (mx.x - o.x) * inv_d.x;
						float lo = fminf(tx1, tx2), hi = fmaxf(tx1, tx2);
						float ty1 = (mn.y - o.y) * inv_d.y, ty2 = (mx.y - o.y) * inv_d.y;
						lo = fmaxf(lo, fminf(ty1, ty2));
						hi = fminf(hi, fmaxf(ty1, ty2));
						float tz1 = (mn.z - o.z) * inv_d.z, tz2 = (mx.z - o.z) * inv_d.z;
						lo = fmaxf(lo, fminf(tz1, tz2));
						hi = fminf(hi, fmaxf(tz1, tz2));
						if (hi > fmaxf(lo, 0.0f))
						{
							if (sp < sc.bvh_box_slots) stack[sp++][col] = BVH_BOX_TAG | b;
							else { resume = rank; tp = WBRT_BVH_STACK; } // the rest in a further pass
						}
					}
				}
				if (tnode == BVH_NONE && tp < WBRT_BVH_STACK) tnode = stack[tp++][col];
			}
		}
		else
		for (int b = 0; b < sc.B; ++b)
		{
			const float4 mn = uniform_ld(sc.box_min, b), mx = uniform_ld(sc.box_max, b);
			float tx1 = (mn.x - o.x) * inv_d.x, tx2 = (mx.x - o.x) * inv_d.x;
			float lo = fminf(tx1, tx2), hi = fmaxf(tx1, tx2);
			float ty1 = (mn.y - o.y) * inv_d.y, ty2 = (mx.y - o.y) * inv_d.y;
			lo = fmaxf(lo, fminf(ty1, ty2));
			hi = fminf(hi, fmaxf(ty1, ty2));
			float tz1 = (mn.z - o.z) * inv_d.z, tz2 = (mx.z - o.z) * inv_d.z;
			lo = fmaxf(lo, fminf(tz1, tz2));
			hi = fminf(hi, fmaxf(tz1, tz2));
			if (trace && b >= start && hi > fmaxf(lo, 0.0f))
			{
				if (sp < sc.bvh_box_slots) stack[sp++][col] = BVH_BOX_TAG | b;
				else if (resume == sc.B) resume = b;
			}
		}
		if (!__ballot(sp > 0)) break;

		for (;;)
		{
			if (node == BVH_NONE && sp > bot)
			{
				node = stack[--sp][col];
				if (sp == bot) sp = bot = 0;
				if (node >= BVH_BOX_TAG) // own entries are queued and taken with the own ray loaded
				{
					cur_box = node - BVH_BOX_TAG;
					node = bvh_enter_box(sc, cur_box, r);
				}
			}
			const unsigned long long idle = __ballot(node == BVH_NONE);
			const unsigned long long donors = __ballot(sp > bot);
			if (idle == ~0ull)
			{
				if (!donors) break;
				continue; // a box without triangles was skipped; its lane takes its next entry
			}
			if (donors && __popcll(idle) >= sc.bvh_steal_min_idle)
			{
				const int nd = __popcll(donors), ni = __popcll(idle);
				const int m = nd < ni ? nd : ni;
				if (((donors >> lane) & 1ull) != 0)
				{
					const int k = __popcll(donors & below);
					if (k < m)
					{
						w.mail[base + k] = make_int2(stack[bot][col], (cur - base) | (cur_box << 8));
						++bot;
						if (sp == bot) sp = bot = 0;
					}
				}
				__builtin_amdgcn_wave_barrier();
				if (((idle >> lane) & 1ull) != 0)
				{
					const int k = __popcll(idle & below);
					if (k < m)
					{
						const int2 got = w.mail[base + k];
						node = got.x;
						cur = base + (got.y & 255);
						cur_box = got.y >> 8;
						bvh_load_ray(w, cur, r);
						if (node >= BVH_BOX_TAG)
						{
							cur_box = node - BVH_BOX_TAG;
							node = bvh_enter_box(sc, cur_box, r);
						}
						else (void)bvh_enter_box(sc, cur_box, r);
					}
				}
				__builtin_amdgcn_wave_barrier();
			}
			++rounds;
			if (node == BVH_NONE) continue;
			if (node >= 0)
			{
				const float4 q0 = sc.bvh_nodes[4 * (size_t)node], q1 = sc.bvh_nodes[4 * (size_t)node + 1];
				const float4 q2 = sc.bvh_nodes[4 * (size_t)node + 2], q3 = sc.bvh_nodes[4 * (size_t)node + 3];
				const float best = __uint_as_float((unsigned)(w.best[cur] >> 32));
				float lo_l, lo_r;
				bool vl, vr;
				if (!r.spheres && sc.bvh_rho > 0.0f)
				{
					// capped margins: one number per ray and box (never below min(strict margin, cap))
					vl = bvh_box(r.o_near, r.o_far, r.inv, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, best, lo_l);
					vr = bvh_box(r.o_near, r.o_far, r.inv, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, best, lo_r);
				}
				else
				{
					float pad_l, pad_r;
					if (r.spheres)
					{
						// 2e-3 * (|o - center| + radius) for every sphere below the child: the 1-norm distance
						// from the origin to the far corner of the child's box bounds that sum
						const float far_l = fmaxf(fabsf(q0.x - r.o.x), fabsf(q0.w - r.o.x)) + fmaxf(fabsf(q0.y - r.o.y), fabsf(q1.x - r.o.y)) +
						                    fmaxf(fabsf(q0.z - r.o.z), fabsf(q1.y - r.o.z));
						const float far_r = fmaxf(fabsf(q1.z - r.o.x), fabsf(q2.y - r.o.x)) + fmaxf(fabsf(q1.w - r.o.y), fabsf(q2.z - r.o.y)) +
						                    fmaxf(fabsf(q2.x - r.o.z), fabsf(q2.w - r.o.z));
						pad_l = 2.001e-3f * far_l + r.slop;
						pad_r = 2.001e-3f * far_r + r.slop;
					}
					else
					{
						const float far_l = far_corner(r.o.x, r.o.y, r.o.z, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y);
						const float far_r = far_corner(r.o.x, r.o.y, r.o.z, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w);
						float f_l = q3.z <= r.e_limit ? 0.20f * q3.z : INFINITY, f_r = q3.w <= r.e_limit ? 0.20f * q3.w : INFINITY;
						// the determinant bound is worth its arithmetic only where the cutoff bound is not already tiny
						if (__ballot(fmaxf(f_l, f_r) * r.dlen > 1e-4f))
						{
							const uint4 mq = sc.bvh_mbox[node];
							const F3 dk = r.d * (1.0f / 127.0f);
							f_l = fminf(f_l, det_factor(mq.x, mq.y, r.d, dk, r.dlen));
							f_r = fminf(f_r, det_factor(mq.z, mq.w, r.d, dk, r.dlen));
						}
						pad_l = __builtin_fmaf(f_l, far_l * r.dlen, __builtin_fmaf(3e-6f, far_l, r.slop));
						pad_r = __builtin_fmaf(f_r, far_r * r.dlen, __builtin_fmaf(3e-6f, far_r, r.slop));
					}
					vl = bvh_box(f3(r.o.x + pad_l, r.o.y + pad_l, r.o.z + pad_l), f3(r.o.x - pad_l, r.o.y - pad_l, r.o.z - pad_l), r.inv,
					             q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, best, lo_l);
					vr = bvh_box(f3(r.o.x + pad_r, r.o.y + pad_r, r.o.z + pad_r), f3(r.o.x - pad_r, r.o.y - pad_r, r.o.z - pad_r), r.inv,
					             q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, best, lo_r);
				}
				const int left = __float_as_int(q3.x), right = __float_as_int(q3.y);
				const bool left_first = !(lo_r < lo_l);
				if (vl & vr) stack[sp++][col] = left_first ? right : left;
				node = (vl & vr) ? (left_first ? left : right) : vl ? left : vr ? right : BVH_NONE;
			}
			else
			{
				// one triangle of the leaf per round
				const int leaf = ~node;
				const int first = leaf >> 4, more = leaf & 15;
				node = more ? ~(((first + 1) << 4) | (more - 1)) : BVH_NONE;
				const float4 g0 = sc.bvh_tris[3 * (size_t)first];
				const float4 g2 = sc.bvh_tris[3 * (size_t)first + 2];
				const F3 ro = f3(r.o.x, r.o.y, r.o.z);
				if (r.spheres)
				{
					// gpu_kernel.cl:455-478 with tMax = +inf: the nearest root at or beyond tMin; the
					// atomic min on (root, S - 1 - index) then keeps what the sequential loop with its
					// shrinking, inclusive tMax keeps (the later sphere on a tie)
					F3 oc = ro - f3(g0.x, g0.y, g0.z);
					float hb = dot(oc, r.d);
					float cc = dot(oc, oc) - g0.w;
					float disc = hb * hb - r.a * cc;
					if (disc < 0) continue;
					float sq = sqrtf(disc);
					float root = (-hb - sq) / r.a;
					if (root < tmin)
					{
						root = (-hb + sq) / r.a;
						if (root < tmin) continue;
					}
					atomicMin(&w.best[cur], ((unsigned long long)__float_as_uint(root) << 32) | (unsigned)(sc.S - 1 - __float_as_int(g2.y)));
					continue;
				}
				const float4 g1 = sc.bvh_tris[3 * (size_t)first + 1];
				const F3 p0 = f3(g0.x, g0.y, g0.z), e0 = f3(g0.w, g1.x, g1.y), e1 = f3(g1.z, g1.w, g2.x);
				F3 h = cross(r.d, e1);
				float det = dot(e0, h);
				if (det > -0.00001f && det < 0.00001f) continue;
				float f = 1.0f / det;
				F3 sv = ro - p0;
				float u = f * dot(sv, h);
				if (u < 0.0f || u > 1.0f) continue;
				F3 q = cross(sv, e0);
				float v = f * dot(r.d, q);
				if (v < 0.0f || u + v > 1.0f) continue;
				float t0 = f * dot(e1, q);
				if (t0 > tmin && t0 < r.rc)
					atomicMin(&w.best[cur], ((unsigned long long)__float_as_uint(t0) << 32) | (unsigned)(sc.S + __float_as_int(g2.y)));
			}
		}
		// every stack is empty again; lanes that could not queue all their boxes go round once more
		if (cur != col) { cur = col; bvh_load_ray(w, col, r); }
		start = resume;
		__builtin_amdgcn_wave_barrier();
		if (!__ballot(resume < sc.B)) break;
	}
	__builtin_amdgcn_wave_barrier();
	const unsigned long long k = w.best[col];
	if (trace && (unsigned)k != 0xFFFFFFFFu)
	{
		// low word: S - 1 - sphere index, or S + position of the triangle in the grouped order
		const int low = (int)(unsigned)k;
		closest = __uint_as_float((unsigned)(k >> 32));
		id = low < sc.S ? sc.S - 1 - low : low;
	}
}

// The counting build always takes the exact sphere loop: its counters are defined by the
// reference's sequential order of decisions.
template <bool STATS>
__device__ __forceinline__ void closest_sphere(const SceneDev &sc, F3 o, F3 d, float &closest, int &id, Counters<STATS> &ctr)
{
	closest = INFINITY;
	id = -1;
	const float a = dot(d, d);
	if (STATS)
	{
		ctr.add(C_SPHERE_TESTS, (unsigned)sc.S);
		spheres_exact<STATS>(sc, o, d, a, closest, id, ctr);
	}
	else if (!spheres_fast(sc, o, d, a, closest, id))
		spheres_exact<STATS>(sc, o, d, a, closest, id, ctr);
}

// gpu_kernel.cl:182-196
template <bool STATS>
__device__ __forceinline__ F3 rand_in_unit_sphere(Rng &rng, Counters<STATS> &ctr)
{
	for (;;)
	{
		if (STATS) { ctr.add(C_SPHERE_TRIES, 1); ctr.add(C_RNG_DRAWS, 3); }
		float x = rng.bipolar();
		float y = rng.bipolar();
		float z = rng.bipolar();
		F3 p = f3(x, y, z);
		if (dot(p, p) >= 1.0f) continue;
		return p;
	}
}

// gpu_kernel.cl:209-212
__device__ __forceinline__ F3 reflect(F3 v, F3 n) { return v - n * (2.0f * dot(v, n)); }

// Pixel-stream state of one lane.
struct Lane
{
	F3 o, d;        // current ray (direction not normalised, gpu_kernel.cl:575)
	F3 thr;         // throughput
	F3 sum;         // clamped per-sample sum
	Rng rng;
	float base_u, base_v; // in-block column/row + block origin (gpu_kernel.cl:626-627)
	int depth, sample;
};

// gpu_kernel.cl:626-629 + :559-578: next camera ray of this lane's stream
template <bool STATS>
__device__ __forceinline__ void camera_ray(const RenderParams &p, Lane &L, Counters<STATS> &ctr)
{
	if (STATS) { ctr.add(C_PATHS, 1); ctr.add(C_RNG_DRAWS, 2); }
	// base + draw: the draw is top*2^-32 exactly, so the fused form rounds once, identically
	const float su = __builtin_fmaf(L.rng.next_top(), 2.3283064365386963e-10f, L.base_u);
	const float sv = __builtin_fmaf(L.rng.next_top(), 2.3283064365386963e-10f, L.base_v);
	const float u = div_by_const(su, (float)p.width, p.inv_width);
	const float v = div_by_const(sv, (float)p.height, p.inv_height);
	F3 org = ld3(p.org);
	if (!(p.aperture <= 0))
	{
		float dx, dy;
		for (;;) // gpu_kernel.cl:243-257
		{
			if (STATS) { ctr.add(C_DISK_TRIES, 1); ctr.add(C_RNG_DRAWS, 2); }
			dx = L.rng.bipolar();
			dy = L.rng.bipolar();
			if (dx * dx + dy * dy + 0.0f * 0.0f >= 1.0f) continue;
			break;
		}
		org = org + (ld3(p.ddu) * dx + ld3(p.ddv) * dy);
	}
	L.d = (ld3(p.llc) + (ld3(p.hor) * u + ld3(p.ver) * v)) - org;
	L.o = org;
	L.thr = f3(1.0f, 1.0f, 1.0f);
	L.depth = 0;
}

// Maps a launch-wide work index to (block, in-block id).  Inside a block, lanes of a wave
// cover an 8x8 pixel tile when the block size allows it (neighbouring pixels follow similar
// paths); any bijection is valid because seeds and outputs are addressed by the in-block id.
__device__ __forceinline__ bool locate(const RenderParams &p, long long w, int &gid, int &bx, int &by)
{
	const int block_px = p.bsx * p.bsy;
	const long long total = (long long)block_px * p.nbx * p.nby;
	if (w >= total) return false;
	int blk = (int)(w / block_px);
	int local = (int)(w - (long long)blk * block_px);
	bx = blk % p.nbx;
	by = blk / p.nbx;
	if (((p.bsx | p.bsy) & 7) == 0)
	{
		int tile = local >> 6, l = local & 63;
		int tpr = p.bsx >> 3;
		int px = (tile % tpr) * 8 + (l & 7);
		int py = (tile / tpr) * 8 + (l >> 3);
		gid = py * p.bsx + px;
	}
	else gid = local;
	return true;
}

__device__ __forceinline__ void init_lane(const RenderParams &p, Lane &L, int gid, int bx, int by)
{
	L.rng.seed(p.seeds[gid]);
	L.base_u = (float)(gid % p.bsx) + ((float)p.bsx * (float)(bx + p.off_bx));
	L.base_v = (float)(gid / p.bsx) + ((float)p.bsy * (float)(by + p.off_by));
	L.sum = f3(0.0f, 0.0f, 0.0f);
	L.sample = 0;
}

// MESH selects the loop shape.  Without bounding boxes a lane simply leaves the loop when its
// pixel stream is finished.  With boxes, finished lanes stay in the loop as helpers of the
// cooperative triangle phase (which runs in wave-uniform control flow) until the whole wave is
// done: what bounds a mesh frame is its longest pixel stream, and a lone live lane still gets
// all 64 lanes of triangle throughput this way.
// BOXTREE: the scene comes with a hierarchy over its boxes (sc.top_nodes); a build of its own, so that scenes
// with few boxes keep the registers of the plain loop over the boxes.
template <bool STATS, bool MESH, bool BOXTREE = false>
__global__ __launch_bounds__(WBRT_RENDER_BLOCK, (MESH && !STATS) ? 4 : 1) void k_render(const RenderParams p)
{
	__shared__ TriTiles s_tiles[MESH && STATS ? WBRT_RENDER_BLOCK / 64 : 1]; // counting build: sequential staged loop, one per wave
	__shared__ unsigned long long s_best[MESH && !STATS ? WBRT_RENDER_BLOCK / 64 : 1][64];
	__shared__ int s_stack[MESH && !STATS ? WBRT_BVH_STACK : 1][WBRT_RENDER_BLOCK]; // one column per thread
	__shared__ StealLds s_steal[1];
	// the winner's centre, radius and material are read with a per-lane index once per segment:
	// for small sphere counts from this per-wave copy (one LDS round trip instead of four
	// vector loads on the wave's critical path)
	__shared__ float4 s_sph[WBRT_LDS_SPHERES][3]; // center.xyz r^2 | albedo fuzz | radius ior type -
	if (p.sc.S <= WBRT_LDS_SPHERES && (int)threadIdx.x < p.sc.S)
	{
		const float2 b = p.sc.sph_mat_b[threadIdx.x];
		s_sph[threadIdx.x][0] = p.sc.sph_geom[threadIdx.x];
		s_sph[threadIdx.x][1] = p.sc.sph_mat_a[threadIdx.x];
		s_sph[threadIdx.x][2] = make_float4(p.sc.sph_radius[threadIdx.x], b.x, b.y, 0.0f);
	}
	__builtin_amdgcn_wave_barrier();
	Counters<STATS> ctr;
	const SceneDev &sc = p.sc;
	long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	int gid = 0, bx = 0, by = 0;
	long long work = w;
	if (p.order)
	{
		// Longest-first schedule: waves (64 consecutive work indices, an 8x8 pixel tile when the
		// block size allows) arrive sorted by estimated cost (descending), so the costliest
		// pixel streams start first; their waves also get a higher issue priority,
		// because the frame cannot finish before its longest stream does.
		if (w < (long long)p.bsx * p.bsy * p.nbx * p.nby) work = (long long)p.order[w >> 6] * 64 + (w & 63);
		const unsigned pos = blockIdx.x, n = gridDim.x;
		if (pos * 50u < n) __builtin_amdgcn_s_setprio(3);
		else if (pos * 10u < n) __builtin_amdgcn_s_setprio(2);
		else if (pos * 3u < n) __builtin_amdgcn_s_setprio(1);
	}
	bool alive = locate(p, work, gid, bx, by);
	Lane L;
	L.o = L.d = L.thr = L.sum = f3(0, 0, 0);
	L.rng.lo = L.rng.hi = 0; L.base_u = L.base_v = 0; L.depth = 0; L.sample = 0;
	if (alive)
	{
		init_lane(p, L, gid, bx, by);
		alive = p.spp > 0;
		if (alive) camera_ray<STATS>(p, L, ctr);
	}

	int iterations = 0, walk_cost = 0;
	for (;;)
	{
		if (MESH ? !__ballot(alive) : !alive) break;
		// 0 = path continues, 1 = ended black, 2 = ended on the sky, 3 = ended with `col`
		int outcome = 0;
		F3 col = f3(0.0f, 0.0f, 0.0f);
		bool trace = false, walk_spheres = false;
		float t = INFINITY;
		int id = -1;
		if (!MESH || alive)
		{
			++iterations;
			if (L.depth >= p.max_depth) // gpu_kernel.cl:276,337-340
			{
				outcome = (L.depth == p.max_depth) ? 1 : 2;
				if (STATS && outcome == 1) ctr.add(C_DEPTH_EXHAUSTED, 1);
			}
			else
			{
				trace = true;
				if (STATS) ctr.add(C_SEGMENTS, 1);
				// Many spheres: the ray walks the sphere hierarchy together with the triangle
				// hierarchies below, unless its numbers are outside the range the margins of
				// that walk were derived for (then the loop over all spheres runs here).
				if (MESH && !STATS && sc.sph_tree && sc.bvh_nodes)
				{
					const float a = dot(L.d, L.d);
					walk_spheres = a > 1e-20f && a < 1e4f && fabsf(L.o.x) < 1e8f && fabsf(L.o.y) < 1e8f && fabsf(L.o.z) < 1e8f;
				}
				if (!walk_spheres) closest_sphere<STATS>(sc, L.o, L.d, t, id, ctr);
			}
		}
		if (MESH)
		{
			if (STATS)
			{
				if (trace) boxes_and_triangles<STATS>(sc, s_tiles[threadIdx.x >> 6], L.o, L.d, t, id, ctr);
			}
			else if (sc.bvh_nodes)
			{
				// pilot launch: the rounds of the walk (the same number in every lane) join the tile's
				// cost estimate
				int rounds = 0;
				triangles_steal<BOXTREE>(sc, s_stack, s_steal[0], trace, walk_spheres, L.o, L.d, t, id, rounds);
				if (p.costs || p.record) walk_cost += rounds;
			}
			else triangles_cooperative(sc, s_best[threadIdx.x >> 6], trace, L.o, L.d, t, id);
		}
		if (MESH && !alive) continue;
		if (trace)
		{
			if (id < 0) outcome = 2;
			else
			{
				// winner's surface (gpu_kernel.cl:480-484, :522-526, :352-356)
				F3 hp = L.o + L.d * t;
				F3 outward;
				float4 ma;
				float2 mb;
				if (id < sc.S)
				{
					if (sc.S <= WBRT_LDS_SPHERES) // small sphere tables live in LDS (filled at kernel start)
					{
						const float4 g = s_sph[id][0], m = s_sph[id][1], q = s_sph[id][2];
						outward = div3(hp - f3(g.x, g.y, g.z), q.x);
						ma = m;
						mb = make_float2(q.y, q.z);
					}
					else
					{
						const float4 g = sc.sph_geom[id];
						outward = div3(hp - f3(g.x, g.y, g.z), sc.sph_radius[id]);
						ma = sc.sph_mat_a[id];
						mb = sc.sph_mat_b[id];
					}
				}
				else
				{
					const int j = id - sc.S;
					const float4 g0 = sc.tri_geom[3 * j + 0];
					const float4 g1 = sc.tri_geom[3 * j + 1];
					const float4 g2 = sc.tri_geom[3 * j + 2];
					outward = unit_fast(cross(f3(g0.w, g1.x, g1.y), f3(g1.z, g1.w, g2.x)));
					ma = sc.tri_mat_a[j];
					mb = sc.tri_mat_b[j];
				}
				const bool front = dot(L.d, outward) < 0;
				const F3 n = front ? outward : outward * -1.0f;
				const F3 albedo = f3(ma.x, ma.y, ma.z);
				const float fuzz = ma.w, ior = mb.x;
				const int type = __float_as_int(mb.y);

				// Steps shared by the material types, so lanes with different materials run
				// them together: (1) a point in the unit sphere for Lambertian and fuzzy-metal
				// lanes (same draws, same order as gpu_kernel.cl:400 / :419 — reflecting first
				// consumes no draws); (2) one normalisation: of that point (Lambertian) or of
				// the incoming direction (metal, dielectric).
				F3 rv = f3(0.0f, 0.0f, 0.0f);
				if (type == 0 || (type == 1 && fuzz > 0.0f)) rv = rand_in_unit_sphere<STATS>(L.rng, ctr);
				F3 un = f3(0.0f, 0.0f, 0.0f);
				if (type == 0 || type == 1 || type == 2) un = unit_fast(type == 0 ? rv : L.d);

				bool scattered = true;
				F3 nd = L.d;
				F3 att = f3(1.0f, 1.0f, 1.0f);
				if (type == 0) // gpu_kernel.cl:398-413
				{
					if (STATS) ctr.add(C_LAMBERT, 1);
					nd = n + un;
					const float eps = 1e-8f;
					if (fabsf(nd.x) < eps && fabsf(nd.y) < eps && fabsf(nd.z) < eps) nd = n;
					att = albedo;
				}
				else if (type == 1) // gpu_kernel.cl:415-423
				{
					if (STATS) ctr.add(C_METAL, 1);
					nd = reflect(un, n);
					if (fuzz > 0.0f)
					{
						if (STATS) ctr.add(C_METAL_FUZZ, 1);
						nd = nd + rv * fuzz;
					}
					att = albedo;
					if (!(dot(nd, n) > 0))
					{
						if (STATS) ctr.add(C_ABSORBED, 1);
						scattered = false;
						outcome = 1;
					}
				}
				else if (type == 2) // gpu_kernel.cl:425-451, :214-227
				{
					if (STATS) ctr.add(C_DIELECTRIC, 1);
					const float ratio = front ? (1.0f / ior) : ior;
					const F3 ud = un;
					const float cos_t = fminf(dot(ud * -1.0f, n), 1.0f);
					const float sin_t = sqrtf(1.0f - cos_t * cos_t);
					bool refl = ratio * sin_t > 1.0f;
					if (!refl)
					{
						if (STATS) { ctr.add(C_DIELECTRIC_DRAW, 1); ctr.add(C_RNG_DRAWS, 1); }
						float r0 = (1.0f - ratio) / (1.0f + ratio);
						r0 = r0 * r0;
						const float fr = r0 + (1.0f - r0) * pt_pow5f(1.0f - cos_t);
						refl = fr > L.rng.next();
					}
					if (refl) nd = reflect(ud, n);
					else
					{
						const float c2 = fminf(dot(ud * -1.0f, n), 1.0f);
						const F3 perp = ud * ratio + n * (c2 * ratio);
						const F3 par = n * (-sqrtf(fabsf(1.0f - dot(perp, perp))));
						nd = perp + par;
					}
				}
				else if (type == 3) // gpu_kernel.cl:326-329
				{
					if (STATS) ctr.add(C_EMISSIVE, 1);
					scattered = false;
					outcome = 3;
					col = L.thr * albedo;
				}
				else
				{
					// unknown material: depth advances, same ray again (gpu_kernel.cl:331-332)
					scattered = false;
					L.depth++;
				}
				if (scattered)
				{
					L.o = hp;
					L.d = nd;
					L.thr = L.thr * att;
					L.depth++;
				}
			}
		}

		if (outcome == 2)
		{
			if (STATS) ctr.add(C_MISS, 1);
			int texel = 0;
			const bool decided = !STATS && sky_texel_fast(sc, L.d, texel);
			F3 sky = ld3(sc.sky + 3 * (size_t)(decided ? texel : 0));
			if (!decided) sky = sky_lookup(sc, L.d); // exact texel coordinates (rare)
			col = L.thr * sky;
		}
		if (outcome != 0)
		{
			// gpu_kernel.cl:632-644: clamp, NaN -> running sum, accumulate
			// (the median of {c, 0, 1} is the two-sided clamp for every number; a NaN, which passes both of
			// the reference's comparisons, is replaced afterwards; a -0 may become +0, which the sum, never
			// -0 itself, does not see)
			const F3 raw = col;
			col = f3(__builtin_amdgcn_fmed3f(raw.x, 0.0f, 1.0f), __builtin_amdgcn_fmed3f(raw.y, 0.0f, 1.0f), __builtin_amdgcn_fmed3f(raw.z, 0.0f, 1.0f));
			if (raw.x != raw.x) col.x = L.sum.x;
			if (raw.y != raw.y) col.y = L.sum.y;
			if (raw.z != raw.z) col.z = L.sum.z;
			L.sum = L.sum + col;
			L.sample++;
			if (L.sample < p.spp) camera_ray<STATS>(p, L, ctr);
			else alive = false;
		}
	}

	if (STATS)
	{
		ctr.flush(p.counters);
		return;
	}
	if (p.costs) // pilot launch: only the loop iterations (ray segments) of each work index
	{
		const long long total = (long long)p.bsx * p.bsy * p.nbx * p.nby;
		if (w < total) p.costs[w] = iterations;
		if (MESH && (threadIdx.x & 63) == 0 && w < total) p.costs[total + (w >> 6)] = walk_cost; // per tile, after the per-pixel part
		return;
	}
	if (w < (long long)p.bsx * p.bsy * p.nbx * p.nby)
	{
		if (p.record) // the next frame of the same view is scheduled from these counts instead of a pilot launch
		{
			const long long total = (long long)p.bsx * p.bsy * p.nbx * p.nby;
			const long long at = p.order ? (long long)p.order[w >> 6] * 64 + (w & 63) : w;
			p.record[at] = iterations;
			if (MESH && (threadIdx.x & 63) == 0) p.record[total + (at >> 6)] = walk_cost;
		}
		const int row = gid / p.bsx + p.bsy * by, colx = gid % p.bsx + p.bsx * bx;
		float *dst = p.out + 3 * ((size_t)row * p.out_w + colx);
		F3 r = L.sum;
		if (p.out_mode == 0) r = r / (float)p.spp; // gpu_kernel.cl:647
		if (p.out_mode == 2) { r.x += dst[0]; r.y += dst[1]; r.z += dst[2]; }
		dst[0] = r.x; dst[1] = r.y; dst[2] = r.z;
	}
}

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
	if (p.sc.top_nodes && p.sc.bvh_nodes) hipLaunchKernelGGL((k_render<false, true, true>), grid, block, 0, stream, p);
	else if (p.sc.B > 0 || (p.sc.sph_tree && p.sc.bvh_nodes)) hipLaunchKernelGGL((k_render<false, true>), grid, block, 0, stream, p);
	else hipLaunchKernelGGL((k_render<false, false>), grid, block, 0, stream, p);
	return hipGetLastError();
}

hipError_t wbrt_launch_count(const RenderParams &p, hipStream_t stream)
{
	const dim3 grid(grid_for(p, WBRT_RENDER_BLOCK)), block(WBRT_RENDER_BLOCK);
	if (p.sc.B > 0) hipLaunchKernelGGL((k_render<true, true>), grid, block, 0, stream, p);
	else hipLaunchKernelGGL((k_render<true, false>), grid, block, 0, stream, p);
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
