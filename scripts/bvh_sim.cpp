// Design tool (host only, not part of the product or the tests): walks the triangle hierarchy of a mesh on
// the CPU for a sample of camera and bounce rays and counts inner-node visits and triangle tests per ray under
// different box-margin rules, so that a rule can be priced before it goes into the kernel.
//   hipcc -O2 -std=c++17 -o /tmp/bvh_sim scripts/bvh_sim.cpp opencl-ray-tracer_amd/csrc/rt_bvh.cpp opencl-ray-tracer_amd/csrc/wb_host.cpp
//   /tmp/bvh_sim tris.f32 [width]      (tris.f32: T x 9 floats, transformed vertices)
#include "../include/ocl_rt_amd.h"
#include "../opencl-ray-tracer_amd/csrc/rt_device.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

struct V3 { float x, y, z; };
static V3 operator+(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
static V3 operator-(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static V3 operator*(V3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
static float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V3 cross(V3 a, V3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
static float len(V3 a) { return std::sqrt(dot(a, a)); }

struct Ray { V3 o, d; };
struct Stats { double nodes = 0, tris = 0, rays = 0, excess_e = 0, excess_det = 0, excess_big = 0; double by_depth[32] = {}; };

static std::vector<float4> g_geom;
static WbrtBvh g_bvh[2];

// reference triangle test (float, no contraction assumed irrelevant for counting)
static bool tri_test(const Ray &r, int slot_tri, float closest, float &t_out)
{
	const float4 g0 = g_geom[3 * (size_t)slot_tri], g1 = g_geom[3 * (size_t)slot_tri + 1], g2 = g_geom[3 * (size_t)slot_tri + 2];
	V3 p0 = { g0.x, g0.y, g0.z }, e0 = { g0.w, g1.x, g1.y }, e1 = { g1.z, g1.w, g2.x };
	V3 h = cross(r.d, e1);
	float det = dot(e0, h);
	if (det > -0.00001f && det < 0.00001f) return false;
	float f = 1.0f / det;
	V3 s = r.o - p0;
	float u = f * dot(s, h);
	if (u < 0.0f || u > 1.0f) return false;
	V3 q = cross(s, e0);
	float v = f * dot(r.d, q);
	if (v < 0.0f || u + v > 1.0f) return false;
	float t0 = f * dot(e1, q);
	if (t0 > 0.001f && t0 < closest) { t_out = t0; return true; }
	return false;
}

static bool box_hit(V3 o, V3 inv, const float *mn, const float *mx, float pad, float best, float &lo)
{
	float tx1 = (mn[0] - pad - o.x) * inv.x, tx2 = (mx[0] + pad - o.x) * inv.x;
	float ty1 = (mn[1] - pad - o.y) * inv.y, ty2 = (mx[1] + pad - o.y) * inv.y;
	float tz1 = (mn[2] - pad - o.z) * inv.z, tz2 = (mx[2] + pad - o.z) * inv.z;
	lo = std::fmax(std::fmax(std::fmin(tx1, tx2), std::fmin(ty1, ty2)), std::fmin(tz1, tz2));
	float hi = std::fmin(std::fmin(std::fmax(tx1, tx2), std::fmax(ty1, ty2)), std::fmax(tz1, tz2));
	return !(lo > hi) && !(hi < 0.0f) && !(lo > best);
}

static float far1(V3 o, const float *mn, const float *mx)
{
	return std::fmax(std::fabs(o.x - mn[0]), std::fabs(o.x - mx[0])) + std::fmax(std::fabs(o.y - mn[1]), std::fabs(o.y - mx[1])) +
	       std::fmax(std::fabs(o.z - mn[2]), std::fabs(o.z - mx[2]));
}
static float far2(V3 o, const float *mn, const float *mx)
{
	float a = std::fmax(std::fabs(o.x - mn[0]), std::fabs(o.x - mx[0])), b = std::fmax(std::fabs(o.y - mn[1]), std::fabs(o.y - mx[1])),
	      c = std::fmax(std::fabs(o.z - mn[2]), std::fabs(o.z - mx[2]));
	return std::sqrt(a * a + b * b + c * c);
}

// Per child of every node: a box around the vectors M = (e0 x e1) / (|e0||e1|) of the triangles below it, each flipped
// into the hemisphere of the subtree's mean (the test is two-sided).  |d . M| >= LB over the box bounds |det| / (|e0||e1|).
struct MBox { float lo[3], hi[3]; };
static std::vector<MBox> g_mbox[2]; // 2 per node: left, right

static void collect(const WbrtBvh &bvh, int ref, std::vector<int> &tris)
{
	if (ref >= 0) { collect(bvh, bvh.nodes[(size_t)ref].left, tris); collect(bvh, bvh.nodes[(size_t)ref].right, tris); }
	else if (ref != (int)0x80000000)
	{
		int leaf = ~ref, first = leaf >> 4, cnt = (leaf & 15) + 1;
		for (int k = 0; k < cnt; ++k) tris.push_back(bvh.tri_index[(size_t)first + k]);
	}
}

static MBox mbox_of(const std::vector<int> &tris)
{
	std::vector<V3> M;
	V3 big = { 0, 0, 0 };
	float bigl = -1;
	for (int t : tris)
	{
		const float4 g0 = g_geom[3 * (size_t)t], g1 = g_geom[3 * (size_t)t + 1], g2 = g_geom[3 * (size_t)t + 2];
		V3 e0 = { g0.w, g1.x, g1.y }, e1 = { g1.z, g1.w, g2.x };
		V3 n = cross(e0, e1);
		float k = len(e0) * len(e1);
		V3 m = k > 0 ? n * (1.0f / k) : V3{ 0, 0, 0 };
		if (len(n) > bigl) { bigl = len(n); big = m; }
		M.push_back(m);
	}
	V3 mean = { 0, 0, 0 };
	for (V3 &m : M) { if (dot(m, big) < 0) m = m * -1.0f; mean = mean + m; }
	MBox b;
	for (int a = 0; a < 3; ++a) { b.lo[a] = 1e30f; b.hi[a] = -1e30f; }
	for (V3 &m : M)
	{
		if (dot(m, mean) < 0) m = m * -1.0f;
		const float c[3] = { m.x, m.y, m.z };
		for (int a = 0; a < 3; ++a) { b.lo[a] = std::fmin(b.lo[a], c[a]); b.hi[a] = std::fmax(b.hi[a], c[a]); }
	}
	return b;
}

static void build_mboxes(int variant)
{
	const WbrtBvh &bvh = g_bvh[variant];
	g_mbox[variant].resize(bvh.nodes.size() * 2);
	for (size_t n = 0; n < bvh.nodes.size(); ++n)
	{
		std::vector<int> l, r;
		collect(bvh, bvh.nodes[n].left, l);
		collect(bvh, bvh.nodes[n].right, r);
		g_mbox[variant][2 * n] = mbox_of(l);
		g_mbox[variant][2 * n + 1] = mbox_of(r);
	}
}

// Per child: ratio[a] = max over triangles of E_tri sqrt(1 - n_a^2) / E_child (in-plane share of the margin along axis a),
// and the largest sliver factor sigma = |e0||e1| / |e0 x e1| (the out-of-plane part of the margin is 2.2e-6 sigma S).
struct Aniso { float ratio[3], sigma; };
static std::vector<Aniso> g_aniso; // variant 1 only, 2 per node
static Aniso aniso_of(const std::vector<int> &tris)
{
	Aniso a = { { 0, 0, 0 }, 0 };
	float emax = 0;
	std::vector<float> ea[3];
	for (int t : tris)
	{
		const float4 g0 = g_geom[3 * (size_t)t], g1 = g_geom[3 * (size_t)t + 1], g2 = g_geom[3 * (size_t)t + 2];
		V3 e0 = { g0.w, g1.x, g1.y }, e1 = { g1.z, g1.w, g2.x };
		V3 n = cross(e0, e1);
		float E = len(e0) * len(e1), A = len(n);
		emax = std::fmax(emax, E);
		float sig = A > 0 ? E / A : INFINITY;
		a.sigma = std::fmax(a.sigma, sig);
		const float nn[3] = { A > 0 ? n.x / A : 0, A > 0 ? n.y / A : 0, A > 0 ? n.z / A : 0 };
		for (int k = 0; k < 3; ++k) a.ratio[k] = std::fmax(a.ratio[k], E * std::sqrt(std::fmax(0.0f, 1 - nn[k] * nn[k])));
	}
	for (int k = 0; k < 3; ++k) a.ratio[k] = emax > 0 ? std::fmin(1.0f, a.ratio[k] / emax * 1.0001f) : 1.0f;
	return a;
}
static void build_aniso()
{
	const WbrtBvh &bvh = g_bvh[1];
	g_aniso.resize(bvh.nodes.size() * 2);
	for (size_t n = 0; n < bvh.nodes.size(); ++n)
	{
		std::vector<int> l, r;
		collect(bvh, bvh.nodes[n].left, l);
		collect(bvh, bvh.nodes[n].right, r);
		g_aniso[2 * n] = aniso_of(l);
		g_aniso[2 * n + 1] = aniso_of(r);
	}
}
static bool box_hit3(V3 o, V3 inv, const float *mn, const float *mx, const float *pad, float best, float &lo)
{
	float tx1 = (mn[0] - pad[0] - o.x) * inv.x, tx2 = (mx[0] + pad[0] - o.x) * inv.x;
	float ty1 = (mn[1] - pad[1] - o.y) * inv.y, ty2 = (mx[1] + pad[1] - o.y) * inv.y;
	float tz1 = (mn[2] - pad[2] - o.z) * inv.z, tz2 = (mx[2] + pad[2] - o.z) * inv.z;
	lo = std::fmax(std::fmax(std::fmin(tx1, tx2), std::fmin(ty1, ty2)), std::fmin(tz1, tz2));
	float hi = std::fmin(std::fmin(std::fmax(tx1, tx2), std::fmax(ty1, ty2)), std::fmax(tz1, tz2));
	return !(lo > hi) && !(hi < 0.0f) && !(lo > best);
}

static float mbox_lb(const MBox &b, V3 d)
{
	const float dd[3] = { d.x, d.y, d.z };
	float lo = 0, hi = 0;
	for (int a = 0; a < 3; ++a) { lo += std::fmin(dd[a] * b.lo[a], dd[a] * b.hi[a]); hi += std::fmax(dd[a] * b.lo[a], dd[a] * b.hi[a]); }
	return lo > 0 ? lo : hi < 0 ? -hi : 0.0f;
}

// mode 0: capped rho*R; 1: strict as shipped (R of the root); 2: strict with R per node (Euclidean far corner + node diagonal)
// and the distance clipped by the best hit so far
static int walk(const WbrtBvh &bvh, int variant, const Ray &r, int mode, float rho, float closest_in, float &t_best, Stats &st)
{
	const WbrtBvhRoot &root = bvh.roots[0];
	V3 inv = { 1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z };
	const float dlen = len(r.d) * 1.00001f;
	const float diag1 = (root.mx[0] - root.mn[0]) + (root.mx[1] - root.mn[1]) + (root.mx[2] - root.mn[2]);
	const float R = (far1(r.o, root.mn, root.mx) + diag1) * 1.000001f;
	const float slop = 1e-6f * (R + std::fabs(r.o.x) + std::fabs(r.o.y) + std::fabs(r.o.z));
	const float padc = 0.35f * R * dlen, e_limit = 5.99f / dlen;
	float best = closest_in;
	int best_id = -1;
	{
		// the reference enters the mesh only through its bounding box (max.y, max.z = FLT_MIN: main.cpp:114-115)
		const float bmn[3] = { root.mn[0], root.mn[1], root.mn[2] }, bmx[3] = { root.mx[0], std::fmax(root.mx[1], 1.17549435e-38f), std::fmax(root.mx[2], 1.17549435e-38f) };
		float lo_;
		if (!box_hit(r.o, inv, bmn, bmx, 0.0f, INFINITY, lo_)) { t_best = best; return -1; }
	}
	int stack[64], sp = 0;
	int node = root.ref;
	for (;;)
	{
		if (node >= 0)
		{
			st.nodes += 1;
			const WbrtBvhNode &n = bvh.nodes[(size_t)node];
			float pl, pr;
			if (mode == 0) pl = pr = rho * R + slop;
			else if (mode == 1)
			{
				pl = n.le <= e_limit ? n.le * padc + slop : INFINITY;
				pr = n.re <= e_limit ? n.re * padc + slop : INFINITY;
			}
			else
			{
				// |o - p0| + |e| per child: far corner of the child's box (Euclidean) + its diagonal; the hit distance part
				// of the bound is limited by the best hit so far
				auto rnode = [&](const float *mn, const float *mx) {
					float dg = std::sqrt((mx[0] - mn[0]) * (mx[0] - mn[0]) + (mx[1] - mn[1]) * (mx[1] - mn[1]) + (mx[2] - mn[2]) * (mx[2] - mn[2]));
					return far2(r.o, mn, mx) + dg;
				};
				const float rl = rnode(n.lmin, n.lmax), rr = rnode(n.rmin, n.rmax);
				pl = n.le <= e_limit ? 0.35f * n.le * rl * dlen + slop : INFINITY;
				pr = n.re <= e_limit ? 0.35f * n.re * rr * dlen + slop : INFINITY;
				if (mode == 4 || mode == 5)
				{
					// the builder's own m-boxes; R = distance to the far corner of the child's box only
					const float fl_ = far2(r.o, n.lmin, n.lmax), fr_ = far2(r.o, n.rmin, n.rmax);
					const float sl = 3e-6f * (fl_ + std::fabs(r.o.x) + std::fabs(r.o.y) + std::fabs(r.o.z)), sr = 3e-6f * (fr_ + std::fabs(r.o.x) + std::fabs(r.o.y) + std::fabs(r.o.z));
					pl = n.le <= e_limit ? 0.20f * n.le * fl_ * dlen + sl : INFINITY;
					pr = n.re <= e_limit ? 0.20f * n.re * fr_ * dlen + sr : INFINITY;
					MBox bl, br;
					const float *mb = &bvh.mbox[12 * (size_t)node];
					for (int a = 0; a < 3; ++a) { bl.lo[a] = mb[a]; bl.hi[a] = mb[3 + a]; br.lo[a] = mb[6 + a]; br.hi[a] = mb[9 + a]; }
					const float lbl = mbox_lb(bl, r.d) - 1e-6f * dlen, lbr = mbox_lb(br, r.d) - 1e-6f * dlen;
					if (lbl > 2e-5f * dlen) pl = std::fmin(pl, 2.05e-6f * fl_ * dlen / lbl + sl);
					if (lbr > 2e-5f * dlen) pr = std::fmin(pr, 2.05e-6f * fr_ * dlen / lbr + sr);
				}
				if (mode == 3)
				{
					// |det| / (|e0||e1|) >= LB(|d . M|): margin 3.5e-6 R |d| / LB instead of the cutoff's 0.35 E R |d|
					const float lbl = mbox_lb(g_mbox[variant][2 * (size_t)node], r.d), lbr = mbox_lb(g_mbox[variant][2 * (size_t)node + 1], r.d);
					if (lbl > 1e-3f * dlen) pl = std::fmin(pl, 3.5e-6f * rl * dlen / lbl + slop);
					if (lbr > 1e-3f * dlen) pr = std::fmin(pr, 3.5e-6f * rr * dlen / lbr + slop);
				}
			}
			float lo_l, lo_r;
			bool vl = box_hit(r.o, inv, n.lmin, n.lmax, pl, best, lo_l);
			bool vr = box_hit(r.o, inv, n.rmin, n.rmax, pr, best, lo_r);
			if (mode == 5)
			{
				// anisotropic: in-plane share per axis + out-of-plane part
				const float fl_ = far2(r.o, n.lmin, n.lmax), fr_ = far2(r.o, n.rmin, n.rmax);
				const float o1 = std::fabs(r.o.x) + std::fabs(r.o.y) + std::fabs(r.o.z);
				const Aniso &al = g_aniso[2 * (size_t)node], &ar = g_aniso[2 * (size_t)node + 1];
				float p3l[3], p3r[3];
				for (int a = 0; a < 3; ++a)
				{
					const float sl = 3e-6f * (fl_ + o1), sr = 3e-6f * (fr_ + o1);
					p3l[a] = std::fmin(pl, (pl - sl) * al.ratio[a] + 2.2e-6f * al.sigma * fl_ + sl);
					p3r[a] = std::fmin(pr, (pr - sr) * ar.ratio[a] + 2.2e-6f * ar.sigma * fr_ + sr);
				}
				vl = box_hit3(r.o, inv, n.lmin, n.lmax, p3l, best, lo_l);
				vr = box_hit3(r.o, inv, n.rmin, n.rmax, p3r, best, lo_r);
			}
			if (mode == 4)
			{
				float dummy;
				const float small = 2e-4f * R;
				const float *mb = &bvh.mbox[12 * (size_t)node];
				MBox bl, br;
				for (int a = 0; a < 3; ++a) { bl.lo[a] = mb[a]; bl.hi[a] = mb[3 + a]; br.lo[a] = mb[6 + a]; br.hi[a] = mb[9 + a]; }
				if (vl && !box_hit(r.o, inv, n.lmin, n.lmax, small, best, dummy)) { (mbox_lb(bl, r.d) > 2e-5f * dlen ? st.excess_det : st.excess_e) += 1; if (n.le > 1e-2f) st.excess_big += 1; }
				if (vr && !box_hit(r.o, inv, n.rmin, n.rmax, small, best, dummy)) { (mbox_lb(br, r.d) > 2e-5f * dlen ? st.excess_det : st.excess_e) += 1; if (n.re > 1e-2f) st.excess_big += 1; }
			}
			bool left_first = !(lo_r < lo_l);
			if (vl && vr) { stack[sp++] = left_first ? n.right : n.left; node = left_first ? n.left : n.right; continue; }
			if (vl) { node = n.left; continue; }
			if (vr) { node = n.right; continue; }
		}
		else if (node != (int)0x80000000)
		{
			int leaf = ~node, first = leaf >> 4, cnt = (leaf & 15) + 1;
			for (int k = 0; k < cnt; ++k)
			{
				st.tris += 1;
				float t;
				int tri = bvh.tri_index[(size_t)first + k];
				if (tri_test(r, tri, best, t)) { best = t; best_id = tri; }
			}
		}
		if (!sp) break;
		node = stack[--sp];
	}
	t_best = best;
	return best_id;
}

static int count_tris(const WbrtBvh &bvh, int ref)
{
	if (ref >= 0) return count_tris(bvh, bvh.nodes[(size_t)ref].left) + count_tris(bvh, bvh.nodes[(size_t)ref].right);
	return ref == (int)0x80000000 ? 0 : ((~ref) & 15) + 1;
}
static void dump(const WbrtBvh &bvh, int node, int depth, int max_depth)
{
	if (node < 0 || depth > max_depth) return;
	const WbrtBvhNode &n = bvh.nodes[(size_t)node];
	const float *mb = &bvh.mbox[12 * (size_t)node];
	for (int side = 0; side < 2; ++side)
	{
		const float *mn = side ? n.rmin : n.lmin, *mx = side ? n.rmax : n.lmax;
		const float *m = mb + 6 * side;
		std::printf("%*s%c n=%5d E=%.2e box=[%.2f %.2f %.2f] m=[%.2f..%.2f %.2f..%.2f %.2f..%.2f]\n", 2 * depth, "", side ? 'R' : 'L',
		            count_tris(bvh, side ? n.right : n.left), side ? n.re : n.le, mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2], m[0], m[3], m[1], m[4], m[2], m[5]);
		dump(bvh, side ? n.right : n.left, depth + 1, max_depth);
	}
}

int main(int argc, char **argv)
{
	if (argc < 2) { std::printf("usage: bvh_sim tris.f32 [width]\n"); return 2; }
	FILE *f = std::fopen(argv[1], "rb");
	if (!f) return 1;
	std::fseek(f, 0, SEEK_END);
	long bytes = std::ftell(f);
	std::fseek(f, 0, SEEK_SET);
	const int T = (int)(bytes / 36);
	std::vector<float> v((size_t)T * 9);
	if (std::fread(v.data(), 36, (size_t)T, f) != (size_t)T) return 1;
	std::fclose(f);
	const int W = argc > 2 ? std::atoi(argv[2]) : 240;
	g_geom.resize((size_t)T * 3);
	for (int i = 0; i < T; ++i)
	{
		const float *p = &v[(size_t)i * 9];
		volatile float e0x = p[3] - p[0], e0y = p[4] - p[1], e0z = p[5] - p[2], e1x = p[6] - p[0], e1y = p[7] - p[1], e1z = p[8] - p[2];
		g_geom[3 * (size_t)i] = make_float4(p[0], p[1], p[2], e0x);
		g_geom[3 * (size_t)i + 1] = make_float4(e0y, e0z, e1x, e1y);
		g_geom[3 * (size_t)i + 2] = make_float4(e1z, 0, 0, 0);
	}
	int2 range = make_int2(0, T);
	for (int variant = 0; variant < 2; ++variant)
	{
		wbrt_build_bvh(g_geom.data(), &range, 1, variant == 1, g_bvh[variant]);
		std::printf("variant %d: %zu nodes, depth %d\n", variant, g_bvh[variant].nodes.size(), g_bvh[variant].depth);
		build_mboxes(variant);
		if (variant == 1) build_aniso();
		if (variant == 1 && std::getenv("DUMP")) dump(g_bvh[1], g_bvh[1].roots[0].ref, 0, std::atoi(std::getenv("DUMP")));
	}
	wbrt_camera cam;
	wbrt_camera_main_cpp(&cam, W, 1, 50);
	const int H = cam.height;
	std::mt19937 rng(7);
	std::uniform_real_distribution<float> U(0.0f, 1.0f);
	// rays: camera rays and, from mesh hits, metal-fuzz bounces (main.cpp material: fuzz 0.5)
	std::vector<Ray> rays;
	std::vector<int> gen;
	for (int y = 0; y < H; ++y)
		for (int x = 0; x < W; ++x)
		{
			float u = (x + U(rng)) / W, vv = (y + U(rng)) / H;
			V3 org = { cam.origin.x, cam.origin.y, cam.origin.z };
			V3 d = { cam.lowerLeftCorner.x + cam.horizontal.x * u + cam.vertical.x * vv - org.x,
			         cam.lowerLeftCorner.y + cam.horizontal.y * u + cam.vertical.y * vv - org.y,
			         cam.lowerLeftCorner.z + cam.horizontal.z * u + cam.vertical.z * vv - org.z };
			rays.push_back({ org, d });
			gen.push_back(0);
		}
	Stats ref_stats;
	for (size_t i = 0; i < rays.size() && rays.size() < 4000000; ++i)
	{
		if (gen[i] >= 6) continue;
		float t;
		int id = walk(g_bvh[0], 0, rays[i], 0, 2e-4f, INFINITY, t, ref_stats);
		if (id < 0) continue;
		const float4 g0 = g_geom[3 * (size_t)id], g1 = g_geom[3 * (size_t)id + 1], g2 = g_geom[3 * (size_t)id + 2];
		V3 e0 = { g0.w, g1.x, g1.y }, e1 = { g1.z, g1.w, g2.x };
		V3 n = cross(e0, e1);
		float nl = len(n);
		if (!(nl > 0)) continue;
		n = n * (1.0f / nl);
		if (dot(n, rays[i].d) > 0) n = n * -1.0f;
		V3 ud = rays[i].d * (1.0f / len(rays[i].d));
		V3 refl = ud - n * (2.0f * dot(ud, n));
		V3 p;
		do { p = { 2 * U(rng) - 1, 2 * U(rng) - 1, 2 * U(rng) - 1 }; } while (dot(p, p) >= 1.0f);
		V3 nd = refl + p * 0.5f;
		if (dot(nd, n) <= 0) continue;
		rays.push_back({ rays[i].o + rays[i].d * t, nd });
		gen.push_back(gen[i] + 1);
	}
	size_t n_cam = (size_t)W * H;
	std::printf("%zu rays (%zu camera, %zu bounce)\n", rays.size(), n_cam, rays.size() - n_cam);
	struct Cfg { const char *name; int variant, mode; float rho; };
	const Cfg cfgs[] = { { "capped 2e-4 (plain tree)", 0, 0, 2e-4f }, { "strict shipped (split tree)", 1, 1, 0 }, { "strict shipped (plain tree)", 0, 1, 0 },
	                     { "strict per-node R (split tree)", 1, 2, 0 }, { "strict per-node R (plain tree)", 0, 2, 0 },
	                     { "strict + M boxes (split tree)", 1, 3, 0 }, { "strict, builder m-boxes (margin tree)", 1, 4, 0 }, { "  + anisotropic margins", 1, 5, 0 }, { "strict + M boxes (plain tree)", 0, 3, 0 } };
	for (const Cfg &c : cfgs)
	{
		Stats cs, bs;
		size_t mismatch = 0;
		for (size_t i = 0; i < rays.size(); ++i)
		{
			float t, t0;
			Stats dummy;
			int want = walk(g_bvh[0], 0, rays[i], 0, 2e-4f, INFINITY, t0, dummy);
			int got = walk(g_bvh[c.variant], c.variant, rays[i], c.mode, c.rho, INFINITY, t, i < n_cam ? cs : bs);
			if (got != want && !(t == t0)) ++mismatch;
		}
		if (c.mode == 4) std::printf("   excess child entries per camera ray: %.2f with E margin (%.2f of them E > 1e-2), %.2f with det margin; bounce: %.2f / %.2f\n", cs.excess_e / n_cam, cs.excess_big / n_cam, cs.excess_det / n_cam, bs.excess_e / (rays.size() - n_cam), bs.excess_det / (rays.size() - n_cam));
		std::printf("%-34s camera: %7.1f nodes %6.1f tris | bounce: %7.1f nodes %6.1f tris | mismatches %zu\n", c.name, cs.nodes / n_cam, cs.tris / n_cam,
		            bs.nodes / std::max<size_t>(1, rays.size() - n_cam), bs.tris / std::max<size_t>(1, rays.size() - n_cam), mismatch);
	}
	return 0;
}
