// Host-side builder of the triangle hierarchies the render kernel walks.
//
// The reference tests every triangle of a bounding box whose slab test passes
// (gpu_kernel.cl:374-391) and keeps the nearest accepted hit, first triangle winning ties
// (gpu_kernel.cl:520).  The winner of that loop is the lexicographic minimum of
// (t0, position in the loop) over the triangles its Moller-Trumbore test accepts, so any
// visiting order that reaches every triangle the test COULD accept gives the same answer.
// One binary hierarchy of axis-aligned boxes per bounding box lets the kernel skip the rest.
//
// What "could accept" means is decided by the float arithmetic of gpu_kernel.cl:489-520, not
// by geometry: with u = 2^-24 a forward error analysis of that code bounds the distance between
// the ray point of an accepted hit and the triangle by 26.5u |o-p0| |d||e0||e1| / |det|, det being
// the float determinant the test computed.  |det| is bounded from below (1) by the test's own
// cutoff 1e-5 and (2) by |e0||e1| times a lower bound of |d . m|, m = (e0 x e1)/(|e0||e1|), over
// the triangles of a node.  Nodes therefore carry E = max |e0||e1| and a box around the m vectors
// of their triangles (m-box), and the kernel widens each box per ray by the smaller of the two
// margins (see the kernel for the formula).  Two builders: plain binned surface-area trees for
// the walk with capped margins, and MarginBuilder, which prices splits with those margins and
// also splits by orientation, for the walk with proven margins.
#include "rt_device.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <cstdlib>

#ifndef WBRT_KLEAF
#define WBRT_KLEAF 4 // triangles per leaf at most (build knob: 2, 6 and 8 measured no better)
#endif

namespace {

struct Bounds
{
	double mn[3], mx[3];
	void reset() { for (int a = 0; a < 3; ++a) { mn[a] = DBL_MAX; mx[a] = -DBL_MAX; } }
	void add(const double p[3]) { for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], p[a]); mx[a] = std::max(mx[a], p[a]); } }
	void add(const Bounds &b) { for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], b.mn[a]); mx[a] = std::max(mx[a], b.mx[a]); } }
	double area() const
	{
		double x = mx[0] - mn[0], y = mx[1] - mn[1], z = mx[2] - mn[2];
		if (x < 0 || y < 0 || z < 0) return 0.0;
		return 2.0 * (x * y + y * z + z * x);
	}
};

struct Prim
{
	Bounds b;
	double c[3]; // centroid of the bounds
	double e;    // |e0||e1|
	double m[3]; // (e0 x e1) / (|e0||e1|): |d . m| = |det| / (|e0||e1|) for a ray direction d; sign free (two-sided test)
	double area; // |e0 x e1|
	int index;   // position in the grouped triangle order
};

// Box around the m vectors of a set of triangles, each flipped into the set's own hemisphere.
struct MBounds
{
	double lo[3], hi[3];
	void reset() { for (int a = 0; a < 3; ++a) { lo[a] = DBL_MAX; hi[a] = -DBL_MAX; } }
	void add(const double v[3]) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], v[a]); hi[a] = std::max(hi[a], v[a]); } }
	void add(const MBounds &b) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
	bool empty() const { return lo[0] > hi[0]; }
	// lower bound of |d . m| over the box
	double lower(const double d[3]) const
	{
		if (empty()) return 0.0;
		double mn = 0.0, mx = 0.0;
		for (int a = 0; a < 3; ++a) { mn += std::min(d[a] * lo[a], d[a] * hi[a]); mx += std::max(d[a] * lo[a], d[a] * hi[a]); }
		return mn > 0.0 ? mn : mx < 0.0 ? -mx : 0.0;
	}
};

float round_down(double v) { float f = (float)v; return (double)f > v ? std::nextafterf(f, -INFINITY) : f; }
float round_up(double v) { float f = (float)v; return (double)f < v ? std::nextafterf(f, INFINITY) : f; }

struct Builder
{
	std::vector<Prim> prims;
	WbrtBvh *out;
	int depth_limit;
	bool split_by_e;
	int max_depth = 0;

	// inner levels a balanced (object-median) subtree over `count` triangles needs
	static int balanced_levels(int count)
	{
		int leaves = (count + kLeaf - 1) / kLeaf, levels = 0;
		while ((1 << levels) < leaves) ++levels;
		return levels;
	}

	static constexpr int kLeaf = WBRT_KLEAF, kBins = 16;

	int make_leaf(int first, int count)
	{
		const int base = (int)out->tri_index.size();
		for (int i = 0; i < count; ++i) out->tri_index.push_back(prims[(size_t)first + i].index);
		return ~((base << 4) | (count - 1));
	}

	// returns the child reference of the subtree over prims[first, first+count)
	int build(int first, int count, int depth, Bounds &bounds, double &emax)
	{
		max_depth = std::max(max_depth, depth);
		bounds.reset();
		emax = 0.0;
		Bounds cb;
		cb.reset();
		for (int i = first; i < first + count; ++i)
		{
			bounds.add(prims[(size_t)i].b);
			cb.add(prims[(size_t)i].c);
			emax = std::max(emax, prims[(size_t)i].e);
		}
		if (count <= kLeaf) return make_leaf(first, count);

		int axis = 0;
		for (int a = 1; a < 3; ++a) if (cb.mx[a] - cb.mn[a] > cb.mx[axis] - cb.mn[axis]) axis = a;
		int mid = -1;
		// A node's box margin grows with the largest |e0||e1| below it, so a few large triangles
		// would widen every box above them.  When they are a small minority they get a subtree of
		// their own, whatever their position.
		if (split_by_e && count > 4 * kLeaf && emax > 0.0 && std::isfinite(emax))
		{
			const double cut = emax / 8.0;
			int big = 0;
			for (int i = first; i < first + count; ++i) big += prims[(size_t)i].e > cut;
			if (big > 0 && big * 8 <= count && depth + 1 + balanced_levels(count - big) <= depth_limit)
			{
				auto it = std::partition(prims.begin() + first, prims.begin() + first + count,
				                         [&](const Prim &p) { return !(p.e > cut); });
				mid = (int)(it - prims.begin());
			}
		}
		const double extent = cb.mx[axis] - cb.mn[axis];
		if (mid < 0 && extent > 0.0 && std::isfinite(extent))
		{
			// binned surface-area heuristic on the widest centroid axis
			Bounds bin_b[kBins];
			int bin_n[kBins];
			for (int k = 0; k < kBins; ++k) { bin_b[k].reset(); bin_n[k] = 0; }
			auto bin_of = [&](const Prim &p) {
				int k = (int)((p.c[axis] - cb.mn[axis]) / extent * kBins);
				return std::min(std::max(k, 0), kBins - 1);
			};
			for (int i = first; i < first + count; ++i)
			{
				int k = bin_of(prims[(size_t)i]);
				bin_b[k].add(prims[(size_t)i].b);
				bin_n[k]++;
			}
			double right_area[kBins];
			Bounds acc;
			acc.reset();
			for (int k = kBins - 1; k > 0; --k) { acc.add(bin_b[k]); right_area[k] = acc.area(); }
			acc.reset();
			double best = DBL_MAX;
			int best_k = -1, nl = 0;
			for (int k = 0; k < kBins - 1; ++k)
			{
				acc.add(bin_b[k]);
				nl += bin_n[k];
				const int nr = count - nl;
				if (!nl || !nr) continue;
				const double cost = acc.area() * nl + right_area[k + 1] * nr;
				if (cost < best) { best = cost; best_k = k; }
			}
			if (best_k >= 0)
			{
				auto it = std::partition(prims.begin() + first, prims.begin() + first + count,
				                         [&](const Prim &p) { return bin_of(p) <= best_k; });
				mid = (int)(it - prims.begin());
				if (mid == first || mid == first + count) mid = -1;
				// a lopsided split is taken only if the larger side can still be finished, balanced
				// if need be, within the depth the kernel's stack allows
				else if (depth + 1 + balanced_levels(std::max(mid - first, first + count - mid)) > depth_limit) mid = -1;
			}
		}
		if (mid < 0)
		{
			// object median (also the fallback that bounds the depth); ties broken by index so that
			// the split always has two sides
			mid = first + count / 2;
			std::nth_element(prims.begin() + first, prims.begin() + mid, prims.begin() + first + count,
			                 [&](const Prim &a, const Prim &b) {
				                 if (a.c[axis] != b.c[axis]) return a.c[axis] < b.c[axis];
				                 return a.index < b.index;
			                 });
		}
		const size_t node = out->nodes.size();
		out->nodes.emplace_back();
		Bounds lb, rb;
		double le, re;
		const int left = build(first, mid - first, depth + 1, lb, le);
		const int right = build(mid, first + count - mid, depth + 1, rb, re);
		WbrtBvhNode n;
		for (int a = 0; a < 3; ++a)
		{
			n.lmin[a] = round_down(lb.mn[a]); n.lmax[a] = round_up(lb.mx[a]);
			n.rmin[a] = round_down(rb.mn[a]); n.rmax[a] = round_up(rb.mx[a]);
		}
		n.left = left; n.right = right;
		n.le = round_up(le * 1.000001); n.re = round_up(re * 1.000001);
		out->nodes[node] = n;
		return (int)node;
	}
};

// Builder of the hierarchies walked with PROVEN margins.  How far outside a node's box the reference's float
// triangle test can still accept a hit depends on the ray: margin = R |d| min(0.35 E, 3.5e-6 / L) with E the largest
// |e0||e1| below the node and L a lower bound of |d^ . m| over the node's m-box (see the kernel).  Boxes of triangles
// that are large AND seen edge-on by some rays are therefore fat for those rays; a tree that only looks at positions
// mixes such triangles with everything else near them.  This builder prices every candidate split with the boxes
// widened by the margin averaged over 13 directions, and considers, besides the usual spatial planes, splits by
// the components of m (orientation) and by log E (size), so that orientation classes and size classes separate
// wherever that pays.
struct MarginBuilder
{
	std::vector<Prim> prims;
	WbrtBvh *out;
	int depth_limit;
	int max_depth = 0;
	double rd; // nominal R |d| the margins are priced at

	static constexpr int kLeaf = WBRT_KLEAF, kBins = 16, kKeys = 6, kDirs = 13;

	static int balanced_levels(int count)
	{
		int leaves = (count + kLeaf - 1) / kLeaf, levels = 0;
		while ((1 << levels) < leaves) ++levels;
		return levels;
	}

	struct Agg
	{
		Bounds b; MBounds m; double e; int n;
		void reset() { b.reset(); m.reset(); e = 0.0; n = 0; }
		void add(const Prim &p) { b.add(p.b); m.add(p.m); e = std::max(e, p.e); ++n; }
		void add(const Agg &a) { if (!a.n) return; b.add(a.b); m.add(a.m); e = std::max(e, a.e); n += a.n; }
	};

	double price(const Agg &a) const
	{
		if (!a.n) return 0.0;
		static const double dirs[kDirs][3] = {
			{ 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 },
			{ 0.7071067811865476, 0.7071067811865476, 0 }, { 0.7071067811865476, -0.7071067811865476, 0 },
			{ 0.7071067811865476, 0, 0.7071067811865476 }, { 0.7071067811865476, 0, -0.7071067811865476 },
			{ 0, 0.7071067811865476, 0.7071067811865476 }, { 0, 0.7071067811865476, -0.7071067811865476 },
			{ 0.5773502691896258, 0.5773502691896258, 0.5773502691896258 }, { 0.5773502691896258, 0.5773502691896258, -0.5773502691896258 },
			{ 0.5773502691896258, -0.5773502691896258, 0.5773502691896258 }, { -0.5773502691896258, 0.5773502691896258, 0.5773502691896258 } };
		double x = a.b.mx[0] - a.b.mn[0], y = a.b.mx[1] - a.b.mn[1], z = a.b.mx[2] - a.b.mn[2];
		if (!(x >= 0 && y >= 0 && z >= 0) || !std::isfinite(x + y + z)) return DBL_MAX / 4;
		const double by_e = std::isfinite(a.e) ? 0.20 * a.e : 1e30;
		double sum = 0.0;
		for (int k = 0; k < kDirs; ++k)
		{
			const double l = a.m.lower(dirs[k]);
			double m = l > 2e-5 ? std::min(by_e, 2.05e-6 / l) : by_e;
			m = std::min(m * rd, 4.0 * rd); // unbounded margins: priced as "always entered"
			const double X = x + 2 * m, Y = y + 2 * m, Z = z + 2 * m;
			sum += X * Y + Y * Z + Z * X;
		}
		return sum / kDirs * a.n;
	}

	// keys 0-2: centroid; 3-5: components of m; 6: size class, log2 E over the 16 octaves below the range's largest E
	double key_of(const Prim &p, int key, double log_emax) const
	{
		if (key < 3) return p.c[key];
		if (key < 6) return p.m[key - 3];
		const double l = p.e > 0.0 ? (std::isfinite(p.e) ? std::log2(p.e) : 1100.0) : -1100.0;
		return std::max(l, log_emax - 16.0);
	}

	int make_leaf(int first, int count)
	{
		const int base = (int)out->tri_index.size();
		for (int i = 0; i < count; ++i) out->tri_index.push_back(prims[(size_t)first + i].index);
		return ~((base << 4) | (count - 1));
	}

	// flips every m of the range into one hemisphere: that of the range's area-weighted mean
	void orient(int first, int count)
	{
		int big = first;
		for (int i = first; i < first + count; ++i) if (prims[(size_t)i].area > prims[(size_t)big].area) big = i;
		const double *ref = prims[(size_t)big].m;
		double mean[3] = { 0, 0, 0 };
		for (int i = first; i < first + count; ++i)
		{
			Prim &p = prims[(size_t)i];
			const double s = p.m[0] * ref[0] + p.m[1] * ref[1] + p.m[2] * ref[2] < 0 ? -1.0 : 1.0;
			for (int a = 0; a < 3; ++a) mean[a] += s * p.m[a] * p.area;
		}
		for (int i = first; i < first + count; ++i)
		{
			Prim &p = prims[(size_t)i];
			if (p.m[0] * mean[0] + p.m[1] * mean[1] + p.m[2] * mean[2] < 0) for (int a = 0; a < 3; ++a) p.m[a] = -p.m[a];
		}
	}

	int build(int first, int count, int depth, Bounds &bounds, double &emax, MBounds &mb)
	{
		max_depth = std::max(max_depth, depth);
		orient(first, count);
		Agg all;
		all.reset();
		for (int i = first; i < first + count; ++i) all.add(prims[(size_t)i]);
		bounds = all.b; emax = all.e; mb = all.m;
		if (count <= kLeaf) return make_leaf(first, count);

		double best = DBL_MAX;
		int best_key = -1, best_bin = -1;
		double best_lo = 0, best_scale = 0;
		const double log_emax = all.e > 0.0 ? (std::isfinite(all.e) ? std::log2(all.e) : 1100.0) : -1100.0;
		for (int key = 0; key < kKeys; ++key)
		{
			double lo = DBL_MAX, hi = -DBL_MAX;
			for (int i = first; i < first + count; ++i) { const double k = key_of(prims[(size_t)i], key, log_emax); lo = std::min(lo, k); hi = std::max(hi, k); }
			const double extent = hi - lo;
			if (!(extent > 0.0) || !std::isfinite(extent)) continue;
			const double scale = kBins / extent;
			Agg bins[kBins], right[kBins];
			for (int k = 0; k < kBins; ++k) bins[k].reset();
			for (int i = first; i < first + count; ++i)
			{
				const Prim &p = prims[(size_t)i];
				const int k = std::min(std::max((int)((key_of(p, key, log_emax) - lo) * scale), 0), kBins - 1);
				bins[k].add(p);
			}
			Agg acc;
			acc.reset();
			for (int k = kBins - 1; k > 0; --k) { acc.add(bins[k]); right[k] = acc; }
			acc.reset();
			for (int k = 0; k < kBins - 1; ++k)
			{
				acc.add(bins[k]);
				const int nl = acc.n, nr = count - nl;
				if (!nl || !nr) continue;
				// a lopsided split is taken only if the larger side can still be finished, balanced if need be,
				// within the depth the kernel's stack allows
				if (depth + 1 + balanced_levels(std::max(nl, nr)) > depth_limit) continue;
				const double cost = price(acc) + price(right[k + 1]);
				if (cost < best) { best = cost; best_key = key; best_bin = k; best_lo = lo; best_scale = scale; }
			}
		}
		int mid = -1;
		if (best_key >= 0)
		{
			auto it = std::partition(prims.begin() + first, prims.begin() + first + count, [&](const Prim &p) {
				const int k = std::min(std::max((int)((key_of(p, best_key, log_emax) - best_lo) * best_scale), 0), kBins - 1);
				return k <= best_bin;
			});
			mid = (int)(it - prims.begin());
			if (mid == first || mid == first + count) mid = -1;
		}
		if (mid < 0)
		{
			int axis = 0;
			Bounds cb;
			cb.reset();
			for (int i = first; i < first + count; ++i) cb.add(prims[(size_t)i].c);
			for (int a = 1; a < 3; ++a) if (cb.mx[a] - cb.mn[a] > cb.mx[axis] - cb.mn[axis]) axis = a;
			mid = first + count / 2;
			std::nth_element(prims.begin() + first, prims.begin() + mid, prims.begin() + first + count,
			                 [&](const Prim &a, const Prim &b) {
				                 if (a.c[axis] != b.c[axis]) return a.c[axis] < b.c[axis];
				                 return a.index < b.index;
			                 });
		}
		const size_t node = out->nodes.size();
		out->nodes.emplace_back();
		out->mbox.resize(out->mbox.size() + 12);
		Bounds lb, rb;
		MBounds lm, rm;
		double le, re;
		const int left = build(first, mid - first, depth + 1, lb, le, lm);
		const int right = build(mid, first + count - mid, depth + 1, rb, re, rm);
		WbrtBvhNode n;
		for (int a = 0; a < 3; ++a)
		{
			n.lmin[a] = round_down(lb.mn[a]); n.lmax[a] = round_up(lb.mx[a]);
			n.rmin[a] = round_down(rb.mn[a]); n.rmax[a] = round_up(rb.mx[a]);
		}
		n.left = left; n.right = right;
		n.le = round_up(le * 1.000001); n.re = round_up(re * 1.000001);
		out->nodes[node] = n;
		float *mbx = &out->mbox[12 * node];
		for (int a = 0; a < 3; ++a)
		{
			mbx[a] = round_down(lm.lo[a]); mbx[3 + a] = round_up(lm.hi[a]);
			mbx[6 + a] = round_down(rm.lo[a]); mbx[9 + a] = round_up(rm.hi[a]);
		}
		return (int)node;
	}
};

} // namespace

// tri_geom: the grouped triangle array of the device scene (3 float4 per triangle: p0.xyz e0.x |
// e0.yz e1.xy | e1.z - - -); ranges[b] = [first, last) of box b.  Node and leaf indices are
// global over all boxes.  split_by_e: give minorities of large triangles subtrees of their own
// (tighter strict margins, looser boxes).
void wbrt_build_bvh(const float4 *tri_geom, const int2 *ranges, int B, bool split_by_e, WbrtBvh &bvh)
{
	bvh.nodes.clear();
	bvh.tri_index.clear();
	bvh.mbox.clear();
	bvh.roots.assign((size_t)B, WbrtBvhRoot{});
	bvh.depth = 0;
	for (int b = 0; b < B; ++b)
	{
		WbrtBvhRoot &root = bvh.roots[(size_t)b];
		const int first = ranges[b].x, count = ranges[b].y - ranges[b].x;
		root.ref = 0; root.empty = 1;
		if (count <= 0) continue;
		std::vector<Prim> prims((size_t)count);
		for (int i = 0; i < count; ++i)
		{
			const float4 g0 = tri_geom[3 * (size_t)(first + i)], g1 = tri_geom[3 * (size_t)(first + i) + 1], g2 = tri_geom[3 * (size_t)(first + i) + 2];
			const double p0[3] = { g0.x, g0.y, g0.z };
			const double e0[3] = { g0.w, g1.x, g1.y }, e1[3] = { g1.z, g1.w, g2.x };
			double p1[3], p2[3];
			for (int a = 0; a < 3; ++a) { p1[a] = p0[a] + e0[a]; p2[a] = p0[a] + e1[a]; }
			Prim &p = prims[(size_t)i];
			p.b.reset(); p.b.add(p0); p.b.add(p1); p.b.add(p2);
			for (int a = 0; a < 3; ++a)
			{
				// NaN or infinite vertices: an unbounded box keeps the triangle reachable
				if (!std::isfinite(p.b.mn[a]) || !std::isfinite(p.b.mx[a]) || p.b.mn[a] > p.b.mx[a]) { p.b.mn[a] = -DBL_MAX; p.b.mx[a] = DBL_MAX; }
				p.c[a] = 0.5 * p.b.mn[a] + 0.5 * p.b.mx[a];
			}
			p.e = std::sqrt(e0[0] * e0[0] + e0[1] * e0[1] + e0[2] * e0[2]) * std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
			const double nx = e0[1] * e1[2] - e0[2] * e1[1], ny = e0[2] * e1[0] - e0[0] * e1[2], nz = e0[0] * e1[1] - e0[1] * e1[0];
			p.area = std::sqrt(nx * nx + ny * ny + nz * nz);
			if (p.e > 0.0 && std::isfinite(p.e) && std::isfinite(p.area)) { p.m[0] = nx / p.e; p.m[1] = ny / p.e; p.m[2] = nz / p.e; }
			else { p.m[0] = p.m[1] = p.m[2] = 0.0; p.area = 0.0; } // the origin in an m-box: no bound from it
			if (!std::isfinite(p.e)) p.e = DBL_MAX;
			p.index = first + i;
		}
		if (split_by_e)
		{
			MarginBuilder mbld;
			mbld.prims = std::move(prims);
			mbld.out = &bvh;
			mbld.depth_limit = WBRT_BVH_STACK;
			Bounds all;
			all.reset();
			for (const Prim &p : mbld.prims) for (int a = 0; a < 3; ++a) if (p.b.mn[a] > -DBL_MAX / 2 && p.b.mx[a] < DBL_MAX / 2) { all.mn[a] = std::min(all.mn[a], p.b.mn[a]); all.mx[a] = std::max(all.mx[a], p.b.mx[a]); }
			double diag = 0.0;
			for (int a = 0; a < 3; ++a) if (all.mx[a] > all.mn[a]) diag += (all.mx[a] - all.mn[a]) * (all.mx[a] - all.mn[a]);
			mbld.rd = 2.0 * std::sqrt(diag) + 1e-30;
			Bounds rb;
			MBounds rm;
			double re;
			root.ref = mbld.build(0, count, 0, rb, re, rm);
			for (int a = 0; a < 3; ++a) { root.mn[a] = round_down(rb.mn[a]); root.mx[a] = round_up(rb.mx[a]); }
			root.e = round_up(re * 1.000001);
			root.empty = 0;
			bvh.depth = std::max(bvh.depth, mbld.max_depth);
			continue;
		}
		Builder bl;
		bl.prims = std::move(prims);
		bl.out = &bvh;
		bl.depth_limit = WBRT_BVH_STACK;
		bl.split_by_e = split_by_e;
		Bounds rb;
		double re;
		root.ref = bl.build(0, count, 0, rb, re);
		for (int a = 0; a < 3; ++a) { root.mn[a] = round_down(rb.mn[a]); root.mx[a] = round_up(rb.mx[a]); }
		root.e = round_up(re * 1.000001);
		root.empty = 0;
		bvh.depth = std::max(bvh.depth, bl.max_depth);
	}
	if (bvh.mbox.size() != bvh.nodes.size() * 12) // trees built without m-boxes: nothing is known
	{
		bvh.mbox.assign(bvh.nodes.size() * 12, 0.0f);
		for (size_t n = 0; n < bvh.nodes.size(); ++n)
			for (int k = 0; k < 12; ++k) bvh.mbox[12 * n + k] = (k % 6) < 3 ? -1.0f : 1.0f;
	}
}

// Appends one more hierarchy, over the SPHERES, to bvh (its root becomes bvh.roots.back()).
// sph_geom: center.xyz, radius^2; radius: the radii (negative ones are legal: the reference only
// uses radius^2 for the test).  Leaves reference positions in the sphere array; nodes carry the
// largest |radius| below each child where triangle nodes carry E.
//
// The reference tests every sphere (gpu_kernel.cl:361-371) and keeps the nearest accepted root,
// the LATER sphere winning ties (its bounds are inclusive, gpu_kernel.cl:466-473).  That winner
// is the minimum of (root, -index) over all spheres whose float test accepts, whatever the
// order.  The float test accepts only if the ray passes within sqrt(r^2 + ~18u (|o-c|^2 + r^2))
// of the centre, and the root it returns lies within that distance of the centre too, so
// boxes widened by 2e-3 (|o - c| + r) for every sphere inside never hide a sphere it would accept.
void wbrt_append_sphere_tree(const float4 *sph_geom, const float *radius, int S, WbrtBvh &bvh)
{
	std::vector<Prim> prims((size_t)S);
	for (int i = 0; i < S; ++i)
	{
		Prim &p = prims[(size_t)i];
		const double r = std::fabs((double)radius[i]);
		const double c[3] = { sph_geom[i].x, sph_geom[i].y, sph_geom[i].z };
		for (int a = 0; a < 3; ++a) { p.b.mn[a] = c[a] - r; p.b.mx[a] = c[a] + r; p.c[a] = c[a]; }
		p.e = r;
		p.index = i;
	}
	WbrtBvhRoot root{};
	Builder bl;
	bl.prims = std::move(prims);
	bl.out = &bvh;
	bl.depth_limit = WBRT_BVH_STACK;
	bl.split_by_e = true; // a few huge spheres (a ground sphere) must not widen every box
	Bounds rb;
	double re;
	root.ref = bl.build(0, S, 0, rb, re);
	for (int a = 0; a < 3; ++a) { root.mn[a] = round_down(rb.mn[a]); root.mx[a] = round_up(rb.mx[a]); }
	root.e = round_up(re * 1.000001);
	root.empty = 0;
	bvh.roots.push_back(root);
	bvh.depth = std::max(bvh.depth, bl.max_depth);
	while (bvh.mbox.size() < bvh.nodes.size() * 12)
		for (int k = 0; k < 12; ++k) bvh.mbox.push_back((k % 6) < 3 ? -1.0f : 1.0f);
}

// Test hook (not in the public header): runs the builder on host arrays and copies the result
// out.  tri_geom: 12 floats per triangle in the device layout; ranges: first,last per box;
// nodes_out: 16 floats per node; roots_out: 8 floats per box (min.xyz ref | max.xyz E, E < 0 =
// empty).  Returns the node count, or -(needed nodes) - 1 when nodes_cap is too small.
extern "C" int wbrt_bvh_build_(const float *tri_geom, const int *ranges, int B, float *nodes_out, int nodes_cap,
                               int *tri_index_out, int tri_cap, float *roots_out, int *depth_out, int split_by_e)
{
	WbrtBvh bvh;
	wbrt_build_bvh(reinterpret_cast<const float4 *>(tri_geom), reinterpret_cast<const int2 *>(ranges), B, split_by_e != 0, bvh);
	if ((int)bvh.nodes.size() > nodes_cap || (int)bvh.tri_index.size() > tri_cap) return -(int)bvh.nodes.size() - 1;
	if (!bvh.nodes.empty()) std::memcpy(nodes_out, bvh.nodes.data(), bvh.nodes.size() * sizeof(WbrtBvhNode));
	if (!bvh.tri_index.empty()) std::memcpy(tri_index_out, bvh.tri_index.data(), bvh.tri_index.size() * sizeof(int));
	for (int b = 0; b < B; ++b)
	{
		const WbrtBvhRoot &r = bvh.roots[(size_t)b];
		float *o = roots_out + 8 * (size_t)b;
		o[0] = r.mn[0]; o[1] = r.mn[1]; o[2] = r.mn[2];
		std::memcpy(&o[3], &r.ref, 4);
		o[4] = r.mx[0]; o[5] = r.mx[1]; o[6] = r.mx[2];
		o[7] = r.empty ? -1.0f : r.e;
	}
	if (depth_out) *depth_out = bvh.depth;
	return (int)bvh.nodes.size();
}


// ------------------------------------------------------------------ hierarchy over the BOXES
// Scenes with many bounding boxes: the reference tests every box for every ray segment
// (gpu_kernel.cl:374-376).  A balanced binary tree over the boxes' stated bounds lets a ray skip
// groups of boxes its slab test would reject; the kernel widens every node per ray by more than
// the rounding of that float test, and runs the reference's own test at the leaves.
// Nodes: 4 float4 {lmin.xyz lmax.x | lmax.yz rmin.xy | rmin.z rmax.xyz | left right - -}; a child
// reference >= 0 names a node, < 0 names a leaf: ~ref = rank of the box in left-to-right order.
// Boxes without triangles are left out.  Returns false (no tree) when a box has bounds that are not
// finite or not ordered, or when fewer than two boxes hold triangles.
namespace {

struct BoxItem { float mn[3], mx[3]; int index; };

int build_box_node(std::vector<BoxItem> &items, int first, int count, std::vector<float4> &nodes, std::vector<int> &box_of_rank,
                   int level, int &depth)
{
	if (count == 1)
	{
		box_of_rank.push_back(items[(size_t)first].index);
		return ~((int)box_of_rank.size() - 1);
	}
	depth = std::max(depth, level + 1);
	float cmn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, cmx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
	for (int i = first; i < first + count; ++i)
		for (int a = 0; a < 3; ++a)
		{
			const float c = 0.5f * items[(size_t)i].mn[a] + 0.5f * items[(size_t)i].mx[a];
			cmn[a] = std::min(cmn[a], c); cmx[a] = std::max(cmx[a], c);
		}
	int axis = 0;
	for (int a = 1; a < 3; ++a) if (cmx[a] - cmn[a] > cmx[axis] - cmn[axis]) axis = a;
	const int half = count / 2;
	std::nth_element(items.begin() + first, items.begin() + first + half, items.begin() + first + count,
	                 [axis](const BoxItem &x, const BoxItem &y) {
		                 const float cx = x.mn[axis] + x.mx[axis], cy = y.mn[axis] + y.mx[axis];
		                 return cx < cy || (cx == cy && x.index < y.index);
	                 });
	const int self = (int)(nodes.size() / 4);
	nodes.resize(nodes.size() + 4);
	float b[2][6];
	const int starts[2] = { first, first + half }, counts[2] = { half, count - half };
	for (int k = 0; k < 2; ++k)
	{
		for (int a = 0; a < 3; ++a) { b[k][a] = FLT_MAX; b[k][3 + a] = -FLT_MAX; }
		for (int i = starts[k]; i < starts[k] + counts[k]; ++i)
			for (int a = 0; a < 3; ++a)
			{
				b[k][a] = std::min(b[k][a], items[(size_t)i].mn[a]);
				b[k][3 + a] = std::max(b[k][3 + a], items[(size_t)i].mx[a]);
			}
	}
	const int left = build_box_node(items, starts[0], counts[0], nodes, box_of_rank, level + 1, depth);
	const int right = build_box_node(items, starts[1], counts[1], nodes, box_of_rank, level + 1, depth);
	float li, ri;
	std::memcpy(&li, &left, 4);
	std::memcpy(&ri, &right, 4);
	nodes[4 * (size_t)self + 0] = make_float4(b[0][0], b[0][1], b[0][2], b[0][3]);
	nodes[4 * (size_t)self + 1] = make_float4(b[0][4], b[0][5], b[1][0], b[1][1]);
	nodes[4 * (size_t)self + 2] = make_float4(b[1][2], b[1][3], b[1][4], b[1][5]);
	nodes[4 * (size_t)self + 3] = make_float4(li, ri, 0.0f, 0.0f);
	return self;
}

} // namespace

bool wbrt_build_box_tree(const float4 *box_min, const float4 *box_max, const int2 *ranges, int B,
                         std::vector<float4> &nodes, std::vector<int> &box_of_rank, int &depth)
{
	nodes.clear();
	box_of_rank.clear();
	depth = 0;
	std::vector<BoxItem> items;
	for (int b = 0; b < B; ++b)
	{
		const float mn[3] = { box_min[b].x, box_min[b].y, box_min[b].z }, mx[3] = { box_max[b].x, box_max[b].y, box_max[b].z };
		for (int a = 0; a < 3; ++a)
			if (!std::isfinite(mn[a]) || !std::isfinite(mx[a]) || mn[a] > mx[a] || std::fabs(mn[a]) > 1e30f || std::fabs(mx[a]) > 1e30f) return false;
		if (ranges[b].y <= ranges[b].x) continue; // no triangles: hitting it changes nothing
		BoxItem it;
		for (int a = 0; a < 3; ++a) { it.mn[a] = mn[a]; it.mx[a] = mx[a]; }
		it.index = b;
		items.push_back(it);
	}
	if (items.size() < 2) return false;
	const int root = build_box_node(items, 0, (int)items.size(), nodes, box_of_rank, 0, depth);
	return root == 0;
}

// Test hook (not in the public header): the hierarchy over the boxes.  box_min/box_max: 4 floats per
// box; ranges: first,last per box; nodes_out: 16 floats per node (nodes_cap nodes), rank_out: B ints.
// Returns the node count (0: no tree), leaves in *leaves_out, inner levels in *depth_out.
extern "C" int wbrt_box_tree_build_(const float *box_min, const float *box_max, const int *ranges, int B, float *nodes_out, int nodes_cap,
                                    int *rank_out, int *leaves_out, int *depth_out)
{
	std::vector<float4> nodes;
	std::vector<int> rank;
	int depth = 0;
	if (!wbrt_build_box_tree(reinterpret_cast<const float4 *>(box_min), reinterpret_cast<const float4 *>(box_max),
	                         reinterpret_cast<const int2 *>(ranges), B, nodes, rank, depth)) return 0;
	if ((int)(nodes.size() / 4) > nodes_cap || (int)rank.size() > B) return -1;
	std::memcpy(nodes_out, nodes.data(), nodes.size() * sizeof(float4));
	std::memcpy(rank_out, rank.data(), rank.size() * sizeof(int));
	*leaves_out = (int)rank.size();
	*depth_out = depth;
	return (int)(nodes.size() / 4);
}
