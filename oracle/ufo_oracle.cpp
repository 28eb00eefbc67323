/*
 * ufo_oracle.cpp -- CPU restatement ("port") of UFOMap's scan-integration path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_abi.h): only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this.  It is NOT a fallback for the HIP path.
 *
 * Parity status: PINNED against the reference itself -- tests/test_oracle_vs_reference.py diffs
 * every function below against oracle/_ref/libufo_ref.so (the unmodified reference compiled from
 * /root/reference) on KATs, random clouds, multi-scan sequences and the synthetic LiDAR / RGB-D
 * scans, and against the golden fixtures under tests/golden/ generated from that build.  (The
 * reference ships no tests of its own: SURVEY.md section 4.)
 *
 * Written from the behaviour of the reference (paths relative to /root/reference/ufomap/include/ufo),
 * in its own structure: an index-linked node pool instead of pointer nodes, std::unordered_set
 * instead of CodeSet/CodeMap.  Arithmetic contract: SURVEY.md 8(a'): binary64, no FMA contraction
 * (built with -ffp-contract=off), occupancy in binary32.
 */
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <sstream>
#include <string>
#include <unordered_set>
#include <vector>

#include "oracle_abi.h"

namespace
{
typedef uint64_t u64;
typedef uint32_t u32;

struct V3 {
	double v[3];
	double& operator[](int i) { return v[i]; }
	double operator[](int i) const { return v[i]; }
};
inline V3 sub(V3 const& a, V3 const& b) { return V3{{a[0] - b[0], a[1] - b[1], a[2] - b[2]}}; }
inline V3 add(V3 const& a, V3 const& b) { return V3{{a[0] + b[0], a[1] + b[1], a[2] + b[2]}}; }
inline V3 mul(V3 const& a, double s) { return V3{{a[0] * s, a[1] * s, a[2] * s}}; }
inline V3 divs(V3 const& a, double s) { return V3{{a[0] / s, a[1] / s, a[2] / s}}; }
/* math/vector3.h:203-207: left-to-right sum of squares */
inline double sqnorm(V3 const& a) { return (a[0] * a[0]) + (a[1] * a[1]) + (a[2] * a[2]); }
inline double norm(V3 const& a) { return std::sqrt(sqnorm(a)); }

/* map/code.h:336-349 -- 21 bits to every third bit */
inline u64 spread3(u32 a)
{
	u64 c = (u64)a & 0x1fffff;
	c = (c | c << 32) & 0x1f00000000ffffULL;
	c = (c | c << 16) & 0x1f0000ff0000ffULL;
	c = (c | c << 8) & 0x100f00f00f00f00fULL;
	c = (c | c << 4) & 0x10c30c30c30c30c3ULL;
	c = (c | c << 2) & 0x1249249249249249ULL;
	return c;
}
/* map/code.h:183-192 */
inline u64 morton(const u32 k[3]) { return spread3(k[0]) | (spread3(k[1]) << 1) | (spread3(k[2]) << 2); }

struct Node {
	float occ;
	uint8_t rgb[3];
	bool cfree, cunk, leaf;
	int child;  // index of the first of 8 children in the pool, -1 = none allocated
};

struct Rec {
	u64 code;
	uint8_t depth;
	float occ;
	uint8_t flags;
	uint8_t rgb[3];
};
inline bool recLess(Rec const& a, Rec const& b) { return a.depth != b.depth ? a.depth < b.depth : a.code < b.code; }

struct Hit {
	u64 code;
	uint8_t rgb[3];
};
}  // namespace

struct ufo_oracle_map {
	/* geometry: map/octree.h:923-943 */
	double res, rf;
	unsigned L;
	u32 M;
	double hs[24];
	bool pruning, color;
	/* sensor model: map/occupancy_map_base.h:864-869, 909 */
	double occ_thr, free_thr, hit_log, miss_log, cmin_log, cmax_log;
	/* change AABB: occupancy_map_base.h:793-822 */
	V3 min_change, max_change;
	/* tree */
	std::vector<Node> pool;  // pool[0] = root
	std::vector<int> freelist;
	/* stage outputs of the last insert */
	std::vector<u64> last_hits, last_misses;
	std::vector<V3> last_rays;
	u64 last_steps, last_oob;
	bool runaway;  // a ray exceeded the step budget (see freeSpaceNormal)

	double size(unsigned d) const { return hs[d + 1]; }
	V3 bbxMin() const { return V3{{-hs[L], -hs[L], -hs[L]}}; }
	V3 bbxMax() const { return V3{{hs[L], hs[L], hs[L]}}; }

	/* octree.h:317-324 */
	u32 toKey1(double c, unsigned d) const
	{
		int kv = (int)std::floor(rf * c);
		if (0 == d) return (u32)kv + M;
		return (u32)(((kv >> d) << d) + (1 << (d - 1))) + M;
	}
	void toKey(V3 const& p, unsigned d, u32 k[3]) const
	{
		for (int i = 0; i < 3; ++i) k[i] = toKey1(p[i], d);
	}
	/* octree.h:374-383 */
	double toCoord1(u32 key, unsigned d) const
	{
		if (L == d) return 0.0;
		double divider = double(1 << d);
		return (std::floor((double(key) - double(M)) / divider) + 0.5) * size(d);
	}
	V3 toCoord(const u32 k[3], unsigned d) const { return V3{{toCoord1(k[0], d), toCoord1(k[1], d), toCoord1(k[2], d)}}; }

	/* occupancy_map_base.h:926-940 */
	bool isOccupied(float v) const { return occ_thr < v; }
	bool isFree(float v) const { return free_thr > v; }
	bool isUnknown(float v) const { return free_thr <= v && occ_thr >= v; }

	static bool inBBX(V3 const& p, V3 const& mn, V3 const& mx)
	{
		return mn[0] <= p[0] && mx[0] >= p[0] && mn[1] <= p[1] && mx[1] >= p[1] && mn[2] <= p[2] && mx[2] >= p[2];
	}
	/* octree.h:1316-1332: strict test on the two OTHER axes */
	static bool inBBXAxis(V3 const& p, int axis, V3 const& mn, V3 const& mx)
	{
		int a = (axis + 1) % 3, b = (axis + 2) % 3;
		return p[a] > mn[a] && p[a] < mx[a] && p[b] > mn[b] && p[b] < mx[b];
	}
	/* octree.h:1306-1314 */
	static bool intersect(double d1, double d2, V3 const& p1, V3 const& p2, V3* hit)
	{
		if (0 <= (d1 * d2)) return false;
		*hit = add(p1, mul(sub(p2, p1), (-d1 / (d2 - d1))));
		return true;
	}
	bool isInside(V3 const& p) const { return inBBX(p, bbxMin(), bbxMax()); }

	/* octree.h:1240-1295 */
	bool moveLineInside(V3& o, V3& e) const
	{
		V3 mn = bbxMin(), mx = bbxMax();
		for (int i = 0; i < 3; ++i) {
			if ((o[i] < mn[i] && e[i] < mn[i]) || (o[i] > mx[i] && e[i] > mx[i])) return false;
		}
		if (inBBX(o, mn, mx) && inBBX(e, mn, mx)) return true;
		int hits = 0;
		V3 hit[2];
		for (int i = 0; i < 3 && hits < 2; ++i) {
			if (intersect(o[i] - mn[i], e[i] - mn[i], o, e, &hit[hits]) && inBBXAxis(hit[hits], i, mn, mx)) ++hits;
		}
		for (int i = 0; i < 3 && hits < 2; ++i) {
			if (intersect(o[i] - mx[i], e[i] - mx[i], o, e, &hit[hits]) && inBBXAxis(hit[hits], i, mn, mx)) ++hits;
		}
		if (1 == hits) {
			if (inBBX(o, mn, mx)) e = hit[0];
			else o = hit[0];
		} else if (2 == hits) {
			if ((sqnorm(sub(o, hit[0])) + sqnorm(sub(e, hit[1]))) <= (sqnorm(sub(o, hit[1])) + sqnorm(sub(e, hit[0])))) {
				o = hit[0];
				e = hit[1];
			} else {
				o = hit[1];
				e = hit[0];
			}
		}
		return true;
	}

	/* ---- node pool (replaces new/delete of 8-child arrays, octree.h:1022-1086) ---- */
	int allocBlock()
	{
		int b;
		if (!freelist.empty()) {
			b = freelist.back();
			freelist.pop_back();
		} else {
			b = (int)pool.size();
			pool.resize(pool.size() + 8);
		}
		for (int i = 0; i < 8; ++i) pool[b + i].child = -1;
		return b;
	}
	/* octree.h:1022-1058: children inherit the whole value and flags of the node */
	void createChildren(int n, unsigned)
	{
		if (!pool[n].leaf) return;
		if (pool[n].child < 0) {
			int b = allocBlock();
			pool[n].child = b;
		}
		int b = pool[n].child;
		for (int i = 0; i < 8; ++i) {
			int gc = pool[b + i].child;
			pool[b + i] = pool[n];
			pool[b + i].leaf = true;
			pool[b + i].child = gc;
		}
		pool[n].leaf = false;
	}
	/* octree.h:1060-1086 */
	void deleteChildren(int n, unsigned depth, bool manual)
	{
		pool[n].leaf = true;
		if (pool[n].child < 0 || (!manual && !pruning)) return;
		int b = pool[n].child;
		if (depth > 1) {
			for (int i = 0; i < 8; ++i) deleteChildren(b + i, depth - 1, true);
		}
		freelist.push_back(b);
		pool[n].child = -1;
	}
	bool sameValue(Node const& a, Node const& b) const
	{
		if (a.occ != b.occ) return false;
		return !color || (a.rgb[0] == b.rgb[0] && a.rgb[1] == b.rgb[1] && a.rgb[2] == b.rgb[2]);
	}
	/* octree.h:1145-1162 */
	bool collapsible(int n, unsigned depth) const
	{
		int b = pool[n].child;
		if (depth > 1) {
			for (int i = 0; i < 8; ++i)
				if (!pool[b + i].leaf) return false;
		}
		for (int i = 1; i < 8; ++i)
			if (!sameValue(pool[b], pool[b + i])) return false;
		return true;
	}
	/* octree.h:997-1016; path[d] = pool index of the node at depth d */
	void createNode(u64 code, unsigned target, int* path)
	{
		path[L] = 0;
		for (unsigned d = L; d > target; --d) {
			int n = path[d];
			if (pool[n].leaf) createChildren(n, d);
			path[d - 1] = pool[n].child + (int)((code >> (3 * (d - 1))) & 7);
		}
	}

	/* occupancy_map_base.h:1139-1145 */
	bool updateOccupancy(float& cur, float upd) const
	{
		float old = cur;
		float lo = (float)cmin_log, hi = (float)cmax_log;
		float v = cur + upd;
		cur = (v < lo) ? lo : ((hi < v) ? hi : v);
		return old != cur;
	}

	/* occupancy_map_color.cpp:177-222 */
	void avgChildColor(int n, uint8_t out[3]) const
	{
		if (pool[n].leaf) {
			std::memcpy(out, pool[n].rgb, 3);
			return;
		}
		int b = pool[n].child;
		double r = 0, g = 0, bl = 0;
		int cnt = 0;
		for (int i = 0; i < 8; ++i) {
			Node const& c = pool[b + i];
			if (c.rgb[0] || c.rgb[1] || c.rgb[2]) {
				double cr = (double)c.rgb[0], cg = (double)c.rgb[1], cb = (double)c.rgb[2];
				r += cr * cr;
				g += cg * cg;
				bl += cb * cb;
				++cnt;
			}
		}
		if (0 == cnt) {
			out[0] = out[1] = out[2] = 0;
			return;
		}
		double num = (double)cnt;
		out[0] = (uint8_t)std::sqrt(r / num);
		out[1] = (uint8_t)std::sqrt(g / num);
		out[2] = (uint8_t)std::sqrt(bl / num);
	}

	/* occupancy_map_base.h:1179-1224 */
	bool updateNodeBase(int n, unsigned depth)
	{
		Node& nd = pool[n];
		if (nd.leaf) {
			bool nf = isFree(nd.occ), nu = isUnknown(nd.occ);
			bool upd = (nd.cfree != nf) || (nd.cunk != nu);
			nd.cfree = nf;
			nd.cunk = nu;
			return upd;
		}
		float nocc = std::numeric_limits<float>::lowest();
		bool nf = false, nu = false;
		int b = nd.child;
		for (int i = 0; i < 8; ++i) {
			Node const& c = pool[b + i];
			nocc = std::max(nocc, c.occ);
			if (1 == depth) {
				nf = nf || isFree(c.occ);
				nu = nu || isUnknown(c.occ);
			} else {
				nf = nf || c.cfree;
				nu = nu || c.cunk;
			}
		}
		if (collapsible(n, depth)) deleteChildren(n, depth, false);
		Node& nd2 = pool[n];
		if (nd2.occ != nocc || nd2.cfree != nf || nd2.cunk != nu) {
			nd2.occ = nocc;
			nd2.cfree = nf;
			nd2.cunk = nu;
			return true;
		}
		return false;
	}
	/* occupancy_map_color.cpp:115-122 */
	bool updateNode(int n, unsigned depth)
	{
		if (!color) return updateNodeBase(n, depth);
		uint8_t nc[3];
		avgChildColor(n, nc);
		bool changed = updateNodeBase(n, depth);
		Node& nd = pool[n];
		changed = changed || std::memcmp(nd.rgb, nc, 3) != 0;
		std::memcpy(nd.rgb, nc, 3);
		return changed;
	}
	/* occupancy_map_base.h:1126-1133 */
	void updateParents(int const* path, unsigned depth)
	{
		for (unsigned d = std::max(1u, depth); d <= L; ++d) {
			if (!updateNode(path[d], d)) return;
		}
	}
	/* occupancy_map_base.h:1085-1120 */
	bool updateAllChildren(int n, unsigned depth, float upd)
	{
		bool changed = false;
		int b = pool[n].child;
		if (1 == depth) {
			for (int i = 0; i < 8; ++i)
				if (updateOccupancy(pool[b + i].occ, upd)) changed = true;
		} else {
			for (int i = 0; i < 8; ++i) {
				if (pool[b + i].leaf) {
					if (updateOccupancy(pool[b + i].occ, upd)) {
						changed = true;
						updateNode(b + i, depth - 1);
					}
				} else if (updateAllChildren(b + i, depth - 1, upd)) {
					changed = true;
				}
			}
		}
		return changed && updateNode(n, depth);
	}
	/* occupancy_map_base.h:1063-1083 */
	void updateValue(u64 code, unsigned depth, float upd)
	{
		int path[24];
		createNode(code, depth, path);
		unsigned d = depth;
		if (0 == d || pool[path[d]].leaf) {
			updateOccupancy(pool[path[d]].occ, upd);
		} else {
			if (!updateAllChildren(path[d], d, upd)) return;
			++d;
		}
		updateParents(path, d);
	}
	/* occupancy_map_base.h:911 with LogitType=float: std::exp(float) is expf */
	static double toProb(float logit) { return 1.0 / (1.0 + std::exp(-logit)); }
	/* occupancy_map_color.cpp:142-171 */
	void updateNodeColor(Node& nd, const uint8_t upd[3], double prob) const
	{
		if (0 == std::memcmp(nd.rgb, upd, 3)) return;
		if (!(nd.rgb[0] || nd.rgb[1] || nd.rgb[2])) {
			std::memcpy(nd.rgb, upd, 3);
			return;
		}
		double total = prob + toProb(nd.occ);
		prob /= total;
		double inv = 1.0 - prob;
		for (int i = 0; i < 3; ++i) {
			double c = (double)nd.rgb[i], u = (double)upd[i];
			nd.rgb[i] = (uint8_t)std::sqrt(((c * c) * inv) + ((u * u) * prob));
		}
	}
	/* occupancy_map_color.h:269-287: colour first (with the OLD occupancy), then occupancy */
	void updateValueColor(u64 code, float upd, const uint8_t rgb[3])
	{
		int path[24];
		createNode(code, 0, path);
		updateNodeColor(pool[path[0]], rgb, toProb(upd));
		updateOccupancy(pool[path[0]].occ, upd);
		updateParents(path, 0);
	}

	/* ---- ray casting: octree.h:1192-1233, occupancy_map_base.h:1261-1339 ---- */
	bool emitMiss(std::unordered_set<u64>& set, const u32 k[3], unsigned depth)
	{
		++last_steps;
		if ((k[0] >> L) || (k[1] >> L) || (k[2] >> L)) ++last_oob;
		return set.insert(morton(k) >> (3 * depth)).second; /* try_emplace(...).second: the cell is new to this scan */
	}
	/* early_stopping (OMB:1289-1298): a ray ends once that many cells IN A ROW were in the scan's set already */
	void freeSpaceNormal(V3 const& from, V3 const& to, std::unordered_set<u64>& set, unsigned depth, unsigned early_stopping)
	{
		V3 cur = to, end = from;
		V3 dir = sub(end, cur);
		double dist = norm(dir);
		dir = divs(dir, dist);
		u32 kc[3], ke[3];
		toKey(cur, depth, kc);
		toKey(end, depth, ke);
		if (kc[0] == ke[0] && kc[1] == ke[1] && kc[2] == ke[2]) {
			emitMiss(set, kc, depth);
			return;
		}
		double node_size = size(depth), half = hs[depth];
		V3 border = sub(toCoord(kc, depth), cur);
		int step[3];
		double tdelta[3], tmax[3];
		for (int i = 0; i < 3; ++i) {
			if (0 < dir[i]) {
				step[i] = (int)(1U << depth);
				border[i] += half;
				tdelta[i] = node_size / std::abs(dir[i]);
				tmax[i] = border[i] / dir[i];
			} else if (0 > dir[i]) {
				step[i] = -(int)(1U << depth);
				border[i] -= half;
				tdelta[i] = node_size / std::abs(dir[i]);
				tmax[i] = border[i] / dir[i];
			} else {
				step[i] = 0;
				tdelta[i] = DBL_MAX;
				tmax[i] = DBL_MAX;
			}
		}
		/* Guard, NOT in the reference: when moveLineInside leaves an end point a few ulp OUTSIDE the
		 * cube, toKey wraps to ~2^32 and the reference walks ~2^31 cells (minutes, GBs).  No sane
		 * ray takes more than 3*2^L steps; beyond that the input is outside the parity contract. */
		u64 budget = 3ull * (1ull << L) + 8, taken = 0;
		unsigned already_in_row = 0;
		do {
			if (++taken > budget) {
				runaway = true;
				return;
			}
			if (emitMiss(set, kc, depth)) already_in_row = 0;
			else if (0 < early_stopping && ++already_in_row >= early_stopping) break;
			/* math/vector3.h:244-251 tie order */
			int a = (tmax[0] <= tmax[1]) ? ((tmax[0] <= tmax[2]) ? 0 : 2) : ((tmax[1] <= tmax[2]) ? 1 : 2);
			kc[a] += (u32)step[a];
			tmax[a] += tdelta[a];
		} while ((kc[0] != ke[0] || kc[1] != ke[1] || kc[2] != ke[2]) &&
		         std::min(std::min(tmax[0], tmax[1]), tmax[2]) <= dist);
	}
	void freeSpaceSimple(V3 const& from, V3 const& to, std::unordered_set<u64>& set, unsigned depth, unsigned early_stopping)
	{
		V3 cur = to, end = from;
		V3 dir = sub(end, cur);
		double dist = norm(dir);
		dir = divs(dir, dist);
		int num_steps = (int)(dist / size(depth));
		if (num_steps < 0 || (u64)num_steps > 3ull * (1ull << L) + 8) {
			runaway = true; /* same guard as freeSpaceNormal */
			return;
		}
		V3 stepv = mul(dir, size(depth));
		unsigned already_in_row = 0;
		for (int s = 0; s <= num_steps; ++s) {
			u32 k[3];
			toKey(cur, depth, k);
			if (emitMiss(set, k, depth)) already_in_row = 0;
			else if (0 < early_stopping && ++already_in_row >= early_stopping) break; /* OMB:1327-1333 */
			cur = add(cur, stepv);
		}
	}

	/* ---- the two head loops + helper: occupancy_map_base.h:270-417, 1345-1373;
	 *      colour head loop occupancy_map_color.h:177-267 ---- */
	int insert(V3 const& sensor, const double* xyz, const uint8_t* rgb, size_t n, double max_range,
	           unsigned depth, bool discrete, bool simple, unsigned early_stopping = 0)
	{
		if (rgb && !color) return -1;
		if (rgb && !discrete) return -2;
		std::vector<Hit> hits;
		std::vector<V3> rays;
		std::unordered_set<u64> seen0, seend;
		V3 mnc = bbxMax(), mxc = bbxMin();
		double sq_max = max_range * max_range;
		u64 oob_hits = 0;
		for (size_t p = 0; p < n; ++p) {
			V3 end = V3{{xyz[3 * p], xyz[3 * p + 1], xyz[3 * p + 2]}};
			if (!discrete) {
				V3 origin = sensor;
				V3 dir = sub(end, origin);
				double dist = norm(dir);
				if (!moveLineInside(origin, end)) continue;
				if (0 > max_range || dist <= max_range) {
					u32 k[3];
					toKey(end, 0, k);
					u64 c = morton(k);
					if ((k[0] >> L) || (k[1] >> L) || (k[2] >> L)) ++oob_hits;
					if (seen0.insert(c).second) hits.push_back(Hit{c, {0, 0, 0}});
				} else {
					dir = divs(dir, dist);
					end = add(origin, mul(dir, max_range));
				}
				rays.push_back(end);
				for (int i = 0; i < 3; ++i) {
					mnc[i] = std::min(mnc[i], std::min(end[i], origin[i]));
					mxc[i] = std::max(mxc[i], std::max(end[i], origin[i]));
				}
				continue;
			}
			/* discrete */
			double dsq = sqnorm(sub(end, sensor));
			if (0 > max_range || dsq < sq_max) {
				if (isInside(end)) {
					u32 k[3];
					toKey(end, 0, k);
					u64 c = morton(k);
					if (!seen0.insert(c).second) continue;
					Hit h{c, {0, 0, 0}};
					if (rgb) std::memcpy(h.rgb, rgb + 3 * p, 3);
					hits.push_back(h);
				}
			} else {
				u32 k[3];
				toKey(end, depth, k);
				V3 dir = sub(toCoord(k, depth), sensor);
				if (rgb) {
					/* occupancy_map_color.h:212-217: squared comparison, divide by sqrt */
					dsq = sqnorm(dir);
					if (0 <= max_range && dsq > sq_max) {
						dir = divs(dir, std::sqrt(dsq));
						end = add(sensor, mul(dir, max_range));
					}
				} else {
					/* occupancy_map_base.h:364-369 */
					double dist = norm(dir);
					dir = divs(dir, dist);
					if (0 <= max_range && dist > max_range) end = add(sensor, mul(dir, max_range));
				}
			}
			V3 cur = sensor;
			if (!moveLineInside(cur, end)) continue;
			u32 ek[3];
			toKey(end, depth, ek);
			if (0 < depth && !seend.insert(morton(ek)).second) continue;
			V3 ec = toCoord(ek, depth);
			rays.push_back(ec);
			u32 ck[3];
			toKey(cur, depth, ck);
			V3 cc = toCoord(ck, depth);
			double t = hs[depth];
			for (int i = 0; i < 3; ++i) {
				mnc[i] = std::min(mnc[i], std::min(ec[i] - t, cc[i] - t));
				mxc[i] = std::max(mxc[i], std::max(ec[i] + t, cc[i] + t));
			}
		}

		float hit_f = (float)hit_log;
		float miss_f = (float)(miss_log / double((2.0 * depth) + 1));

		/* free space: OMB:1229-1259.  freeSpace() is const and never reads the tree (OMB:1230-1232), so
		 * computing it before the hits are applied is equivalent to the reference's helper-thread overlap. */
		last_steps = 0;
		last_oob = oob_hits;
		runaway = false;
		std::unordered_set<u64> free_hits;
		for (V3 const& pt : rays) {
			V3 cur = sensor, end = pt;
			if (!moveLineInside(cur, end)) continue;
			if (simple) freeSpaceSimple(cur, end, free_hits, depth, early_stopping);
			else freeSpaceNormal(cur, end, free_hits, depth, early_stopping);
			if (runaway) return -4; /* nothing has been applied to the map yet */
		}

		/* hits first (helper thread joined before any miss lands: OMB:1351-1361) */
		last_hits.clear();
		for (Hit const& h : hits) {
			last_hits.push_back(h.code);
			if (rgb) updateValueColor(h.code, hit_f, h.rgb);
			else updateValue(h.code, 0, hit_f);
		}
		std::sort(last_hits.begin(), last_hits.end());

		last_rays = rays;
		last_misses.assign(free_hits.begin(), free_hits.end());
		std::sort(last_misses.begin(), last_misses.end());
		for (u64 c : last_misses) updateValue(c << (3 * depth), depth, miss_f);

		for (int i = 0; i < 3; ++i) {
			min_change[i] = std::min(min_change[i], mnc[i]);
			max_change[i] = std::max(max_change[i], mxc[i]);
		}
		return 0;
	}

	/* node payload: OccupancyNode::writeData / ColorOccupancyNode::writeData (occupancy_map_node.h:67-71, 150-153) */
	/* occupancy_map_base.h:1151-1157 (LogitType = float: the double logit is converted, then clamped) */
	bool setOccupancy(float& cur, double new_value) const
	{
		float old = cur;
		float nv = (float)new_value, lo = (float)cmin_log, hi = (float)cmax_log;
		cur = (nv < lo) ? lo : ((hi < nv) ? hi : nv);
		return old != cur;
	}
	/* geometry/collision_checks.cpp:256-264 with AABB::getMin/getMax (aabb.h:67-69): volume = (centre, half size) */
	static bool intersects(V3 const& c1, V3 const& h1, V3 const& c2, V3 const& h2)
	{
		for (int a = 0; a < 3; ++a) {
			double min1 = c1.v[a] - h1.v[a], max1 = c1.v[a] + h1.v[a], min2 = c2.v[a] - h2.v[a], max2 = c2.v[a] + h2.v[a];
			if (!(min1 <= max2) || !(min2 <= max1)) return false;
		}
		return true;
	}
	/* occupancy_map_base.h:986-1031 */
	bool setValueVolumeRecurs(V3 const& vc, V3 const& vh, double value, int n, V3 const& center, unsigned depth, unsigned min_depth)
	{
		unsigned const cd = depth - 1;
		double const chs = hs[cd];
		createChildren(n, depth);
		V3 const h{{chs, chs, chs}};
		bool changed = false;
		for (int i = 0; i < 8; ++i) {
			V3 c = center;  // octree.h:625-633
			c.v[0] += ((i & 1) ? chs : -chs);
			c.v[1] += ((i & 2) ? chs : -chs);
			c.v[2] += ((i & 4) ? chs : -chs);
			if (!intersects(vc, vh, c, h)) continue;
			int ch = pool[n].child + i;
			if (0 == cd) {
				if (setOccupancy(pool[ch].occ, value)) changed = true;
			} else if (min_depth < cd) {
				if (setValueVolumeRecurs(vc, vh, value, ch, c, cd, min_depth)) changed = true;
			} else {
				deleteChildren(ch, cd, false);
				if (setOccupancy(pool[ch].occ, value)) changed = true;
				if (updateNode(ch, cd)) changed = true;
			}
		}
		return !changed || updateNode(n, depth);
	}
	/* occupancy_map_base.h:492-518 */
	void setValueVolume(const double mn[3], const double mx[3], double occupancy_value, unsigned min_depth)
	{
		if (L < min_depth) return;
		// AABB(min, max) (aabb.h:62-65): half_size = (max - min) / 2, center = min + half_size
		V3 vh, vc;
		for (int a = 0; a < 3; ++a) {
			vh.v[a] = (mx[a] - mn[a]) / 2.0;
			vc.v[a] = mn[a] + vh.v[a];
		}
		V3 const center{{0, 0, 0}};
		double const half = hs[L];
		if (!intersects(vc, vh, center, V3{{half, half, half}})) return;
		double const logit = std::log(occupancy_value / (1.0 - occupancy_value));  // toLogit, OMB:909
		if (L == min_depth) {
			deleteChildren(0, L, false);
			setOccupancy(pool[0].occ, logit);
			updateNode(0, L);
			return;
		}
		if (setValueVolumeRecurs(vc, vh, logit, 0, center, L, min_depth)) updateNode(0, L);
	}

	void putData(std::string& out, Node const& nd) const
	{
		out.append(reinterpret_cast<const char*>(&nd.occ), 4);
		if (color) out.append(reinterpret_cast<const char*>(nd.rgb), 3);
	}
	/* writeNodesRecurs (occupancy_map_base.h:1484-1533), no bounding volume, min_depth 0 */
	void writeRecurs(std::string& out, int n, unsigned depth) const
	{
		int b = pool[n].child;
		uint8_t children = 0;
		for (int i = 0; i < 8; ++i)
			if (depth - 1 > 0 && !pool[b + i].leaf) children |= (uint8_t)(1u << i);
		out.push_back((char)children);
		for (int i = 0; i < 8; ++i) {
			if ((children >> i) & 1) {
				if (1 == depth - 1) {
					int gb = pool[b + i].child;
					for (int j = 0; j < 8; ++j) putData(out, pool[gb + j]);
				} else {
					writeRecurs(out, b + i, depth - 1);
				}
			} else {
				putData(out, pool[b + i]);
			}
		}
	}
	/* Octree::write (octree.h:833-868) + writeNodes (occupancy_map_base.h:1457-1482) */
	std::string writeStream() const
	{
		std::string data;
		if (!pool[0].leaf) {
			data.push_back((char)0xFF);
			writeRecurs(data, 0, L);
		} else {
			data.push_back((char)0);
			putData(data, pool[0]);
		}
		std::ostringstream hd;
		hd << "# UFOMap file";
		hd << "\n# (feel free to add / change comments, but leave the first line as it is!)\n#\n";
		hd << "version " << "1.0.0" << std::endl;
		hd << "id " << (color ? "occupancy_map_color" : "occupancy_map") << std::endl;
		hd << "resolution " << res << std::endl;
		hd << "depth_levels " << L << std::endl;
		hd << "compressed " << false << std::endl;
		hd << "uncompressed_data_size " << (int)data.size() << std::endl;
		hd << "data" << std::endl;
		return hd.str() + data;
	}

	void walk(int n, unsigned depth, u64 prefix, bool inc_unknown, std::vector<Rec>* leaves, std::vector<Rec>* inner) const
	{
		Node const& nd = pool[n];
		Rec r;
		r.code = prefix;
		r.depth = (uint8_t)depth;
		r.occ = nd.occ;
		r.flags = 0;
		std::memcpy(r.rgb, nd.rgb, 3);
		if (0 == depth || nd.leaf) {
			if (leaves && (inc_unknown || !isUnknown(nd.occ))) leaves->push_back(r);
			return;
		}
		if (inner) {
			r.flags = (nd.cfree ? 1 : 0) | (nd.cunk ? 2 : 0);
			inner->push_back(r);
		}
		for (int i = 0; i < 8; ++i) walk(nd.child + i, depth - 1, (prefix << 3) | (u64)i, inc_unknown, leaves, inner);
	}
};

static size_t copyOut(std::vector<Rec> const& v, uint64_t* codes, uint8_t* depths, float* logodds,
                      uint8_t* flags, uint8_t* rgb, size_t cap)
{
	size_t n = std::min(v.size(), cap);
	for (size_t i = 0; i < n; ++i) {
		if (codes) codes[i] = v[i].code;
		if (depths) depths[i] = v[i].depth;
		if (logodds) logodds[i] = v[i].occ;
		if (flags) flags[i] = v[i].flags;
		if (rgb) std::memcpy(rgb + 3 * i, v[i].rgb, 3);
	}
	return v.size();
}

extern "C" {

ufo_oracle_map* ufo_oracle_create(double resolution, unsigned depth_levels, int automatic_pruning,
                                  double occupied_thres, double free_thres, double prob_hit,
                                  double prob_miss, double clamp_min, double clamp_max, int color)
{
	if (depth_levels < 2 || depth_levels > 21) return nullptr; /* octree.h:931-935 */
	ufo_oracle_map* m = new ufo_oracle_map;
	m->res = resolution;
	m->rf = 1.0 / resolution;
	m->L = depth_levels;
	m->M = (u32)std::pow(2, depth_levels - 1);
	m->hs[0] = resolution / 2.0;
	m->hs[1] = resolution;
	for (unsigned i = 2; i <= depth_levels; ++i) m->hs[i] = m->hs[i - 1] * 2.0;
	m->pruning = 0 != automatic_pruning;
	m->color = 0 != color;
	auto logit = [](double p) { return std::log(p / (1.0 - p)); };
	m->occ_thr = logit(occupied_thres);
	m->free_thr = logit(free_thres);
	m->hit_log = logit(prob_hit);
	m->miss_log = logit(prob_miss);
	m->cmin_log = logit(clamp_min);
	m->cmax_log = logit(clamp_max);
	Node root;
	root.occ = 0;
	root.rgb[0] = root.rgb[1] = root.rgb[2] = 0;
	root.cfree = root.cunk = false;
	root.leaf = true;
	root.child = -1;
	m->pool.push_back(root);
	m->pool.resize(8); /* keep child blocks 8-aligned; entries 1..7 unused */
	m->updateNode(0, depth_levels); /* OMB:871 */
	m->min_change = m->bbxMax();
	m->max_change = m->bbxMin();
	m->last_steps = 0;
	m->last_oob = 0;
	m->runaway = false;
	return m;
}

void ufo_oracle_destroy(ufo_oracle_map* m) { delete m; }

int ufo_oracle_insert(ufo_oracle_map* m, const double origin[3], const double* xyz,
                      const uint8_t* rgb, size_t n, double max_range, unsigned depth, int discrete,
                      int simple_ray_casting, unsigned early_stopping)
{
	/* (early_stopping > 0 depends on the order of the rays: the port casts them in the reference's order, OMB:1234) */
	V3 o = V3{{origin[0], origin[1], origin[2]}};
	return m->insert(o, xyz, rgb, n, max_range, depth, 0 != discrete, 0 != simple_ray_casting, early_stopping);
}

size_t ufo_oracle_export_leaves(const ufo_oracle_map* m, int include_unknown, uint64_t* codes,
                                uint8_t* depths, float* logodds, uint8_t* rgb, size_t cap)
{
	std::vector<Rec> leaves;
	m->walk(0, m->L, 0, 0 != include_unknown, &leaves, nullptr);
	std::sort(leaves.begin(), leaves.end(), recLess);
	return copyOut(leaves, codes, depths, logodds, nullptr, rgb, cap);
}

size_t ufo_oracle_export_inner(const ufo_oracle_map* m, uint64_t* codes, uint8_t* depths,
                               float* logodds, uint8_t* flags, uint8_t* rgb, size_t cap)
{
	std::vector<Rec> inner;
	m->walk(0, m->L, 0, true, nullptr, &inner);
	std::sort(inner.begin(), inner.end(), recLess);
	return copyOut(inner, codes, depths, logodds, flags, rgb, cap);
}

size_t ufo_oracle_write(const ufo_oracle_map* m, uint8_t* buf, size_t cap)
{
	std::string const bytes = m->writeStream();
	if (buf && cap >= bytes.size()) std::memcpy(buf, bytes.data(), bytes.size());
	return bytes.size();
}

int ufo_oracle_minmax_change(const ufo_oracle_map* m, double mn[3], double mx[3])
{
	for (int i = 0; i < 3; ++i) {
		mn[i] = m->min_change[i];
		mx[i] = m->max_change[i];
	}
	return 0;
}

size_t ufo_oracle_last_hits(const ufo_oracle_map* m, uint64_t* codes, size_t cap)
{
	size_t n = std::min(m->last_hits.size(), cap);
	if (codes) std::memcpy(codes, m->last_hits.data(), n * sizeof(u64));
	return m->last_hits.size();
}
size_t ufo_oracle_last_rays(const ufo_oracle_map* m, double* ends, size_t cap)
{
	size_t n = std::min(m->last_rays.size(), cap);
	if (ends) std::memcpy(ends, m->last_rays.data(), n * 3 * sizeof(double));
	return m->last_rays.size();
}
size_t ufo_oracle_last_misses(const ufo_oracle_map* m, uint64_t* codes, size_t cap)
{
	size_t n = std::min(m->last_misses.size(), cap);
	if (codes) std::memcpy(codes, m->last_misses.data(), n * sizeof(u64));
	return m->last_misses.size();
}
uint64_t ufo_oracle_last_steps(const ufo_oracle_map* m) { return m->last_steps; }
uint64_t ufo_oracle_last_oob(const ufo_oracle_map* m) { return m->last_oob; }

void ufo_oracle_query(const ufo_oracle_map* m, const double* xyz, size_t n, unsigned depth, float* logodds, uint8_t* state)
{
	for (size_t q = 0; q < n; ++q) {
		V3 p{{xyz[3 * q], xyz[3 * q + 1], xyz[3 * q + 2]}};
		u32 k[3];
		m->toKey(p, depth, k);
		const u64 code = morton(k);  // Code(toCode(key), depth) (code.h:183-192)
		// Octree::getNode (octree.h:974-985)
		int node = 0;
		unsigned rd = depth;
		bool early = false;
		for (unsigned d = m->L - 1; d > depth; --d) {
			if (m->pool[node].leaf) {  // !hasChildren (octree.h:1134)
				rd = d + 1;
				early = true;
				break;
			}
			node = m->pool[node].child + (int)((code >> (3 * d)) & 7);  // getChildIdx (code.h:245-248)
		}
		(void)early;
		const Node& nd = m->pool[node];
		logodds[q] = nd.occ;
		uint8_t st = m->isOccupied(nd.occ) ? 1 : (m->isFree(nd.occ) ? 2 : 4);  // OMB:619-634
		const bool cfree = (0 == rd) ? m->isFree(nd.occ) : nd.cfree;        // OMB:962-968
		const bool cunk = (0 == rd) ? m->isUnknown(nd.occ) : nd.cunk;       // OMB:953-959
		st |= (cfree ? 8 : 0) | (cunk ? 16 : 0);
		state[q] = st;
	}
}

int ufo_oracle_set_value_volume(ufo_oracle_map* m, const double mn[3], const double mx[3], double occupancy_value, unsigned min_depth)
{
	m->setValueVolume(mn, mx, occupancy_value, min_depth);
	return 0;
}
void ufo_oracle_clamping_thres(const ufo_oracle_map* m, double* thres_min, double* thres_max)
{
	// toProb(LogitType) with LogitType = float: std::exp(float) (OMB:911, 742-744)
	*thres_min = ufo_oracle_map::toProb((float)m->cmin_log);
	*thres_max = ufo_oracle_map::toProb((float)m->cmax_log);
}

// Quaternion::operator* (math/quaternion.h:253-259), operands (w, x, y, z)
static void quatMul(const double a[4], const double b[4], double r[4])
{
	r[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
	r[1] = a[2] * b[3] - b[2] * a[3] + a[0] * b[1] + b[0] * a[1];
	r[2] = a[3] * b[1] - b[3] * a[1] + a[0] * b[2] + b[0] * a[2];
	r[3] = a[1] * b[2] - b[1] * a[2] + a[0] * b[3] + b[0] * a[3];
}

size_t ufo_oracle_ingest(const uint8_t* data, size_t n, uint32_t step, int off_x, int off_y, int off_z, int off_r, int off_g,
                         int off_b, const double q[4], const double t[3], double* xyz_out, uint8_t* rgb_out)
{
	size_t k = 0;
	const double qi[4] = {q[0], -q[1], -q[2], -q[3]};  // Quaternion::inversed (quaternion.h:266)
	for (size_t i = 0; i < n; ++i) {
		const uint8_t* rec = data + i * (size_t)step;
		float fx, fy, fz;
		memcpy(&fx, rec + off_x, 4);
		memcpy(&fy, rec + off_y, 4);
		memcpy(&fz, rec + off_z, 4);
		if (std::isnan(fx) || std::isnan(fy) || std::isnan(fz)) continue;  // conversions.cpp:92 / 124-125
		// Pose6::transform (pose6.h:114-125): rotation_.rotate(v) (quaternion.h:277-286: *this * v * inversed()), += translation_
		const double v[4] = {0.0, (double)fx, (double)fy, (double)fz};  // Quaternion(0, v(0), v(1), v(2)) (quaternion.h:263)
		double a[4], r[4];
		quatMul(q, v, a);
		quatMul(a, qi, r);
		xyz_out[3 * k] = r[1] + t[0];
		xyz_out[3 * k + 1] = r[2] + t[1];
		xyz_out[3 * k + 2] = r[3] + t[2];
		if (rgb_out) {
			rgb_out[3 * k] = off_r >= 0 ? rec[off_r] : 0;
			rgb_out[3 * k + 1] = off_g >= 0 ? rec[off_g] : 0;
			rgb_out[3 * k + 2] = off_b >= 0 ? rec[off_b] : 0;
		}
		++k;
	}
	return k;
}

const char* ufo_oracle_kind(void) { return "port"; }

}  // extern "C"

// ---- round 2 rows (iterators, change set, write/read with all arguments, model accessors): the REFERENCE build is the
// checker for these (oracle_abi.h); the port does not restate them.
extern "C" {
size_t ufo_oracle_iterate(const ufo_oracle_map*, const double*, const double*, int, int, int, int, unsigned, int, uint64_t*, uint8_t*, float*,
                          uint8_t*, uint8_t*, size_t) { return (size_t)-1; }
int ufo_oracle_enable_change_detection(ufo_oracle_map*, int) { return -1; }
int ufo_oracle_reset_change_detection(ufo_oracle_map*) { return -1; }
size_t ufo_oracle_changes(const ufo_oracle_map*, uint64_t*, uint8_t*, size_t) { return (size_t)-1; }
int ufo_oracle_enable_minmax_change_detection(ufo_oracle_map*, int) { return -1; }
size_t ufo_oracle_write_ex(const ufo_oracle_map*, const double*, const double*, int, unsigned, int, int, int, uint8_t*, size_t, long long*)
{
	return (size_t)-1;
}
int ufo_oracle_read(ufo_oracle_map*, const uint8_t*, size_t) { return -1; }
int ufo_oracle_read_data(ufo_oracle_map*, const uint8_t*, size_t, const double*, const double*, double, unsigned, int, int) { return -1; }
int ufo_oracle_get_sensor_model(const ufo_oracle_map*, double*) { return -1; }
int ufo_oracle_set_model_value(ufo_oracle_map*, int, double) { return -1; }
int ufo_oracle_set_occupied_free_thres(ufo_oracle_map*, double, double) { return -1; }
int ufo_oracle_clear_to(ufo_oracle_map*, double, unsigned) { return -1; }
int ufo_oracle_set_value_volume_ch(ufo_oracle_map*, const double*, const double*, double, unsigned) { return -1; }
}
