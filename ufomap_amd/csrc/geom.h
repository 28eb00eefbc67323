// geom.h -- the arithmetic contract of the integration path as device functions (gfx950).
//
// Every function states the reference lines whose *behaviour* it reproduces (paths relative to
// ufomap/include/ufo/ in the reference tree); SURVEY.md 8(a') is the normative op order:
// IEEE binary64, no FMA contraction (this translation unit is built with -ffp-contract=off),
// correctly rounded / and sqrt, occupancy arithmetic in binary32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ufo
{
typedef uint64_t u64;
typedef uint32_t u32;
typedef int32_t i32;
typedef uint8_t u8;

struct D3 {
	double x, y, z;
	__host__ __device__ double& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
	__host__ __device__ double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
__host__ __device__ inline D3 operator-(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__host__ __device__ inline D3 operator+(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__host__ __device__ inline D3 operator*(D3 a, double s) { return D3{a.x * s, a.y * s, a.z * s}; }
__host__ __device__ inline D3 operator/(D3 a, double s) { return D3{a.x / s, a.y / s, a.z / s}; }
// math/vector3.h:203-207 -- sum of squares left to right, then one sqrt
__host__ __device__ inline double sqnorm(D3 a) { return (a.x * a.x) + (a.y * a.y) + (a.z * a.z); }
__host__ __device__ inline double norm(D3 a) { return sqrt(sqnorm(a)); }

// Static description of a map's geometry and sensor model (map/octree.h:923-943,
// map/occupancy_map_base.h:864-869). Passed to kernels by value.
struct MapGeom {
	double res;      // leaf size
	double rf;       // 1.0 / res (octree.h:926)
	double hs[23];   // node half sizes, hs[d+1] = node size at depth d (octree.h:938-942)
	u32 L;           // depth levels
	u32 M;           // 2^(L-1) key offset (octree.h:928)
	double occ_thr;  // log-odds thresholds kept in double (occupancy_map_base.h:926-940)
	double free_thr;
	float hit;       // float(logit(prob_hit)) (occupancy_map_base.h:296)
	float cmin, cmax;  // float(logit(clamp)) (occupancy_map_base.h:1142-1143)
	double miss_log;   // logit(prob_miss) in double; divided by (2*depth+1) per call (OMB:311)
	double prob_hit_f;  // toProb(hit) as the colour blend sees it (occupancy_map_color.h:275)
	u32 color;
	u32 pruning;
};

__host__ __device__ inline double nodeSize(const MapGeom& g, u32 d) { return g.hs[d + 1]; }

// map/octree.h:317-324
__host__ __device__ inline u32 toKey1(const MapGeom& g, double c, u32 d)
{
	int kv = (int)floor(g.rf * c);
	if (0 == d) return (u32)kv + g.M;
	return (u32)(((kv >> d) << d) + (1 << (d - 1))) + g.M;
}
// map/octree.h:374-383
__host__ __device__ inline double toCoord1(const MapGeom& g, u32 key, u32 d)
{
	if (g.L == d) return 0.0;
	double divider = double(1 << d);
	return (floor((double(key) - double(g.M)) / divider) + 0.5) * nodeSize(g, d);
}

// map/code.h:336-349: 21 bits -> every third bit
__host__ __device__ inline u64 spread3(u32 a)
{
	u64 c = (u64)a & 0x1fffffULL;
	c = (c | c << 32) & 0x1f00000000ffffULL;
	c = (c | c << 16) & 0x1f0000ff0000ffULL;
	c = (c | c << 8) & 0x100f00f00f00f00fULL;
	c = (c | c << 4) & 0x10c30c30c30c30c3ULL;
	c = (c | c << 2) & 0x1249249249249249ULL;
	return c;
}
// inverse of spread3: every third bit -> 21 bits
__host__ __device__ inline u32 compact3(u64 c)
{
	c &= 0x1249249249249249ULL;
	c = (c | c >> 2) & 0x10c30c30c30c30c3ULL;
	c = (c | c >> 4) & 0x100f00f00f00f00fULL;
	c = (c | c >> 8) & 0x1f0000ff0000ffULL;
	c = (c | c >> 16) & 0x1f00000000ffffULL;
	c = (c | c >> 32) & 0x1fffffULL;
	return (u32)c;
}
// map/code.h:183-192: x -> bits 0,3,6.., y -> 1,4,7.., z -> 2,5,8..
__host__ __device__ inline u64 morton3(u32 x, u32 y, u32 z) { return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2); }

// map/octree.h:1299-1304 (closed box)
__host__ __device__ inline bool inBBX(D3 p, double h) { return -h <= p.x && h >= p.x && -h <= p.y && h >= p.y && -h <= p.z && h >= p.z; }
// map/octree.h:1316-1332 (strict on the two other axes)
__host__ __device__ inline bool inBBXAxis(D3 p, int axis, double h)
{
	int a = (axis + 1) % 3, b = (axis + 2) % 3;
	return p[a] > -h && p[a] < h && p[b] > -h && p[b] < h;
}
// map/octree.h:1306-1314
__host__ __device__ inline bool planeHit(double d1, double d2, D3 p1, D3 p2, D3* hit)
{
	if (0 <= (d1 * d2)) return false;
	*hit = p1 + (p2 - p1) * (-d1 / (d2 - d1));
	return true;
}
// map/octree.h:1240-1295: clip the segment to the map cube [-h, h]^3; false = fully outside
__host__ __device__ inline bool moveLineInside(const MapGeom& g, D3& o, D3& e)
{
	const double h = g.hs[g.L];
	for (int i = 0; i < 3; ++i) {
		if ((o[i] < -h && e[i] < -h) || (o[i] > h && e[i] > h)) return false;
	}
	if (inBBX(o, h) && inBBX(e, h)) return true;
	int hits = 0;
	D3 hit0{0, 0, 0}, hit1{0, 0, 0};
	for (int i = 0; i < 3 && hits < 2; ++i) {
		D3 t;
		if (planeHit(o[i] + h, e[i] + h, o, e, &t) && inBBXAxis(t, i, h)) {
			if (hits == 0) hit0 = t; else hit1 = t;
			++hits;
		}
	}
	for (int i = 0; i < 3 && hits < 2; ++i) {
		D3 t;
		if (planeHit(o[i] - h, e[i] - h, o, e, &t) && inBBXAxis(t, i, h)) {
			if (hits == 0) hit0 = t; else hit1 = t;
			++hits;
		}
	}
	if (1 == hits) {
		if (inBBX(o, h)) e = hit0; else o = hit0;
	} else if (2 == hits) {
		if ((sqnorm(o - hit0) + sqnorm(e - hit1)) <= (sqnorm(o - hit1) + sqnorm(e - hit0))) {
			o = hit0;
			e = hit1;
		} else {
			o = hit1;
			e = hit0;
		}
	}
	return true;
}

// map/occupancy_map_base.h:926-940 (thresholds in double, value promoted)
__host__ __device__ inline bool isFreeV(const MapGeom& g, float v) { return g.free_thr > (double)v; }
__host__ __device__ inline bool isUnknownV(const MapGeom& g, float v) { return g.free_thr <= (double)v && g.occ_thr >= (double)v; }
// map/occupancy_map_base.h:1139-1145: std::clamp<float>(cur + upd, min, max)
__host__ __device__ inline float clampAdd(float cur, float upd, float lo, float hi)
{
	float v = cur + upd;
	return (v < lo) ? lo : ((hi < v) ? hi : v);
}
}  // namespace ufo
