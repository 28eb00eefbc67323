// scan_kernels.h -- per-scan kernels that do NOT touch the map: classify points, de-duplicate
// hits, cast rays (3D-DDA) into the scan's dedup grid, extract the update list.
//
// Replaces (reference, ufomap/include/ufo/map/): the head loops of insertPointCloud /
// insertPointCloudDiscrete (occupancy_map_base.h:281-309, 354-399; colour: occupancy_map_color.h:
// 195-247), CodeSet `indices_` (code.h:378-566), freeSpace / freeSpaceNormal / freeSpaceSimple
// (occupancy_map_base.h:1229-1339), computeRayInit / computeRayTakeStep (octree.h:1192-1233) and
// CodeMap `free_hits` (code.h:568-785).
//
// The per-scan dedup containers become a *dense bit grid over the scan's bounding box*: one byte per
// 2x2x2 cell block (= one 8-child node block of the octree), bit i = child i. Hits go to grid H
// (depth 0), misses to grid M (depth = insert depth); when depth == 0 both share one grid geometry.
// Marking is a fire-and-forget atomicOr; "unique" falls out of the bit set; the update list is read
// off the non-zero bytes, already grouped per node block.
#pragma once
#include "table.h"

namespace ufo
{
enum : u32 {
	ERR_RUNAWAY = 1u,     // a ray exceeded 3*2^L+8 steps (clipped end outside the cube)
	ERR_GRID_OOB = 2u,    // a DDA cell fell outside the scan grid (should not happen)
	ERR_TABLE_FULL = 4u,  // node table full
	ERR_HASH_FULL = 8u,   // hit hash full (should not happen: sized 4x points)
	ERR_ENTRIES = 16u,    // update list larger than the buffer the host guessed: host retries with the exact size
	ERR_GATE = 128u,      // a stream hand-over (fast_kernels.h: k_gate) timed out: kernels serialised by a tool? the scan left the map alone
	ERR_NOT_STORED = 0x40000000u,  // (host side only: the pinned result block of a fast-path update has not been written)
	ERR_PREV = 64u,       // the integration enqueued just before this one flagged an error: this one stands back, the host re-runs both in order
	ERR_SPEC = 32u,       // the scan was launched on a grid predicted from the previous scan and does not fit it: the host repeats it
	ERR_GROW = 256u,      // volume path (vol_kernels.h): a tile found the node table's reserve used up and stood back; the host grows the table and
	                      // runs the tiles that are left
	ERR_VOL = 512u,       // volume path: a ray that needs the checked walk (clipped at the map cube); the scan takes the general path
};

// Geometry of one dedup grid: cells at depth `depth`, blocks of 2x2x2 cells.
struct Grid {
	i32 base[3];  // cell coordinate (key >> depth, as signed) of block (0,0,0); even
	i32 nb[3];    // blocks per axis
	u32 depth;
	u32 layout;  // 0: one byte per block (bit = child index); 1: one bit per cell, x fastest, rows padded to 32 cells;
	             // 2: no dense grid, a hash set of node blocks (MissSet) -- boxes beyond the scratch limit
	u64 bytes;   // layout 0: nb[0]*nb[1]*nb[2]; layout 1: rowBits/8 * 2nb[1] * 2nb[2] (= blocks incl. padding); rounded up to 16
};
// layout 1: bits per row of cells
__host__ __device__ inline u32 gridRowBits(const Grid& gr) { return (((u32)gr.nb[0] * 2u) + 31u) & ~31u; }

struct ScanCtl {
	u32 n_rays;
	u32 n_hits;
	u32 n_entries[2];  // [0] hit entries (level 1), [1] miss entries (level depth+1)
	u32 err;
	u32 n_codes;
	struct PhaseCtr {
		u32 n_new;       // blocks created / revived by this phase
		u32 wl_cnt[24];  // wl_cnt[l] = number of level-l blocks queued for propagation
	} ph[2];             // [0] hits phase, [1] misses phase (zeroed with the control block upload)
	i32 mb_min[3], mb_max[3];  // miss-grid cell bbox (cells at insert depth)
	i32 hb_min[3], hb_max[3];  // hit-grid cell bbox (depth 0)
	u64 aabb_min[3], aabb_max[3];  // order-encoded doubles: change AABB of this scan
	unsigned long long n_steps;
	u32 n_oob;  // cells dropped because their key lies outside [0, 2^L) (the reference aliases them)
	u32 used_now;  // MapRoot::used, mirrored here by k_propagate_tail so that the host reads ONE block per update
	u32 used_g_now, used_u_now;  // ... and the table's fill as its two regions count it (table.h: groups claimed, blocks of the first region)
	u32 n_hit_tiles;             // depth-3 nodes that hold a hit voxel of the scan (k_select): what the hits can need in tile groups
	u32 dl_total;      // coarse-miss phase: blocks visited so far (all levels, appended level by level)
	u32 dl_start[25];  // dl_start[l] .. dl_start[l-1] = range of the level-l blocks in the visit list
	u32 walk_scans;    // steady-state path: scans the walk applied, in the block of the walk's last scan (the host's statistics)
	unsigned long long dbg[64];  // diagnostics (ufomap_map_debug): per-level clocks of the propagation tails
};

// Per-workgroup partial results of k_classify / k_select (no atomics on shared words: thousands of
// waves updating 18 words of one cache line serialise at ~12 ns each); k_reduce_boxes folds them.
struct BoxPartial {
	i32 mb_min[3], mb_max[3];
	i32 hb_min[3], hb_max[3];
	double aabb_min[3], aabb_max[3];
};

struct Entry {
	u64 lk;    // location key of the node block
	u8 hit;    // children that received a hit (level-1 blocks only)
	u8 miss;   // children that received a miss
	u8 level;  // level of the block = depth of its children + 1
	u8 c_last;  // hits phase: child that the reference updates last (latest first-point, cloud order)
	u32 t_last; // hits phase: point index of that child's first point ("time" of the block's last update)
};

// order-preserving double <-> u64 (for atomicMin/Max on doubles)
__host__ __device__ inline u64 encD(double d)
{
	u64 b;
	memcpy(&b, &d, 8);
	return (b & 0x8000000000000000ULL) ? ~b : (b | 0x8000000000000000ULL);
}
__host__ __device__ inline double decD(u64 e)
{
	u64 b = (e & 0x8000000000000000ULL) ? (e & 0x7FFFFFFFFFFFFFFFULL) : ~e;
	double d;
	memcpy(&d, &b, 8);
	return d;
}

__device__ inline i32 waveMinI(i32 v)
{
	for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
	return v;
}
__device__ inline i32 waveMaxI(i32 v)
{
	for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
	return v;
}
__device__ inline double waveMinD(double v)
{
	for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
	return v;
}
__device__ inline double waveMaxD(double v)
{
	for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
	return v;
}

// Bounding-box updates: only issue the atomic when it would change the value (nearly never after the
// first few waves), otherwise thousands of waves serialise on six words.
__device__ inline void relaxMinI(i32* a, i32 v)
{
	if (v < __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(a, v);
}
__device__ inline void relaxMaxI(i32* a, i32 v)
{
	if (v > __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a, v);
}
__device__ inline void relaxMinU64(u64* a, u64 v)
{
	if (v < __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin((unsigned long long*)a, (unsigned long long)v);
}
__device__ inline void relaxMaxU64(u64* a, u64 v)
{
	if (v > __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax((unsigned long long*)a, (unsigned long long)v);
}

// Atomics at a chosen scope. WG = true: workgroup scope (executed in the XCD's L2, no sc1 round trip to
// the memory side) -- only legal when a single workgroup touches the word during the launch.
template <bool WG>
__device__ inline u32 aOr(u32* p, u32 v)
{
	return WG ? __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : atomicOr(p, v);
}
template <bool WG>
__device__ inline u32 aAnd(u32* p, u32 v)
{
	return WG ? __hip_atomic_fetch_and(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : atomicAnd(p, v);
}
template <bool WG>
__device__ inline u32 aAdd(u32* p, u32 v)
{
	return WG ? __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : atomicAdd(p, v);
}
template <bool WG>
__device__ inline u32 aLoad(const u32* p)
{
	return WG ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wave-aggregated append: one atomicAdd per wave instead of one per lane. A single hot counter costs
// ~12 ns per atomic (MI355X_MICROARCH.md "fanin"): 36 k appends to one word would be ~0.4 ms.
// Must be called by all active lanes of the wave at the same program point.
template <bool WG = false>
__device__ inline u32 waveAppend(u32* counter, bool pred)
{
	const u64 mask = __ballot(pred);
	if (0 == mask) return 0;
	const u32 lane = __lane_id();
	const int leader = __ffsll((unsigned long long)mask) - 1;
	u32 base = 0;
	if ((int)lane == leader) base = aAdd<WG>(counter, (u32)__popcll(mask));
	base = __shfl(base, leader);
	return base + (u32)__popcll(mask & ((1ULL << lane) - 1ULL));
}
// Workgroup-aggregated append (one atomic per workgroup); all threads of the block must call it.
__device__ inline u32 blockAppend(u32* counter, bool pred)
{
	__shared__ u32 wcnt[16];
	__shared__ u32 wbase;
	const u64 mask = __ballot(pred);
	const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
	if (0 == lane) wcnt[wave] = (u32)__popcll(mask);
	__syncthreads();
	if (0 == threadIdx.x) {
		u32 total = 0;
		for (u32 w = 0; w < nw; ++w) total += wcnt[w];
		wbase = total ? atomicAdd(counter, total) : 0u;
	}
	__syncthreads();
	u32 off = wbase;
	for (u32 w = 0; w < wave; ++w) off += wcnt[w];
	off += (u32)__popcll(mask & ((1ULL << lane) - 1ULL));
	__syncthreads();  // wcnt / wbase may be reused by the next call
	return off;
}
// Same for a per-lane count (0..n): returns the lane's first slot.
__device__ inline u32 waveAppendN(u32* counter, u32 cnt)
{
	u32 incl = cnt;
	const u32 lane = __lane_id();
	for (int o = 1; o < 64; o <<= 1) {
		u32 v = __shfl_up(incl, o);
		if ((int)lane >= o) incl += v;
	}
	u32 total = __shfl(incl, 63);
	u32 base = 0;
	if (lane == 63 && total) base = atomicAdd(counter, total);
	base = __shfl(base, 63);
	return base + incl - cnt;
}
__device__ inline void waveAddU64(unsigned long long* counter, unsigned long long v)
{
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	if (__lane_id() == 0 && v) atomicAdd(counter, v);
}

// ------------------------------------------------------------------------------------------------
// K0 classify: one thread per input point. Head loop of insertPointCloud (OMB:281-303) or
// insertPointCloudDiscrete (OMB:354-386 / OMC.h:195-233) up to, but not including, the
// first-point-wins de-duplication. Emits per point: ray end, flags, candidate hit code; inserts the
// candidate into the hit hash with atomicMin(point index) -> "first point in the voxel wins".
// ------------------------------------------------------------------------------------------------
enum : u8 { PF_HITCAND = 1, PF_CAST = 2 };

// Ingest fused into the head loop (ufomap_map_insert_pointcloud2): the cloud arrives as the
// raw records of a sensor_msgs/PointCloud2 -- float32 x, y, z (+ r, g, b bytes) at byte offsets inside records of
// `step` bytes -- and each point is converted, NaN-filtered (rosToUfo, ufomap_ros/src/conversions.cpp:98-138) and
// moved to the map frame (PointCloud::transform, point_cloud.h:157-166 -> Pose6::transform, pose6.h:114-125 ->
// Quaternion::rotate, quaternion.h:277-286) right where it is read; the float64 cloud never exists in memory.
struct Ingest {
	const uint8_t* data;  // nullptr: the cloud is an array of doubles x, y, z
	u32 step, ox, oy, oz;
	i32 orr, og, ob;      // -1: no colour fields
	double q[4];          // rotation w, x, y, z
	double t[3];
	uint8_t* rgb_out;     // compact r, g, b per point for the colour blend of the map half (nullptr: none)
};
// Quaternion::operator* (quaternion.h:253-259), operands (w, x, y, z), the reference's operation order
__device__ inline void quatMul(const double a[4], const double b[4], double r[4])
{
	r[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
	r[1] = a[2] * b[3] - b[2] * a[3] + a[0] * b[1] + b[0] * a[1];
	r[2] = a[3] * b[1] - b[3] * a[1] + a[0] * b[2] + b[0] * a[2];
	r[3] = a[1] * b[2] - b[1] * a[2] + a[0] * b[3] + b[0] * a[3];
}
// Point i of the cloud in the map frame; false for a point with a NaN coordinate (dropped by rosToUfo; on the
// plain double path the reference would feed NaN into toKey -- undefined -- so such points are skipped as well).
__device__ inline bool loadPoint(const double* __restrict__ xyz, const Ingest& ing, u32 i, D3* out)
{
	if (!ing.data) {
		const D3 p{xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]};
		*out = p;
		return !(p.x != p.x || p.y != p.y || p.z != p.z);
	}
	*out = D3{0.0, 0.0, 0.0};
	const uint8_t* rec = ing.data + (size_t)i * ing.step;
	float fx, fy, fz;
	memcpy(&fx, rec + ing.ox, 4);
	memcpy(&fy, rec + ing.oy, 4);
	memcpy(&fz, rec + ing.oz, 4);
	if (ing.rgb_out) {
		ing.rgb_out[3 * (size_t)i] = ing.orr >= 0 ? rec[ing.orr] : (uint8_t)0;
		ing.rgb_out[3 * (size_t)i + 1] = ing.og >= 0 ? rec[ing.og] : (uint8_t)0;
		ing.rgb_out[3 * (size_t)i + 2] = ing.ob >= 0 ? rec[ing.ob] : (uint8_t)0;
	}
	if (fx != fx || fy != fy || fz != fz) return false;
	const double v[4] = {0.0, (double)fx, (double)fy, (double)fz};  // Quaternion(0, v) (quaternion.h:263)
	const double qi[4] = {ing.q[0], -ing.q[1], -ing.q[2], -ing.q[3]};  // inversed() (quaternion.h:266)
	double a[4], r[4];
	quatMul(ing.q, v, a);
	quatMul(a, qi, r);
	out->x = r[1] + ing.t[0];
	out->y = r[2] + ing.t[1];
	out->z = r[3] + ing.t[2];
	return true;
}

struct HitHash {
	u64* keys;  // ~0 = empty
	u32* minidx;
	u32 mask;
};

__device__ inline u32 hitHashInsert(const HitHash& h, u64 code, u32 idx, u32* err)
{
	u32 s = hash64(code) & h.mask;
	for (u32 probe = 0; probe <= h.mask; ++probe) {
		u64 k = __hip_atomic_load(&h.keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (k == ~0ULL) {
			u64 prev = atomicCAS((unsigned long long*)&h.keys[s], ~0ULL, (unsigned long long)code);
			k = (prev == ~0ULL) ? code : prev;
		}
		if (k == code) {
			atomicMin(&h.minidx[s], idx);
			return s;
		}
		s = (s + 1) & h.mask;
	}
	atomicOr(err, ERR_HASH_FULL);
	return NONE;
}
// Insert `code` if absent; true only for the one thread whose CAS created the entry.
__device__ inline bool hitHashInsertUnique(const HitHash& h, u64 code, u32* err)
{
	u32 s = hash64(code) & h.mask;
	for (u32 probe = 0; probe <= h.mask; ++probe) {
		u64 k = __hip_atomic_load(&h.keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (k == ~0ULL) {
			u64 prev = atomicCAS((unsigned long long*)&h.keys[s], ~0ULL, (unsigned long long)code);
			if (prev == ~0ULL) return true;
			k = prev;
		}
		if (k == code) return false;
		s = (s + 1) & h.mask;
	}
	atomicOr(err, ERR_HASH_FULL);
	return false;
}
__device__ inline u32 hitHashFind(const HitHash& h, u64 code)
{
	u32 s = hash64(code) & h.mask;
	for (u32 probe = 0; probe <= h.mask; ++probe) {
		u64 k = h.keys[s];
		if (k == code) return s;
		if (k == ~0ULL) return NONE;
		s = (s + 1) & h.mask;
	}
	return NONE;
}

// Workgroup reduction of up to three (min, max) double pairs and six (min,max) int pairs into part[blockIdx.x].
// which: bit0 = aabb valid in this kernel, bit1 = hb, bit2 = mb. Called by all 256 threads.
__device__ inline void blockBoxReduce(BoxPartial* __restrict__ part, u32 which, const double amn[3], const double amx[3],
                                      const i32 hmn[3], const i32 hmx[3], const i32 mmn[3], const i32 mmx[3])
{
	__shared__ double sd[4][6];
	__shared__ i32 si[4][12];
	const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	for (int a = 0; a < 3; ++a) {
		if (which & 1) {
			double l = waveMinD(amn[a]), h = waveMaxD(amx[a]);
			if (0 == lane) {
				sd[wave][a] = l;
				sd[wave][3 + a] = h;
			}
		}
		if (which & 2) {
			i32 l = waveMinI(hmn[a]), h = waveMaxI(hmx[a]);
			if (0 == lane) {
				si[wave][a] = l;
				si[wave][3 + a] = h;
			}
		}
		if (which & 4) {
			i32 l = waveMinI(mmn[a]), h = waveMaxI(mmx[a]);
			if (0 == lane) {
				si[wave][6 + a] = l;
				si[wave][9 + a] = h;
			}
		}
	}
	__syncthreads();
	if (0 == threadIdx.x) {
		BoxPartial& p = part[blockIdx.x];
		const u32 nw = (blockDim.x + 63) >> 6;
		for (int a = 0; a < 3; ++a) {
			if (which & 1) {
				double l = sd[0][a], h = sd[0][3 + a];
				for (u32 w = 1; w < nw; ++w) {
					l = fmin(l, sd[w][a]);
					h = fmax(h, sd[w][3 + a]);
				}
				p.aabb_min[a] = l;
				p.aabb_max[a] = h;
			}
			if (which & 2) {
				i32 l = si[0][a], h = si[0][3 + a];
				for (u32 w = 1; w < nw; ++w) {
					l = min(l, si[w][a]);
					h = max(h, si[w][3 + a]);
				}
				p.hb_min[a] = l;
				p.hb_max[a] = h;
			}
			if (which & 4) {
				i32 l = si[0][6 + a], h = si[0][9 + a];
				for (u32 w = 1; w < nw; ++w) {
					l = min(l, si[w][6 + a]);
					h = max(h, si[w][9 + a]);
				}
				p.mb_min[a] = l;
				p.mb_max[a] = h;
			}
		}
	}
}

// Fold the per-workgroup partials into the control block (one workgroup).
// Fold the per-workgroup partials into the control block. Called by all 256 threads of ONE workgroup.
__device__ inline void reduceBoxes(const BoxPartial* __restrict__ part, u32 nparts, u32 aabb_from_classify,
                                   const BoxPartial* __restrict__ part_classify, ScanCtl* ctl, Grid spec, u32 use_spec)
{
	double amn[3] = {1e300, 1e300, 1e300}, amx[3] = {-1e300, -1e300, -1e300};
	i32 hmn[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hmx[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
	i32 mmn[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, mmx[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
	for (u32 i = threadIdx.x; i < nparts; i += blockDim.x) {
		const BoxPartial& p = part[i];
		const BoxPartial& q = aabb_from_classify ? part_classify[i] : part[i];
		for (int a = 0; a < 3; ++a) {
			amn[a] = fmin(amn[a], q.aabb_min[a]);
			amx[a] = fmax(amx[a], q.aabb_max[a]);
			hmn[a] = min(hmn[a], p.hb_min[a]);
			hmx[a] = max(hmx[a], p.hb_max[a]);
			mmn[a] = min(mmn[a], p.mb_min[a]);
			mmx[a] = max(mmx[a], p.mb_max[a]);
		}
	}
	__shared__ BoxPartial out;
	blockBoxReduce(&out - blockIdx.x, 7, amn, amx, hmn, hmx, mmn, mmx);
	__syncthreads();
	if (0 == threadIdx.x) {
		for (int a = 0; a < 3; ++a) {
			ctl->hb_min[a] = out.hb_min[a];
			ctl->hb_max[a] = out.hb_max[a];
			ctl->mb_min[a] = out.mb_min[a];
			ctl->mb_max[a] = out.mb_max[a];
			if (out.aabb_min[a] < 1e299) {
				ctl->aabb_min[a] = encD(out.aabb_min[a]);
				ctl->aabb_max[a] = encD(out.aabb_max[a]);
			}
			// the rest of the scan was enqueued on a grid predicted from the previous scan (no host round trip for the
			// boxes): it is valid iff the grid the host would have made (makeGrid: one block of padding, even base)
			// lies inside the predicted one
			if (use_spec && ctl->n_rays) {
				const long long lo = ((long long)out.mb_min[a] - 2) & ~1LL, hi = (long long)out.mb_max[a] + 2;
				const long long nb = (hi - lo) / 2 + 1;
				if (lo < (long long)spec.base[a] || lo + 2 * nb > (long long)spec.base[a] + 2ll * spec.nb[a]) atomicOr(&ctl->err, ERR_SPEC);
			}
		}
	}
}

__global__ __launch_bounds__(256) void k_reduce_boxes(const BoxPartial* __restrict__ part, u32 nparts, u32 aabb_from_classify,
                                                      const BoxPartial* __restrict__ part_classify, ScanCtl* ctl, Grid spec, u32 use_spec)
{
	reduceBoxes(part, nparts, aabb_from_classify, part_classify, ctl, spec, use_spec);
}


template <bool DISCRETE>
__global__ __launch_bounds__(256) void k_classify(MapGeom g, D3 sensor, const double* __restrict__ xyz, u32 n,
                                                  double max_range, u32 depth, u32 color_variant, HitHash hh,
                                                  D3* __restrict__ pt_end, u8* __restrict__ pt_flag,
                                                  u32* __restrict__ pt_slot, BoxPartial* __restrict__ part, ScanCtl* ctl,
                                                  Ingest ing)
{
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (!DISCRETE) {
		// OMB:281-303; the change AABB (OMB:305-308) is reduced per wave, so no early return here
		double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
		if (i < n) {
			D3 end;
			const bool valid = loadPoint(xyz, ing, i, &end);
			u8 flag = 0;
			u32 slot = NONE;
			D3 origin = sensor;
			D3 dir = end - origin;
			double dist = norm(dir);
			if (valid && moveLineInside(g, origin, end)) {
				if (0 > max_range || dist <= max_range) {
					u64 code = morton3(toKey1(g, end.x, 0), toKey1(g, end.y, 0), toKey1(g, end.z, 0));
					slot = hitHashInsert(hh, code, i, &ctl->err);
					flag |= PF_HITCAND;
				} else {
					dir = dir / dist;
					end = origin + (dir * max_range);
				}
				flag |= PF_CAST;
				for (int a = 0; a < 3; ++a) {
					mn[a] = fmin(end[a], origin[a]);
					mx[a] = fmax(end[a], origin[a]);
				}
				pt_end[i] = end;
			}
			pt_flag[i] = flag;
			pt_slot[i] = slot;
		}
		blockBoxReduce(part, 1, mn, mx, nullptr, nullptr, nullptr, nullptr);
		return;
	}
	if (i >= n) return;
	D3 end;
	if (!loadPoint(xyz, ing, i, &end)) {
		pt_flag[i] = 0;
		pt_slot[i] = NONE;
		return;
	}
	u8 flag = 0;
	u32 slot = NONE;
	// discrete: OMB:354-371 (colour variant OMC.h:195-219)
	double sq_max = max_range * max_range;
	double dsq = sqnorm(end - sensor);
	if (0 > max_range || dsq < sq_max) {
		if (inBBX(end, g.hs[g.L])) {
			u64 code = morton3(toKey1(g, end.x, 0), toKey1(g, end.y, 0), toKey1(g, end.z, 0));
			slot = hitHashInsert(hh, code, i, &ctl->err);
			flag |= PF_HITCAND;
		}
	} else {
		D3 c{toCoord1(g, toKey1(g, end.x, depth), depth), toCoord1(g, toKey1(g, end.y, depth), depth),
		     toCoord1(g, toKey1(g, end.z, depth), depth)};
		D3 dir = c - sensor;
		if (color_variant) {
			dsq = sqnorm(dir);
			if (0 <= max_range && dsq > sq_max) {
				dir = dir / sqrt(dsq);
				end = sensor + (dir * max_range);
			}
		} else {
			double dist = norm(dir);
			dir = dir / dist;
			if (0 <= max_range && dist > max_range) end = sensor + (dir * max_range);
		}
	}
	flag |= PF_CAST;
	pt_end[i] = end;
	pt_flag[i] = flag;
	pt_slot[i] = slot;
}

// ------------------------------------------------------------------------------------------------
// K1 select: first-point-wins (CodeSet semantics, OMB:295 / 358-360), compaction of the surviving
// rays and of the unique hits, bounding boxes of both dedup grids, change AABB (discrete mode).
// ------------------------------------------------------------------------------------------------
template <bool DISCRETE>
__global__ __launch_bounds__(256) void k_select(MapGeom g, D3 sensor, u32 n, u32 depth, HitHash hh,
                                                const D3* __restrict__ pt_end, const u8* __restrict__ pt_flag,
                                                const u32* __restrict__ pt_slot, D3* __restrict__ ray_end,
                                                u64* __restrict__ hit_code, u32* __restrict__ hit_pt,
                                                BoxPartial* __restrict__ part, ScanCtl* ctl, u32* __restrict__ blk_range,
                                                u32* __restrict__ ray_pt = nullptr, u32 es_mode = 0)
{
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	u8 flag = (i < n) ? pt_flag[i] : 0;
	bool cast = (flag & PF_CAST) != 0;
	bool winner = false;
	D3 end{0, 0, 0};
	if (i < n && flag) end = pt_end[i];
	if (flag & PF_HITCAND) {
		u32 s = pt_slot[i];
		winner = (s != NONE) && (hh.minidx[s] == i);
		if (DISCRETE && !winner) cast = false;  // OMB:358-360: dropped entirely, no ray
	}
	i32 hk[3] = {INT32_MAX, INT32_MAX, INT32_MAX};
	if (winner) {
		u32 kx = toKey1(g, end.x, 0), ky = toKey1(g, end.y, 0), kz = toKey1(g, end.z, 0);
		if ((kx >> g.L) || (ky >> g.L) || (kz >> g.L)) {
			winner = false;  // key outside [0, 2^L): dropped (see gridMark); rare, clipped rays only
			atomicAdd(&ctl->n_oob, 1u);
		}
	}
	const u32 hpos = blockAppend(&ctl->n_hits, winner);
	{
		// Distinct depth-3 nodes (tiles) among the hit voxels, counted through the same hash (keys tagged in bits 62-63; maps of
		// up to 19 levels, whose codes stay below bit 57): the tile groups the hits can need in the node table -- 3e5 hit
		// voxels of a 2 mm frame lie in 1e4 tiles, and the bound "every hit in a tile of its own" made that table 73 x too large
		bool newtile = false;
		if (winner && g.L <= 19u) {
			const u64 code = morton3(toKey1(g, end.x, 0), toKey1(g, end.y, 0), toKey1(g, end.z, 0));
			newtile = hitHashInsertUnique(hh, (code >> 9) | (3ULL << 62), &ctl->err);
		}
		(void)blockAppend(&ctl->n_hit_tiles, newtile);
	}
	if (winner) {
		u32 pos = hpos;
		u32 kx = toKey1(g, end.x, 0), ky = toKey1(g, end.y, 0), kz = toKey1(g, end.z, 0);
		hit_code[pos] = morton3(kx, ky, kz);
		hit_pt[pos] = i;
		hk[0] = (i32)kx;
		hk[1] = (i32)ky;
		hk[2] = (i32)kz;
	}
	i32 hkx[3] = {winner ? hk[0] : INT32_MIN, winner ? hk[1] : INT32_MIN, winner ? hk[2] : INT32_MIN};
	i32 ck[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, ek[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
	double amn[3] = {1e300, 1e300, 1e300}, amx[3] = {-1e300, -1e300, -1e300};
	bool has_aabb = false;
	if (cast) {
		D3 cur = sensor;
		D3 e2 = end;
		if (DISCRETE) {
			// OMB:371-398: clip, snap the ray end to the centre of its depth-`depth` cell
			if (!moveLineInside(g, cur, e2)) {
				cast = false;
			} else if (0 < depth) {
				u32 k0 = toKey1(g, e2.x, depth), k1 = toKey1(g, e2.y, depth), k2 = toKey1(g, e2.z, depth);
				// OMB:380-382: for depth > 0 one ray per depth-`depth` cell. All rays into one cell are
				// identical (same sensor, end = cell centre), so whichever thread creates the entry casts it.
				// Bit 63 keeps cell codes apart from the depth-0 hit codes sharing the table.
				if (es_mode) {
					// (early stopping: the cell's ray is its first point's, registered by k_es_raycells)
					const u32 s2 = hitHashFind(hh, morton3(k0, k1, k2) | (1ULL << 63));
					if (!(s2 != NONE && hh.minidx[s2] == i)) cast = false;
				} else if (!hitHashInsertUnique(hh, morton3(k0, k1, k2) | (1ULL << 63), &ctl->err)) cast = false;
			}
			if (cast) {
				u32 k0 = toKey1(g, e2.x, depth), k1 = toKey1(g, e2.y, depth), k2 = toKey1(g, e2.z, depth);
				D3 ec{toCoord1(g, k0, depth), toCoord1(g, k1, depth), toCoord1(g, k2, depth)};
				D3 cc{toCoord1(g, toKey1(g, cur.x, depth), depth), toCoord1(g, toKey1(g, cur.y, depth), depth),
				      toCoord1(g, toKey1(g, cur.z, depth), depth)};
				double t = g.hs[depth];
				for (int a = 0; a < 3; ++a) {
					amn[a] = fmin(ec[a] - t, cc[a] - t);
					amx[a] = fmax(ec[a] + t, cc[a] + t);
				}
				has_aabb = true;
				end = ec;
			}
		}
	}
	// freeSpace's own clip (OMB:1248) decides whether the ray is walked at all
	D3 c2 = sensor, e3 = end;
	if (cast && !moveLineInside(g, c2, e3)) cast = false;
	const u32 rpos = blockAppend(&ctl->n_rays, cast);
	{
		// where this workgroup's rays lie in the list (they are contiguous, in point order): k_cast<2> hands whole workgroups'
		// stretches to its workgroups, because rays of neighbouring points are neighbours in space
		const u32 rcount = (u32)__syncthreads_count(cast ? 1 : 0);
		if (0 == threadIdx.x) {
			blk_range[2u * blockIdx.x] = rpos;  // (thread 0's slot is the stretch's first)
			blk_range[2u * blockIdx.x + 1u] = rcount;
		}
	}
	if (cast) {
		ray_end[rpos] = end;
		if (ray_pt) ray_pt[rpos] = i;  // (the ray's rank in the cloud's order: early stopping)
		const i32 lim = (i32)((1u << (g.L - depth)) - 1u);
		for (int a = 0; a < 3; ++a) {
			// cells outside [0, 2^(L-depth)) are dropped by gridMark: keep them out of the bbox
			i32 ka = min(max((i32)(toKey1(g, e3[a], depth) >> depth), 0), lim);
			i32 kb = min(max((i32)(toKey1(g, c2[a], depth) >> depth), 0), lim);
			ck[a] = min(ka, kb);
			ek[a] = max(ka, kb);
		}
	}
	blockBoxReduce(part, DISCRETE ? 7u : 6u, amn, amx, hk, hkx, ck, ek);
	(void)has_aabb;
	// (Folding the partials here by the last workgroup to finish -- release fence, ticket, acquire fence -- was
	// measured: 512 agent-scope fences write back / invalidate the L2 so often that k_select went from 24 to 64 us
	// and the map kernels running on the other stream slowed down too. The reduction stays a launch of its own.)
}

// ------------------------------------------------------------------------------------------------
// grid marking helpers
// ------------------------------------------------------------------------------------------------
// Cells whose key lies outside the map's key range [0, 2^L) are DROPPED (counted in n_oob): they only
// arise when a segment is clipped at the map cube and its end rounds onto/over a face; the reference
// aliases such keys to the opposite side of the map (code.h:245-248 ignores the bits above 3L).
__device__ inline bool gridMark(const Grid& gr, u32* __restrict__ grid, i32 cx, i32 cy, i32 cz, u32 lim, u32* n_oob)
{
	if ((u32)cx >= lim || (u32)cy >= lim || (u32)cz >= lim) {
		atomicAdd(n_oob, 1u);
		return true;
	}
	i32 lx = cx - gr.base[0], ly = cy - gr.base[1], lz = cz - gr.base[2];
	i32 bx = lx >> 1, by = ly >> 1, bz = lz >> 1;
	if ((u32)bx >= (u32)gr.nb[0] || (u32)by >= (u32)gr.nb[1] || (u32)bz >= (u32)gr.nb[2]) return false;
	u64 idx = ((u64)bz * (u64)gr.nb[1] + (u64)by) * (u64)gr.nb[0] + (u64)bx;
	u32 bit = (u32)((lx & 1) | ((ly & 1) << 1) | ((lz & 1) << 2)) + 8u * (u32)(idx & 3);
	atomicOr(&grid[idx >> 2], 1u << bit);
	return true;
}

// Hits are few (<= N) and, at fine resolutions, scattered over a huge bounding box (C3: 3e5 hits in 1e9
// cells), so they are grouped per node block through a small open-addressing hash instead of a dense grid:
// key = code >> 3 (the level-1 node), value = mask of hit children + the cloud-order "time" of the block's
// last hit (first-point index << 3 | child; the reference applies hits in cloud order, OMB:1351-1354).
struct HitBlocks {
	u64* keys;  // ~0 = empty
	u32* mask;
	u32* time;
	u32 cap_mask;
};

__global__ __launch_bounds__(256) void k_hitmark(HitBlocks hb, const u64* __restrict__ hit_code, const u32* __restrict__ hit_pt,
                                                 const ScanCtl* ctl_in, ScanCtl* ctl)
{
	u32 n = ctl_in->n_hits;
	for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const u64 c = hit_code[i];
		const u64 key = c >> 3;
		const u32 child = (u32)(c & 7);
		u32 s = hash64(key) & hb.cap_mask;
		for (u32 probe = 0; probe <= hb.cap_mask; ++probe) {
			u64 k = __hip_atomic_load(&hb.keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (k == ~0ULL) {
				u64 prev = atomicCAS((unsigned long long*)&hb.keys[s], ~0ULL, (unsigned long long)key);
				k = (prev == ~0ULL) ? key : prev;
			}
			if (k == key) {
				atomicOr(&hb.mask[s], 1u << child);
				atomicMax(&hb.time[s], (hit_pt[i] << 3) | child);
				break;
			}
			s = (s + 1) & hb.cap_mask;
		}
	}
}

// slot of a node block in the hit-block hash, NONE when the block received no hit
__device__ inline u32 hitBlocksFind(const HitBlocks& hb, u64 key)
{
	if (!hb.keys) return 0xFFFFFFFFu;
	u32 s = hash64(key) & hb.cap_mask;
	for (u32 probe = 0; probe <= hb.cap_mask; ++probe) {
		const u64 k = hb.keys[s];
		if (k == key) return s;
		if (k == ~0ULL) return 0xFFFFFFFFu;
		s = (s + 1) & hb.cap_mask;
	}
	return 0xFFFFFFFFu;
}
#define UFO_HB_TAKEN 0x100u  // HitBlocks::mask: the block's hits already travel with its miss entry (merged list)

// hit blocks -> update list (hit entries, level 1). Each lane takes 8 slots so that a wave reserves its
// output with ONE atomic (an atomic per wave-iteration on a single counter costs ~12 ns each).
// Blocks flagged UFO_HB_TAKEN by k_extract<true> are skipped (their hits are part of a merged entry).
__global__ __launch_bounds__(256) void k_extract_hits(MapGeom g, HitBlocks hb, Entry* __restrict__ entries, u32 cap, ScanCtl* ctl)
{
	const u32 nslots = hb.cap_mask + 1;
	const u32 base = (blockIdx.x * blockDim.x + threadIdx.x) * 8u;
	u32 have = 0;
	for (u32 k = 0; k < 8u; ++k)
		if (base + k < nslots && hb.keys[base + k] != ~0ULL && !(hb.mask[base + k] & UFO_HB_TAKEN)) have |= 1u << k;
	u32 pos = waveAppendN(&ctl->n_entries[0], (u32)__popc(have));
	for (u32 k = 0; k < 8u; ++k) {
		if (!((have >> k) & 1u)) continue;
		const u32 s = base + k;
		const u32 my = pos++;
		if (my >= cap) continue;
		Entry e;
		e.lk = (1ULL << (3 * (g.L - 1))) | hb.keys[s];
		e.hit = (u8)hb.mask[s];
		e.miss = 0;
		e.level = 1;
		const u32 tv = hb.time[s];
		e.c_last = (u8)(tv & 7u);
		e.t_last = tv >> 3;
		entries[my] = e;
	}
	// this kernel is the last reader of the hash: leave it empty for the next scan (saves three memsets per scan)
	for (u32 k = 0; k < 8u; ++k) {
		const u32 s = base + k;
		if (s < nslots && hb.keys[s] != ~0ULL) {
			hb.keys[s] = ~0ULL;
			hb.mask[s] = 0u;
			hb.time[s] = 0u;
		}
	}
}

// ------------------------------------------------------------------------------------------------
// K2 dda: one lane per ray, sequential FP64 recurrence in the reference's op order
// (freeSpaceNormal OMB:1261-1301, computeRayInit OCT:1192-1225, computeRayTakeStep OCT:1227-1233,
// minElementIndex VEC3:244-251). Walks BACKWARDS from the ray end to the sensor. Each visited cell
// is one atomicOr into grid M.
// ------------------------------------------------------------------------------------------------
#define UFO_DDA_BLOCK 1024
#define UFO_DDA_FILT 32768            // filter entries (u32 tags): 128 KiB of LDS per workgroup
#define UFO_DDA_LDSGRID_MAX (144u << 10)  // largest grid kept entirely in LDS (160 KiB per CU)

// How the cells a workgroup's rays visit reach grid M. All rays of a scan converge on the sensor, so
// the same cells are produced tens of thousands of times (S/U_f = 9..28): an atomic per step
// on the global grid serialises on a few words (~1 ms for config C2). Three modes, chosen by the host:
//   DDA_LDSGRID  the whole grid fits in LDS (e.g. 16 cm / 20 m: 74 KB): every workgroup marks a private
//                LDS copy with ds_or (no global traffic in the ray loop) and ORs its non-zero words
//                into the global grid once, at the end;
//   DDA_FILTER   larger grids: a direct-mapped LDS filter with exact 32-bit tags removes the duplicates
//                produced by the 1024 rays of the workgroup, a plain (possibly stale) look at the grid
//                word removes most of the rest, then a fire-and-forget atomicOr;
//   DDA_DIRECT   grids of 2^32 cells or more: no filter tags wide enough, straight atomicOr.
enum { DDA_LDSGRID = 0, DDA_FILTER = 1, DDA_DIRECT = 2 };

template <int MODE>
__device__ inline u32 ddaMark(const Grid& gr, u32* __restrict__ grid, u32* __restrict__ lds, i32 cx, i32 cy, i32 cz, u32 lim,
                              u32* oob)
{
	if ((u32)cx >= lim || (u32)cy >= lim || (u32)cz >= lim) {
		++*oob;
		return 0;
	}
	i32 lx = cx - gr.base[0], ly = cy - gr.base[1], lz = cz - gr.base[2];
	i32 bx = lx >> 1, by = ly >> 1, bz = lz >> 1;
	if ((u32)bx >= (u32)gr.nb[0] || (u32)by >= (u32)gr.nb[1] || (u32)bz >= (u32)gr.nb[2]) return ERR_GRID_OOB;
	u32 child = (u32)((lx & 1) | ((ly & 1) << 1) | ((lz & 1) << 2));
	if (MODE == DDA_LDSGRID) {
		u32 idx = ((u32)bz * (u32)gr.nb[1] + (u32)by) * (u32)gr.nb[0] + (u32)bx;
		atomicOr(&lds[idx >> 2], 1u << (child + 8u * (idx & 3)));
		return 0;
	}
	u64 idx = ((u64)bz * (u64)gr.nb[1] + (u64)by) * (u64)gr.nb[0] + (u64)bx;
	u32 bit = 1u << (child + 8u * (u32)(idx & 3));
	u32* w = &grid[idx >> 2];
	if (MODE == DDA_FILTER) {
		u32 tag = (u32)(idx << 3) | child;
		u32 h = (tag * 0x9E3779B1u) >> 17;  // 15 bits
		if (lds[h] == tag) return 0;
		lds[h] = tag;
		if (*w & bit) return 0;
	}
	atomicOr(w, bit);
	return 0;
}

template <bool SIMPLE, int MODE>
__global__ __launch_bounds__(UFO_DDA_BLOCK) void k_dda(MapGeom g, D3 sensor, u32 depth, Grid gr, u32* __restrict__ grid,
                                                       const D3* __restrict__ ray_end, const ScanCtl* ctl_in, ScanCtl* ctl)
{
	extern __shared__ __attribute__((aligned(16))) u32 lds[];
	const u32 lds_words = (MODE == DDA_LDSGRID) ? (u32)(gr.bytes >> 2) : (MODE == DDA_FILTER ? (u32)UFO_DDA_FILT : 0u);
	const u32 lds_init = (MODE == DDA_LDSGRID) ? 0u : 0xFFFFFFFFu;  // no cell has tag 2^32-1 (host guarantees)
	if (MODE == DDA_LDSGRID) {
		uint4* l4 = reinterpret_cast<uint4*>(lds);
		for (u32 j = threadIdx.x; j < (lds_words >> 2); j += blockDim.x) l4[j] = make_uint4(0, 0, 0, 0);
	} else {
		for (u32 j = threadIdx.x; j < lds_words; j += blockDim.x) lds[j] = lds_init;
	}
	__syncthreads();
	u32 n = ctl_in->n_rays;
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	unsigned long long steps = 0;
	u32 err = 0, oob = 0;
	const u32 lim = 1u << (g.L - depth);
	if (i < n) {
		D3 from = sensor, to = ray_end[i];
		if (moveLineInside(g, from, to)) {
			// "Do it backwards" OMB:1266-1272
			D3 cur = to, end = from;
			D3 dir = end - cur;
			double dist = norm(dir);
			dir = dir / dist;
			const u64 budget = 3ull * (1ull << g.L) + 8;
			if (SIMPLE) {
				// freeSpaceSimple OMB:1303-1339
				double ns = nodeSize(g, depth);
				int num_steps = (int)(dist / ns);
				if (num_steps < 0 || (u64)num_steps > budget) {
					err |= ERR_RUNAWAY;
				} else {
					D3 stepv = dir * ns;
					for (int s = 0; s <= num_steps; ++s) {
						i32 cx = (i32)(toKey1(g, cur.x, depth) >> depth), cy = (i32)(toKey1(g, cur.y, depth) >> depth),
						    cz = (i32)(toKey1(g, cur.z, depth) >> depth);
						err |= ddaMark<MODE>(gr, grid, lds, cx, cy, cz, lim, &oob);
						++steps;
						cur = cur + stepv;
					}
				}
			} else {
				u32 kx = toKey1(g, cur.x, depth), ky = toKey1(g, cur.y, depth), kz = toKey1(g, cur.z, depth);
				u32 ex = toKey1(g, end.x, depth), ey = toKey1(g, end.y, depth), ez = toKey1(g, end.z, depth);
				if (kx == ex && ky == ey && kz == ez) {
					// OMB:1281-1284
					err |= ddaMark<MODE>(gr, grid, lds, (i32)(kx >> depth), (i32)(ky >> depth), (i32)(kz >> depth), lim, &oob);
					steps = 1;
				} else {
					// computeRayInit OCT:1204-1224
					double node_size = nodeSize(g, depth), half = g.hs[depth];
					double bx = toCoord1(g, kx, depth) - cur.x, by = toCoord1(g, ky, depth) - cur.y,
					       bz = toCoord1(g, kz, depth) - cur.z;
					i32 sx, sy, sz;
					double tdx, tdy, tdz, tmx, tmy, tmz;
#define UFO_AXIS_INIT(d, b, s, td, tm)                 \
	if (0 < d) {                                        \
		s = 1;                                          \
		b += half;                                      \
		td = node_size / fabs(d);                       \
		tm = b / d;                                     \
	} else if (0 > d) {                                 \
		s = -1;                                         \
		b -= half;                                      \
		td = node_size / fabs(d);                       \
		tm = b / d;                                     \
	} else {                                            \
		s = 0;                                          \
		td = 1.7976931348623157e308;                    \
		tm = 1.7976931348623157e308;                    \
	}
					UFO_AXIS_INIT(dir.x, bx, sx, tdx, tmx)
					UFO_AXIS_INIT(dir.y, by, sy, tdy, tmy)
					UFO_AXIS_INIT(dir.z, bz, sz, tdz, tmz)
#undef UFO_AXIS_INIT
					// cells: key >> depth (step is +-2^depth in key units = +-1 cell)
					i32 cx = (i32)(kx >> depth), cy = (i32)(ky >> depth), cz = (i32)(kz >> depth);
					const i32 gx = (i32)(ex >> depth), gy = (i32)(ey >> depth), gz = (i32)(ez >> depth);
					// One dependent instruction chain per ray: a single wave retires ~1 instruction per 8 cycles,
					// so the step body is kept short and straight-line. The walk is monotone per axis between the
					// start and the goal cell, so range checks are done once per ray: if both ends lie inside the
					// key range and (with the one-block padding) inside the grid, no step needs a check.
					const i32 l0x = cx - gr.base[0], l0y = cy - gr.base[1], l0z = cz - gr.base[2];
					const i32 l1x = gx - gr.base[0], l1y = gy - gr.base[1], l1z = gz - gr.base[2];
					const i32 mxx = 2 * gr.nb[0] - 2, mxy = 2 * gr.nb[1] - 2, mxz = 2 * gr.nb[2] - 2;
					const bool safe = (u32)cx < lim && (u32)cy < lim && (u32)cz < lim && (u32)gx < lim && (u32)gy < lim && (u32)gz < lim &&
					                  l0x >= 1 && l0y >= 1 && l0z >= 1 && l1x >= 1 && l1y >= 1 && l1z >= 1 && l0x <= mxx && l0y <= mxy &&
					                  l0z <= mxz && l1x <= mxx && l1y <= mxy && l1z <= mxz && budget < 0xFFFFFFFFull;
					if (safe && mxx < 1023 && mxy < 1023 && mxz < 1023) {
						// Packed local cell coordinates x | y<<10 | z<<20 (each < 1024: always true for LDS-sized grids):
						// one add moves the cell, one compare tests the goal, bit-field extracts feed the grid index.
						u32 pk = (u32)l0x | ((u32)l0y << 10) | ((u32)l0z << 20);
						const u32 gpk = (u32)l1x | ((u32)l1y << 10) | ((u32)l1z << 20);
						const u32 dxs = (u32)sx, dys = (u32)sy << 10, dzs = (u32)sz << 20;  // two's complement: -1 << k works field-wise
						const u32 nbx = (u32)gr.nb[0], nby = (u32)gr.nb[1];
						const u32 budget32 = (u32)budget;
						u32 cnt = 0;
						// the minimum of the three t_max serves twice: it selects the axis (lowest index among the
						// minima == VEC3:244-251) and, one step later, it is the `t_max.min() <= distance` test (OMB:1300)
						double m = (tmy < tmx) ? tmy : tmx;
						m = (tmz < m) ? tmz : m;
						bool go;
						do {
							if (++cnt > budget32) {
								err |= ERR_RUNAWAY;
								break;
							}
							const u32 bxx = (pk >> 1) & 511u, byy = (pk >> 11) & 511u, bzz = pk >> 21;
							const u32 idx = __umul24(__umul24(bzz, nby) + byy, nbx) + bxx;
							const u32 bit = (pk & 1u) | ((pk >> 9) & 2u) | ((pk >> 18) & 4u) | ((idx & 3u) << 3);
							u32 seen = 0;
							if (MODE == DDA_LDSGRID) seen = lds[idx >> 2];  // issued early, consumed after the arithmetic below
							if (MODE != DDA_LDSGRID) {
								u32* w = &grid[idx >> 2];
								bool skip = false;
								if (MODE == DDA_FILTER) {
									const u32 tag = (idx << 3) | (bit & 7u);
									const u32 h = (tag * 0x9E3779B1u) >> 17;
									skip = lds[h] == tag;
									if (!skip) {
										lds[h] = tag;
										skip = (*w >> bit) & 1u;
									}
								}
								if (!skip) atomicOr(w, 1u << bit);
							}
							const bool selx = tmx == m;
							const bool sely = !selx && (tmy == m);
							pk += selx ? dxs : (sely ? dys : dzs);
							const double nx = tmx + tdx, ny = tmy + tdy, nz = tmz + tdz;
							tmx = selx ? nx : tmx;
							tmy = sely ? ny : tmy;
							tmz = (selx || sely) ? tmz : nz;
							m = (tmy < tmx) ? tmy : tmx;  // std::min (VEC3:241)
							m = (tmz < m) ? tmz : m;
							// Test before set: near the sensor every lane of every wave wants the same few LDS words; a
							// 64-way same-address ds_or serialises (64+ cycles per wave-instruction), a same-address read
							// is a broadcast. Once a cell is marked nobody pays for it again.
							if (MODE == DDA_LDSGRID && !((seen >> bit) & 1u)) atomicOr(&lds[idx >> 2], 1u << bit);
							go = (pk != gpk) && (m <= dist);
						} while (go);
						steps = cnt > budget32 ? budget32 : cnt;
					} else if (safe) {
						// local cell coordinates; the goal test and the grid index work on them directly
						u32 lx = (u32)l0x, ly = (u32)l0y, lz = (u32)l0z;
						const u32 tx = (u32)l1x, ty = (u32)l1y, tz = (u32)l1z;
						const u32 nbx = (u32)gr.nb[0], nby = (u32)gr.nb[1];
						const u32 budget32 = (u32)budget;
						u32 cnt = 0;
						bool go;
						do {
							if (++cnt > budget32) {
								err |= ERR_RUNAWAY;
								break;
							}
							const u32 idx = ((lz >> 1) * nby + (ly >> 1)) * nbx + (lx >> 1);
							const u32 bit = (lx & 1u) | ((ly & 1u) << 1) | ((lz & 1u) << 2) | ((idx & 3u) << 3);
							if (MODE == DDA_LDSGRID) {
								atomicOr(&lds[idx >> 2], 1u << bit);
							} else {
								u32* w = &grid[idx >> 2];
								bool skip = false;
								if (MODE == DDA_FILTER) {
									const u32 tag = (idx << 3) | (bit & 7u);
									const u32 h = (tag * 0x9E3779B1u) >> 17;
									skip = lds[h] == tag;
									if (!skip) {
										lds[h] = tag;
										skip = (*w >> bit) & 1u;
									}
								}
								if (!skip) atomicOr(w, 1u << bit);
							}
							// minElementIndex VEC3:244-251: x<=y ? (x<=z ? x : z) : (y<=z ? y : z), branch-free.
							// Only the selected axis' t_max is touched (OCT:1227-1233), so the other two keep their bits.
							const bool xy = tmx <= tmy, xz = tmx <= tmz, yz = tmy <= tmz;
							const bool selx = xy && xz;
							const bool sely = !xy && yz;
							const bool selz = !(selx || sely);
							lx += selx ? (u32)sx : 0u;
							ly += sely ? (u32)sy : 0u;
							lz += selz ? (u32)sz : 0u;
							const double nx = tmx + tdx, ny = tmy + tdy, nz = tmz + tdz;
							tmx = selx ? nx : tmx;
							tmy = sely ? ny : tmy;
							tmz = selz ? nz : tmz;
							const double m1 = (tmy < tmx) ? tmy : tmx;  // std::min (VEC3:241)
							const double m2 = (tmz < m1) ? tmz : m1;
							go = (((lx ^ tx) | (ly ^ ty) | (lz ^ tz)) != 0u) && (m2 <= dist);
						} while (go);
						steps = cnt > budget32 ? budget32 : cnt;
					} else {
						// clipped or padded-out rays (rare): every step checked
						bool go;
						do {
							if (++steps > budget) {
								err |= ERR_RUNAWAY;
								break;
							}
							err |= ddaMark<MODE>(gr, grid, lds, cx, cy, cz, lim, &oob);
							if (tmx <= tmy) {
								if (tmx <= tmz) {
									cx += sx;
									tmx += tdx;
								} else {
									cz += sz;
									tmz += tdz;
								}
							} else {
								if (tmy <= tmz) {
									cy += sy;
									tmy += tdy;
								} else {
									cz += sz;
									tmz += tdz;
								}
							}
							go = (cx != gx || cy != gy || cz != gz) && (fmin(fmin(tmx, tmy), tmz) <= dist);
						} while (go);
					}
				}
			}
		}
	}
	if (MODE == DDA_LDSGRID) {
		// Hand the private copy over as this workgroup's SLAB: plain coalesced 16-byte stores, no atomics
		// (hundreds of workgroups OR-ing into the same few thousand words cost ~0.1 us per atomic in
		// aggregate). k_merge_slabs ORs the slabs into grid M.
		__syncthreads();
		const uint4* l4 = reinterpret_cast<const uint4*>(lds);
		uint4* out4 = reinterpret_cast<uint4*>(grid) + (size_t)blockIdx.x * (lds_words >> 2);
		const u32 n4 = lds_words >> 2;  // the host rounds grid.bytes to 16
		for (u32 j = threadIdx.x; j < n4; j += blockDim.x) out4[j] = l4[j];
	}
	// total step count (diagnostic; drives the algorithmic-bytes figure of bench.py)
	waveAddU64(&ctl->n_steps, steps);
	if (oob) atomicAdd(&ctl->n_oob, oob);
	if (err) atomicOr(&ctl->err, err);
}

// ------------------------------------------------------------------------------------------------
// K2' dda, segmented: several lanes per ray.
//
// The lane-per-ray kernel above is bound by the latency of ONE ray's dependent chain (~180 steps x ~350
// cycles); occupancy does not help. But the traversal is a 3-way merge of three independent sequences
// T_a[k] = t_max_a + k * t_delta_a (each built by REPEATED rounding additions, OCT:1232), popped smallest
// first with ties to the lower axis (VEC3:244-251). Element B[m] of axis b is popped before element A[k]
// of axis a iff B[m] < A[k], or B[m] == A[k] and b < a -- independent of the third axis. So the state
// right after the k0-th pop of the ray's dominant axis a* can be rebuilt exactly without walking there:
// k0 additions give A[k0-1] and A[k0]; for each other axis a short add/compare loop counts the elements
// that precede A[k0-1] and leaves that axis' t_max. Each of the 2 or 4 lanes of a ray rebuilds the start
// of its segment (boundaries at equal counts of a*-pops), checks the loop condition there (OMB:1300; t_max
// minima are monotone, and the goal cell cannot be reached before the last segment) and walks only its
// share with the same branch-free step. Bit-identical cells, 2-4 x shorter dependent chain. (More lanes
// per ray stop paying: measured 8 lanes/ray is throughput-bound and slower than 4.)
// Rays that are not "safe" (clipped at the map cube) are walked by the group's first lane alone.
// ------------------------------------------------------------------------------------------------
#define UFO_SEG_MAX 4  // lanes per ray: 1, 2 or 4, chosen by the host so that the launch is ~4k waves

// Initial state of one ray's walk, written by k_ray_setup (one lane per ray, dense) and read by the
// lanes that walk it.
struct RayState {
	double tm[3], td[3], dist;
	u32 pk0, gpk;   // packed local start / goal cell (x | y<<10 | z<<20)
	i32 start[3];   // unpacked cells for the checked sequential walk
	i32 goal[3];
	int8_t s[3];    // step per axis (-1, 0, +1)
	uint8_t status; // 0 nothing to do, 1 single cell, 2 safe: segmented walk, 3 clipped: checked sequential walk
};

// clip (OMB:1248), keys, computeRayInit (OCT:1192-1225): everything of a ray that is not the walk itself
__device__ inline void raySetup(const MapGeom& g, const D3& sensor, u32 depth, const Grid& gr, D3 to, RayState& r)
{
	r.status = 0;
	const u32 lim = 1u << (g.L - depth);
	D3 from = sensor;
	if (moveLineInside(g, from, to)) {
		D3 cur = to, end = from;  // "do it backwards" OMB:1266-1272
		D3 dir = end - cur;
		const double dist = norm(dir);
		dir = dir / dist;
		r.dist = dist;
		const u32 kx = toKey1(g, cur.x, depth), ky = toKey1(g, cur.y, depth), kz = toKey1(g, cur.z, depth);
		const u32 ex = toKey1(g, end.x, depth), ey = toKey1(g, end.y, depth), ez = toKey1(g, end.z, depth);
		const i32 cx = (i32)(kx >> depth), cy = (i32)(ky >> depth), cz = (i32)(kz >> depth);
		const i32 gx = (i32)(ex >> depth), gy = (i32)(ey >> depth), gz = (i32)(ez >> depth);
		r.start[0] = cx;
		r.start[1] = cy;
		r.start[2] = cz;
		r.goal[0] = gx;
		r.goal[1] = gy;
		r.goal[2] = gz;
		if (kx == ex && ky == ey && kz == ez) {
			r.status = 1;  // OMB:1281-1284
		} else {
			const double node_size = nodeSize(g, depth), half = g.hs[depth];
			double bx = toCoord1(g, kx, depth) - cur.x, by = toCoord1(g, ky, depth) - cur.y, bz = toCoord1(g, kz, depth) - cur.z;
			i32 sx, sy, sz;
			double tdx, tdy, tdz, tmx, tmy, tmz;
#define UFO_AXIS_INIT(d, b, s, td, tm)                 \
	if (0 < d) {                                        \
		s = 1;                                          \
		b += half;                                      \
		td = node_size / fabs(d);                       \
		tm = b / d;                                     \
	} else if (0 > d) {                                 \
		s = -1;                                         \
		b -= half;                                      \
		td = node_size / fabs(d);                       \
		tm = b / d;                                     \
	} else {                                            \
		s = 0;                                          \
		td = 1.7976931348623157e308;                    \
		tm = 1.7976931348623157e308;                    \
	}
			UFO_AXIS_INIT(dir.x, bx, sx, tdx, tmx)
			UFO_AXIS_INIT(dir.y, by, sy, tdy, tmy)
			UFO_AXIS_INIT(dir.z, bz, sz, tdz, tmz)
#undef UFO_AXIS_INIT
			r.s[0] = (int8_t)sx;
			r.s[1] = (int8_t)sy;
			r.s[2] = (int8_t)sz;
			r.td[0] = tdx;
			r.td[1] = tdy;
			r.td[2] = tdz;
			r.tm[0] = tmx;
			r.tm[1] = tmy;
			r.tm[2] = tmz;
			const i32 l0x = cx - gr.base[0], l0y = cy - gr.base[1], l0z = cz - gr.base[2];
			const i32 l1x = gx - gr.base[0], l1y = gy - gr.base[1], l1z = gz - gr.base[2];
			const i32 mxx = 2 * gr.nb[0] - 2, mxy = 2 * gr.nb[1] - 2, mxz = 2 * gr.nb[2] - 2;
			const bool safe = (u32)cx < lim && (u32)cy < lim && (u32)cz < lim && (u32)gx < lim && (u32)gy < lim && (u32)gz < lim &&
			                  l0x >= 1 && l0y >= 1 && l0z >= 1 && l1x >= 1 && l1y >= 1 && l1z >= 1 && l0x <= mxx && l0y <= mxy &&
			                  l0z <= mxz && l1x <= mxx && l1y <= mxy && l1z <= mxz;
			if (safe) {
				r.pk0 = (u32)l0x | ((u32)l0y << 10) | ((u32)l0z << 20);
				r.gpk = (u32)l1x | ((u32)l1y << 10) | ((u32)l1z << 20);
				r.status = 2;
			} else {
				r.status = 3;
			}
		}
	}
}

__global__ __launch_bounds__(256) void k_ray_setup(MapGeom g, D3 sensor, u32 depth, Grid gr, const D3* __restrict__ ray_end,
                                                   RayState* __restrict__ rs, const ScanCtl* ctl_in)
{
	const u32 n = ctl_in->n_rays;
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	RayState r;
	raySetup(g, sensor, depth, gr, ray_end[i], r);
	rs[i] = r;
}

template <int MODE>
__global__ __launch_bounds__(UFO_DDA_BLOCK) void k_dda_seg(MapGeom g, u32 depth, Grid gr, u32* __restrict__ grid,
                                                           const RayState* __restrict__ rs, u32 seg_shift, const ScanCtl* ctl_in,
                                                           ScanCtl* ctl)
{
	extern __shared__ __attribute__((aligned(16))) u32 lds[];
	const u32 lds_words = (MODE == DDA_LDSGRID) ? (u32)(gr.bytes >> 2) : (MODE == DDA_FILTER ? (u32)UFO_DDA_FILT : 0u);
	if (MODE == DDA_LDSGRID) {
		uint4* l4 = reinterpret_cast<uint4*>(lds);
		for (u32 j = threadIdx.x; j < (lds_words >> 2); j += blockDim.x) l4[j] = make_uint4(0, 0, 0, 0);
	} else {
		for (u32 j = threadIdx.x; j < lds_words; j += blockDim.x) lds[j] = 0xFFFFFFFFu;
	}
	__syncthreads();
	const u32 n = ctl_in->n_rays;
	const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
	const u32 nseg = 1u << seg_shift;
	const u32 ray = tid >> seg_shift;
	const u32 sg = tid & (nseg - 1u);
	unsigned long long steps = 0;
	u32 err = 0, oob = 0;
	const u32 lim = 1u << (g.L - depth);
	if (ray < n) {
		const RayState* r = rs + ray;
		const u32 status = r->status;
		if (0 == sg && 1 == status) {
			err |= ddaMark<MODE>(gr, grid, lds, r->start[0], r->start[1], r->start[2], lim, &oob);
			steps = 1;
		} else if (0 == sg && 3 == status) {
			// clipped ray (rare): sequential walk with every step checked, by this lane alone
			i32 cx = r->start[0], cy = r->start[1], cz = r->start[2];
			const i32 gx = r->goal[0], gy = r->goal[1], gz = r->goal[2];
			const i32 sx = r->s[0], sy = r->s[1], sz = r->s[2];
			double tmx = r->tm[0], tmy = r->tm[1], tmz = r->tm[2];
			const double tdx = r->td[0], tdy = r->td[1], tdz = r->td[2], dist = r->dist;
			const u64 budget = 3ull * (1ull << g.L) + 8;
			bool go;
			do {
				if (++steps > budget) {
					err |= ERR_RUNAWAY;
					break;
				}
				err |= ddaMark<MODE>(gr, grid, lds, cx, cy, cz, lim, &oob);
				if (tmx <= tmy) {
					if (tmx <= tmz) {
						cx += sx;
						tmx += tdx;
					} else {
						cz += sz;
						tmz += tdz;
					}
				} else {
					if (tmy <= tmz) {
						cy += sy;
						tmy += tdy;
					} else {
						cz += sz;
						tmz += tdz;
					}
				}
				go = (cx != gx || cy != gy || cz != gz) && (fmin(fmin(tmx, tmy), tmz) <= dist);
			} while (go);
		} else if (2 == status) {
			const u32 pk0 = r->pk0, gpk = r->gpk;
			const i32 sx = r->s[0], sy = r->s[1], sz = r->s[2];
			double tmx = r->tm[0], tmy = r->tm[1], tmz = r->tm[2];
			const double tdx = r->td[0], tdy = r->td[1], tdz = r->td[2], dist = r->dist;
			// dominant axis a* and this lane's segment [k0, k1) in pops of a*
			const u32 dxn = (u32)abs((i32)(gpk & 1023u) - (i32)(pk0 & 1023u));
			const u32 dyn = (u32)abs((i32)((gpk >> 10) & 1023u) - (i32)((pk0 >> 10) & 1023u));
			const u32 dzn = (u32)abs((i32)(gpk >> 20) - (i32)(pk0 >> 20));
			const u32 ax = (dxn >= dyn && dxn >= dzn) ? 0u : (dyn >= dzn ? 1u : 2u);
			const u32 dmax = ax == 0 ? dxn : (ax == 1 ? dyn : dzn);
			const u32 w = (dmax + nseg - 1) >> seg_shift;  // >= 1 (the cells differ)
			const u32 k0 = sg * w;
			const bool active = (0 == sg) || (k0 < dmax);
			const u32 k1 = (k0 + w < dmax) ? (k0 + w) : 0xFFFFFFFFu;  // the last active segment runs to the end
			if (active) {
				const u32 dxs = (u32)sx, dys = (u32)sy << 10, dzs = (u32)sz << 20;
				u32 pk = pk0;
				if (sg > 0) {
					// state right after the k0-th pop of axis a*: element A[k0-1] was popped, t_max_a* = A[k0]
					double ta = ax == 0 ? tmx : (ax == 1 ? tmy : tmz);
					const double tda = ax == 0 ? tdx : (ax == 1 ? tdy : tdz);
					double v = ta;
					for (u32 i = 0; i < k0; ++i) {
						v = ta;
						ta = ta + tda;
					}
					// other axes: elements popped before A[k0-1] (strictly smaller, or equal when the axis has priority)
					u32 cb0 = 0, cb1 = 0;
					const u32 b0 = ax == 0 ? 1u : 0u, b1 = ax == 2 ? 1u : 2u;  // the two other axes, ascending
					double t0 = b0 == 0 ? tmx : tmy, d0 = b0 == 0 ? tdx : tdy;
					double t1 = b1 == 1 ? tmy : tmz, d1 = b1 == 1 ? tdy : tdz;
					const bool p0 = b0 < ax, p1 = b1 < ax;  // lower axis index wins ties (VEC3:244-251)
					while (cb0 < 2048u && (p0 ? (t0 <= v) : (t0 < v))) {
						t0 = t0 + d0;
						++cb0;
					}
					while (cb1 < 2048u && (p1 ? (t1 <= v) : (t1 < v))) {
						t1 = t1 + d1;
						++cb1;
					}
					if (cb0 >= 2048u || cb1 >= 2048u) err |= ERR_RUNAWAY;  // (cannot trip: < 1024 cells per axis)
					const u32 da = ax == 0 ? dxs : (ax == 1 ? dys : dzs);
					const u32 db0 = b0 == 0 ? dxs : dys, db1 = b1 == 1 ? dys : dzs;
					pk = pk0 + k0 * da + cb0 * db0 + cb1 * db1;
					if (ax == 0) {
						tmx = ta;
						tmy = t0;
						tmz = t1;
					} else if (ax == 1) {
						tmy = ta;
						tmx = t0;
						tmz = t1;
					} else {
						tmz = ta;
						tmx = t0;
						tmy = t1;
					}
				}
				double m = (tmy < tmx) ? tmy : tmx;
				m = (tmz < m) ? tmz : m;
				// loop condition at the segment start (OMB:1300); segment 0 starts the do-while unconditionally
				bool go = (0 == sg) || ((pk != gpk) && (m <= dist));
				const u32 nbx = (u32)gr.nb[0], nby = (u32)gr.nb[1];
				u32 ka = k0, cnt = 0;
				while (go) {
					++cnt;
					const u32 bxx = (pk >> 1) & 511u, byy = (pk >> 11) & 511u, bzz = pk >> 21;
					const u32 idx = __umul24(__umul24(bzz, nby) + byy, nbx) + bxx;
					const u32 bit = (pk & 1u) | ((pk >> 9) & 2u) | ((pk >> 18) & 4u) | ((idx & 3u) << 3);
					if (MODE == DDA_LDSGRID) {
						atomicOr(&lds[idx >> 2], 1u << bit);
					} else {
						u32* wd = &grid[idx >> 2];
						bool skip = false;
						if (MODE == DDA_FILTER) {
							const u32 tag = (idx << 3) | (bit & 7u);
							const u32 h = (tag * 0x9E3779B1u) >> 17;
							skip = lds[h] == tag;
							if (!skip) {
								lds[h] = tag;
								skip = (*wd >> bit) & 1u;
							}
						}
						if (!skip) atomicOr(wd, 1u << bit);
					}
					const bool selx = tmx == m;
					const bool sely = !selx && (tmy == m);
					const bool selz = !(selx || sely);
					pk += selx ? dxs : (sely ? dys : dzs);
					const double nx = tmx + tdx, ny = tmy + tdy, nz = tmz + tdz;
					tmx = selx ? nx : tmx;
					tmy = sely ? ny : tmy;
					tmz = selz ? nz : tmz;
					m = (tmy < tmx) ? tmy : tmx;
					m = (tmz < m) ? tmz : m;
					const bool sela = ax == 0 ? selx : (ax == 1 ? sely : selz);
					ka += sela ? 1u : 0u;
					go = (pk != gpk) && (m <= dist) && (ka != k1) && (cnt < 4096u);
				}
				if (cnt >= 4096u) err |= ERR_RUNAWAY;  // (a packed grid has < 1024 cells per axis: a ray has < 3072 steps)
				steps += cnt;
			}
		}
	}
	if (MODE == DDA_LDSGRID) {
		__syncthreads();
		const uint4* l4 = reinterpret_cast<const uint4*>(lds);
		uint4* out4 = reinterpret_cast<uint4*>(grid) + (size_t)blockIdx.x * (lds_words >> 2);
		const u32 n4 = lds_words >> 2;
		for (u32 j = threadIdx.x; j < n4; j += blockDim.x) out4[j] = l4[j];
	}
	waveAddU64(&ctl->n_steps, steps);
	if (oob) atomicAdd(&ctl->n_oob, oob);
	if (err) atomicOr(&ctl->err, err);
}

// Sum `steps` over the workgroup and store it as this workgroup's partial (k_merge_slabs adds them up).
__device__ inline void blockStoreSteps(unsigned long long steps, unsigned long long* __restrict__ steps_part)
{
	__shared__ unsigned long long acc;
	if (0 == threadIdx.x) acc = 0;
	__syncthreads();
	for (int o = 32; o > 0; o >>= 1) steps += __shfl_xor(steps, o);
	if (0 == (threadIdx.x & 63u) && steps) atomicAdd(&acc, steps);
	__syncthreads();
	if (0 == threadIdx.x) steps_part[blockIdx.x] = acc;
}

// ------------------------------------------------------------------------------------------------
// K2w: the ray walk for scans whose cell grid fits in LDS as ONE BIT PER CELL (Grid::layout 1) -- the fast
// path of every LiDAR-sized scan. Same rays, same cells, same order as k_dda_seg (see there for the segment
// construction); what changes is the cost of a step. A lone wave per SIMD issues one VALU instruction every
// ~4-8 cycles whatever it waits for, so a ray's time is (steps) x (instructions per step), and the step is
// cut from ~65 to ~30 instructions:
//   * the cell is ONE linear bit index lin (x + rowBits*(y + ny*z)); a step adds one of three strides, the
//     mark is ds_or(lds[lin>>5], 1<<(lin&31)) -- no unpacking, no block/child arithmetic;
//   * the axis choice is three independent compares (x if tx<=ty && tx<=tz, else y if ty<=tz, else z:
//     VEC3:244-251) instead of a min3 followed by equality tests, and the loop test min(t) <= dist
//     (OMB:1300) is "any t <= dist": no dependent min chain;
//   * a segment ends where the next lane's segment starts (a DDA path never revisits a cell), so counting
//     the pops of the dominant axis disappears from the loop: one integer compare serves goal and hand-over.
// ------------------------------------------------------------------------------------------------
__device__ inline u32 pkToLin(u32 pk, u32 rowBits, u32 planeBits)
{
	return (pk & 1023u) + ((pk >> 10) & 1023u) * rowBits + (pk >> 20) * planeBits;
}
// checked mark of one cell (single-cell rays, clipped rays)
__device__ inline u32 markBitChecked(const Grid& gr, u32* __restrict__ lds, u32 rowBits, u32 planeBits, i32 cx, i32 cy, i32 cz, u32 lim,
                                     u32* oob)
{
	if ((u32)cx >= lim || (u32)cy >= lim || (u32)cz >= lim) {
		++*oob;
		return 0;
	}
	const i32 lx = cx - gr.base[0], ly = cy - gr.base[1], lz = cz - gr.base[2];
	if ((u32)lx >= 2u * (u32)gr.nb[0] || (u32)ly >= 2u * (u32)gr.nb[1] || (u32)lz >= 2u * (u32)gr.nb[2]) return ERR_GRID_OOB;
	const u32 lin = (u32)lx + (u32)ly * rowBits + (u32)lz * planeBits;
	atomicOr(&lds[lin >> 5], 1u << (lin & 31u));
	return 0;
}

__global__ __launch_bounds__(UFO_DDA_BLOCK) void k_walk(MapGeom g, u32 depth, Grid gr, u32* __restrict__ slabs,
                                                        const RayState* __restrict__ rs, u32 seg_shift, const ScanCtl* ctl_in,
                                                        ScanCtl* ctl, unsigned long long* __restrict__ steps_part)
{
	extern __shared__ __attribute__((aligned(16))) u32 lds[];
	const u32 lds_words = (u32)(gr.bytes >> 2);
	{
		uint4* l4 = reinterpret_cast<uint4*>(lds);
		for (u32 j = threadIdx.x; j < (lds_words >> 2); j += blockDim.x) l4[j] = make_uint4(0, 0, 0, 0);
	}
	__syncthreads();
	const u32 rowBits = gridRowBits(gr), planeBits = rowBits * 2u * (u32)gr.nb[1];
	const u32 n = ctl_in->n_rays;
	const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
	const u32 nseg = 1u << seg_shift;
	const u32 ray = tid >> seg_shift;
	const u32 sg = tid & (nseg - 1u);
	unsigned long long steps = 0;
	u32 err = 0, oob = 0;
	const u32 lim = 1u << (g.L - depth);
	// ---- per-lane preparation (divergent) ----
	bool go = false;
	u32 lin = 0, glin = 0xFFFFFFFFu;
	i32 dlx = 0, dly = 0, dlz = 0;
	double tmx = 0, tmy = 0, tmz = 0, tdx = 0, tdy = 0, tdz = 0, dist = 0;
	if (ray < n) {
		const RayState* r = rs + ray;
		const u32 status = r->status;
		if (0 == sg && 1 == status) {
			err |= markBitChecked(gr, lds, rowBits, planeBits, r->start[0], r->start[1], r->start[2], lim, &oob);
			steps = 1;
		} else if (0 == sg && 3 == status) {
			// clipped ray (rare): sequential walk with every step checked, by this lane alone
			i32 cx = r->start[0], cy = r->start[1], cz = r->start[2];
			const i32 gx = r->goal[0], gy = r->goal[1], gz = r->goal[2];
			const i32 sx = r->s[0], sy = r->s[1], sz = r->s[2];
			double ax_ = r->tm[0], ay_ = r->tm[1], az_ = r->tm[2];
			const double bx_ = r->td[0], by_ = r->td[1], bz_ = r->td[2], dd = r->dist;
			const u64 budget = 3ull * (1ull << g.L) + 8;
			bool more;
			do {
				if (++steps > budget) {
					err |= ERR_RUNAWAY;
					break;
				}
				err |= markBitChecked(gr, lds, rowBits, planeBits, cx, cy, cz, lim, &oob);
				if (ax_ <= ay_) {
					if (ax_ <= az_) {
						cx += sx;
						ax_ += bx_;
					} else {
						cz += sz;
						az_ += bz_;
					}
				} else {
					if (ay_ <= az_) {
						cy += sy;
						ay_ += by_;
					} else {
						cz += sz;
						az_ += bz_;
					}
				}
				more = (cx != gx || cy != gy || cz != gz) && (fmin(fmin(ax_, ay_), az_) <= dd);
			} while (more);
		} else if (2 == status) {
			const u32 pk0 = r->pk0, gpk = r->gpk;
			const i32 sx = r->s[0], sy = r->s[1], sz = r->s[2];
			tmx = r->tm[0];
			tmy = r->tm[1];
			tmz = r->tm[2];
			tdx = r->td[0];
			tdy = r->td[1];
			tdz = r->td[2];
			dist = r->dist;
			dlx = sx;
			dly = sy * (i32)rowBits;
			dlz = sz * (i32)planeBits;
			glin = pkToLin(gpk, rowBits, planeBits);
			// dominant axis a* and this lane's segment start k0 in pops of a* (k_dda_seg)
			const u32 dxn = (u32)abs((i32)(gpk & 1023u) - (i32)(pk0 & 1023u));
			const u32 dyn = (u32)abs((i32)((gpk >> 10) & 1023u) - (i32)((pk0 >> 10) & 1023u));
			const u32 dzn = (u32)abs((i32)(gpk >> 20) - (i32)(pk0 >> 20));
			const u32 ax = (dxn >= dyn && dxn >= dzn) ? 0u : (dyn >= dzn ? 1u : 2u);
			const u32 dmax = ax == 0 ? dxn : (ax == 1 ? dyn : dzn);
			const u32 w = (dmax + nseg - 1) >> seg_shift;  // >= 1 (the cells differ)
			const u32 k0 = sg * w;
			const bool active = (0 == sg) || (k0 < dmax);
			if (active) {
				lin = pkToLin(pk0, rowBits, planeBits);
				if (sg > 0) {
					// state right after the k0-th pop of axis a*: element A[k0-1] was popped, t_max_a* = A[k0]
					double ta = ax == 0 ? tmx : (ax == 1 ? tmy : tmz);
					const double tda = ax == 0 ? tdx : (ax == 1 ? tdy : tdz);
					double v = ta;
					for (u32 i = 0; i < k0; ++i) {
						v = ta;
						ta = ta + tda;
					}
					// other axes: elements popped before A[k0-1] (strictly smaller, or equal when the axis has priority)
					u32 cb0 = 0, cb1 = 0;
					const u32 b0 = ax == 0 ? 1u : 0u, b1 = ax == 2 ? 1u : 2u;  // the two other axes, ascending
					double t0 = b0 == 0 ? tmx : tmy, d0 = b0 == 0 ? tdx : tdy;
					double t1 = b1 == 1 ? tmy : tmz, d1 = b1 == 1 ? tdy : tdz;
					const bool p0 = b0 < ax, p1 = b1 < ax;  // lower axis index wins ties (VEC3:244-251)
					while (cb0 < 2048u && (p0 ? (t0 <= v) : (t0 < v))) {
						t0 = t0 + d0;
						++cb0;
					}
					while (cb1 < 2048u && (p1 ? (t1 <= v) : (t1 < v))) {
						t1 = t1 + d1;
						++cb1;
					}
					if (cb0 >= 2048u || cb1 >= 2048u) err |= ERR_RUNAWAY;  // (cannot trip: < 1024 cells per axis)
					const i32 da = ax == 0 ? dlx : (ax == 1 ? dly : dlz);
					const i32 db0 = b0 == 0 ? dlx : dly, db1 = b1 == 1 ? dly : dlz;
					lin = lin + (u32)((i32)k0 * da + (i32)cb0 * db0 + (i32)cb1 * db1);
					if (ax == 0) {
						tmx = ta;
						tmy = t0;
						tmz = t1;
					} else if (ax == 1) {
						tmy = ta;
						tmx = t0;
						tmz = t1;
					} else {
						tmz = ta;
						tmx = t0;
						tmy = t1;
					}
				}
				// loop condition at the segment start (OMB:1300); segment 0 starts the do-while unconditionally
				go = (0 == sg) || ((lin != glin) && (tmx <= dist || tmy <= dist || tmz <= dist));
			}
		}
	}
	// ---- a segment ends where the next one starts (uniform: every lane of the wave takes part) ----
	const u32 my_start = go ? lin : glin;  // a segment that does not start leaves the rest to its predecessor's own test
	const u32 nxt = __shfl_down(my_start, 1);
	const u32 end = (sg + 1u < nseg) ? nxt : glin;  // lanes of one ray are adjacent and never straddle a wave (nseg | 64)
	// ---- the walk ----
	const long long idist = __double_as_longlong(dist);
	const u32 lin_first = lin;
	u32 cnt = 0;  // uniform guard only; the lane's step count is the L1 distance it covered (below)
	while (go) {
		++cnt;
		atomicOr(&lds[lin >> 5], 1u << (lin & 31u));
		const bool cxy = tmx <= tmy, cxz = tmx <= tmz, cyz = tmy <= tmz;
		const bool selx = cxy & cxz;
		const bool sely = !cxy & cyz;
		const bool selz = !(selx | sely);
		lin += (u32)(selx ? dlx : (sely ? dly : dlz));
		const double nx = tmx + tdx, ny = tmy + tdy, nz = tmz + tdz;
		tmx = selx ? nx : tmx;
		tmy = sely ? ny : tmy;
		tmz = selz ? nz : tmz;
		// t_max values and dist are non-negative doubles: their order is the order of their bit patterns
		const bool more = (__double_as_longlong(tmx) <= idist) | (__double_as_longlong(tmy) <= idist) | (__double_as_longlong(tmz) <= idist);
		go = (lin != end) & more & (cnt < 4096u);
	}
	if (cnt >= 4096u) err |= ERR_RUNAWAY;  // (a packed grid has < 1024 cells per axis: a ray has < 3072 steps)
	{
		// every step moves one cell along one axis and never turns back: steps = L1 distance covered
		const u32 ny = 2u * (u32)gr.nb[1];
		const u32 r0 = lin_first / rowBits, r1 = lin / rowBits;
		const i32 ddx = (i32)(lin % rowBits) - (i32)(lin_first % rowBits), ddy = (i32)(r1 % ny) - (i32)(r0 % ny),
		          ddz = (i32)(r1 / ny) - (i32)(r0 / ny);
		steps += (u32)(abs(ddx) + abs(ddy) + abs(ddz));
	}
	__syncthreads();
	{
		const uint4* l4 = reinterpret_cast<const uint4*>(lds);
		uint4* out4 = reinterpret_cast<uint4*>(slabs) + (size_t)blockIdx.x * (lds_words >> 2);
		const u32 n4 = lds_words >> 2;
		for (u32 j = threadIdx.x; j < n4; j += blockDim.x) out4[j] = l4[j];
	}
	blockStoreSteps(steps, steps_part);
	if (oob) atomicAdd(&ctl->n_oob, oob);
	if (err) atomicOr(&ctl->err, err);
}

// ------------------------------------------------------------------------------------------------
// K2c: ray set-up, segmentation and walk in ONE launch (bit-per-cell LDS grid; the default for LiDAR-sized
// scans). k_walk's time is its longest ray: a lone wave issues a ~40-instruction step every ~190 cycles and
// a 250-step ray takes 20 us, while the chip as a whole would need 4 us for all the steps. So the unit of
// work becomes a SEGMENT of about K steps, whatever the ray:
//   1. one lane per ray: clip, keys, computeRayInit (raySetup);
//   2. the same lane cuts its ray at every w-th pop of the dominant axis (w chosen so that a segment has
//      ~K steps) and builds each cut's exact state with the three independent addition chains of k_dda_seg --
//      incrementally from cut to cut, so the whole ray costs one addition per step, not one step per step;
//   3. segments go through an LDS queue to whichever lane is free: every lane walks ~K steps, waves are
//      homogeneous, and a workgroup takes every nWG-th ray, so all workgroups see the same mix of rays.
// Marks go to the workgroup's private LDS bit grid, handed over as a slab (k_merge_slabs), as in k_walk.
// ------------------------------------------------------------------------------------------------
#define UFO_CAST_BATCH 256u   // rays set up per round of a workgroup
#define UFO_CAST_QCAP 1024u   // segment queue entries
struct SegRec {
	double tm[3];
	u32 lin, end;
	u32 ray;  // index into the round's ray constants | first << 31 (segment 0 of its ray: the do-while body runs unconditionally, OMB:1286)
	u32 pad;
};
struct RayConst {
	double td[3], dist;
	i32 dl[3];
	u32 glin;
};
struct RayHdr {
	double tm[3];
	u32 lin0, ax, w, nseg;  // nseg == 0: nothing to cut (not a "safe" ray)
	u32 off, pad;
};
#define UFO_CAST_LDS_EXTRA (UFO_CAST_BATCH * (sizeof(RayConst) + sizeof(RayHdr)) + UFO_CAST_QCAP * sizeof(SegRec) + 128u)
#define UFO_CAST2_BATCH 128u
#define UFO_CAST2_QCAP 512u
#define UFO_CAST2_LDS_EXTRA (UFO_CAST2_BATCH * (sizeof(RayConst) + sizeof(RayHdr)) + UFO_CAST2_QCAP * sizeof(SegRec) + 128u)

// MODE 0: the whole bit grid in LDS, handed over as a slab (above).
// Grids beyond LDS (e.g. 8 cm / 20 m: 0.8 MB) keep the rounds -- set-up, cuts, balanced segment walk -- and change where a
// mark goes; `slabs` is the (zeroed) global bit grid then and nothing is merged afterwards:
// MODE 2 (default): a workgroup takes the rays of CONSECUTIVE points of the cloud (whole stretches of the ray list as
//         workgroups of k_select wrote them). Points that follow one another in a scan lie next to one another in space,
//         and all rays start at the sensor: the rays stay inside a small box (sensor cell + their end cells), and that box
//         of the grid is what the workgroup keeps in LDS -- same marks, same speed as MODE 0
//         -- and ORs into the global grid at the end, one atomic per non-zero word. A stretch whose box does not fit
//         (an unordered cloud) falls back to MODE 1's marks for this workgroup.
// MODE 1: marks go to the global grid one by one: through a direct-mapped LDS filter with exact tags (the cell's linear
//         index) that removes what this workgroup's own rays repeat, then a fire-and-forget atomicOr. Scattered device-
//         scope atomics complete at ~17 per ns on this chip: 0.8 ms for the 8 cm scan's 14.5 M steps (measured), which is
//         why this is only the fallback.
#define UFO_CAST_FILT_LOG2 14u  // 16 Ki tags = 64 KiB of LDS
#define UFO_CAST_STRETCHES 16u  // k_cast<2>: a workgroup takes the rays of at most this many workgroups of k_select
template <bool GLOBAL>
__device__ inline void castMark(u32* __restrict__ lds, u32* __restrict__ grid, u32 lin)
{
	if (!GLOBAL) {
		atomicOr(&lds[lin >> 5], 1u << (lin & 31u));
		return;
	}
	const u32 h = (lin * 0x9E3779B1u) >> (32u - UFO_CAST_FILT_LOG2);
	if (lds[h] == lin) return;
	lds[h] = lin;
	// fire and forget: looking at the word first (is the bit there already?) would put a global round trip into the
	// dependent chain of every step (measured: 1.24 ms instead of 0.47 for the 8 cm scan)
	__hip_atomic_fetch_or(&grid[lin >> 5], 1u << (lin & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool GLOBAL>
__device__ inline u32 castMarkChecked(const Grid& gr, u32* __restrict__ lds, u32* __restrict__ grid, u32 rowBits, u32 planeBits, i32 cx, i32 cy,
                                      i32 cz, u32 lim, u32* oob)
{
	if (!GLOBAL) return markBitChecked(gr, lds, rowBits, planeBits, cx, cy, cz, lim, oob);
	if ((u32)cx >= lim || (u32)cy >= lim || (u32)cz >= lim) {
		++*oob;
		return 0;
	}
	const i32 lx = cx - gr.base[0], ly = cy - gr.base[1], lz = cz - gr.base[2];
	if ((u32)lx >= 2u * (u32)gr.nb[0] || (u32)ly >= 2u * (u32)gr.nb[1] || (u32)lz >= 2u * (u32)gr.nb[2]) return ERR_GRID_OOB;
	castMark<true>(lds, grid, (u32)lx + (u32)ly * rowBits + (u32)lz * planeBits);
	return 0;
}

// start and goal cell of a ray's walk (the head of raySetup): everything the walk marks lies in the box they span
__device__ inline bool rayEndCells(const MapGeom& g, const D3& sensor, u32 depth, D3 to, i32 c0[3], i32 c1[3])
{
	D3 from = sensor;
	if (!moveLineInside(g, from, to)) return false;
	c0[0] = (i32)(toKey1(g, to.x, depth) >> depth);
	c0[1] = (i32)(toKey1(g, to.y, depth) >> depth);
	c0[2] = (i32)(toKey1(g, to.z, depth) >> depth);
	c1[0] = (i32)(toKey1(g, from.x, depth) >> depth);
	c1[1] = (i32)(toKey1(g, from.y, depth) >> depth);
	c1[2] = (i32)(toKey1(g, from.z, depth) >> depth);
	return true;
}

__device__ inline u32 castMarkCheckedDyn(bool glob, const Grid& gr, u32* __restrict__ lds, u32* __restrict__ grid, u32 rowBits, u32 planeBits,
                                         i32 cx, i32 cy, i32 cz, u32 lim, u32* oob)
{
	return glob ? castMarkChecked<true>(gr, lds, grid, rowBits, planeBits, cx, cy, cz, lim, oob)
	            : castMarkChecked<false>(gr, lds, grid, rowBits, planeBits, cx, cy, cz, lim, oob);
}

template <int MODE>
__global__ __launch_bounds__(512) void k_cast(MapGeom g, D3 sensor, u32 depth, Grid grid_all, u32* __restrict__ slabs,
                                              const D3* __restrict__ ray_end, u32 k_min, const ScanCtl* ctl_in, ScanCtl* ctl,
                                              unsigned long long* __restrict__ steps_part, u32 lds_grid_bytes,
                                              const u32* __restrict__ blk_range, u32 n_blk, u32 box_limit_bytes)
{
	extern __shared__ __attribute__((aligned(16))) u32 lds[];
	// rays set up per round, segment queue entries (MODE 2: smaller rounds leave more of the LDS to the box)
	constexpr u32 BATCH = (2 == MODE) ? UFO_CAST2_BATCH : UFO_CAST_BATCH, QCAP = (2 == MODE) ? UFO_CAST2_QCAP : UFO_CAST_QCAP;
	if (ctl_in->err & ERR_SPEC) return;  // the scan does not fit the predicted grid: it will be repeated (uniform exit)
	const u32 n = ctl_in->n_rays;
	// rays of this workgroup: blockIdx.x, blockIdx.x + gridDim.x, ...; MODE 2: the stretches of the ray list that
	// UFO_CAST_STRETCHES-or-fewer consecutive workgroups of k_select wrote (their points are consecutive in the cloud)
	__shared__ u32 st_first[UFO_CAST_STRETCHES], st_pre[UFO_CAST_STRETCHES + 1];
	u32 mine = (n > blockIdx.x) ? (n - blockIdx.x + gridDim.x - 1) / gridDim.x : 0u;
	if (2 == MODE) {
		const u32 per = (n_blk + gridDim.x - 1u) / gridDim.x;  // <= UFO_CAST_STRETCHES (host)
		if (0 == threadIdx.x) {
			u32 acc = 0;
			for (u32 k = 0; k < UFO_CAST_STRETCHES; ++k) {
				const u32 b = blockIdx.x * per + k;
				const bool have = k < per && b < n_blk;
				st_first[k] = have ? blk_range[2u * b] : 0u;
				st_pre[k] = acc;
				acc += have ? blk_range[2u * b + 1u] : 0u;
			}
			st_pre[UFO_CAST_STRETCHES] = acc;
		}
		__syncthreads();
		mine = st_pre[UFO_CAST_STRETCHES];
	}
	auto rayIndex = [&](u32 i) -> size_t {  // position in the ray list of this workgroup's i-th ray
		if (2 != MODE) return blockIdx.x + (size_t)i * gridDim.x;
		u32 k = 0;
		while (k + 1u < UFO_CAST_STRETCHES && i >= st_pre[k + 1u]) ++k;
		return (size_t)st_first[k] + (i - st_pre[k]);
	};
	// LDS: [grid region: the bit grid / this pass's box of it / the filter][round constants, headers, segment queue]
	const u32 region_words = (0 == MODE) ? (u32)(grid_all.bytes >> 2) : (1 == MODE ? (1u << UFO_CAST_FILT_LOG2) : (lds_grid_bytes >> 2));
	RayConst* rc = reinterpret_cast<RayConst*>(lds + region_words);
	RayHdr* hd = reinterpret_cast<RayHdr*>(rc + BATCH);
	SegRec* q = reinterpret_cast<SegRec*>(hd + BATCH);
	u32* sh = reinterpret_cast<u32*>(q + QCAP);  // [0..7], [16..23]: per-wave partial sums
	const u32 lim = 1u << (g.L - depth);
	const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	const u32 nwaves = min(8u, (blockDim.x + 63u) >> 6);
	unsigned long long steps = 0;
	u32 err = 0, oob = 0;
	// MODE 2 works through its rays in PASSES: as many consecutive rays as have a box that fits in LDS (the stretch is
	// halved until it does); a stretch of <= 32 rays whose box still does not fit is marked through the filter
	u32 ps = 0, pe = mine;
	for (;;) {
	Grid gr = grid_all;  // the grid the walk addresses: the scan's, or this pass's box of it
	bool glob = 1 == MODE;
	i32 boxo[3] = {0, 0, 0};  // offset of the box inside the scan's grid (cells; x a multiple of 32)
	pe = mine;
	if (2 == MODE) {
		__shared__ i32 bb[6];
		for (;;) {
			__syncthreads();
			if (threadIdx.x < 3u) {
				bb[threadIdx.x] = INT32_MAX;
				bb[3 + threadIdx.x] = INT32_MIN;
			}
			__syncthreads();
			i32 lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
			for (u32 i = ps + threadIdx.x; i < pe; i += blockDim.x) {
				i32 c0[3], c1[3];
				if (!rayEndCells(g, sensor, depth, ray_end[rayIndex(i)], c0, c1)) continue;
				for (int a = 0; a < 3; ++a) {
					lo[a] = min(lo[a], min(c0[a], c1[a]));
					hi[a] = max(hi[a], max(c0[a], c1[a]));
				}
			}
			for (int a = 0; a < 3; ++a) {
				const i32 l = waveMinI(lo[a]), h = waveMaxI(hi[a]);
				if (0 == (threadIdx.x & 63u)) {
					if (l != INT32_MAX) atomicMin(&bb[a], l);
					if (h != INT32_MIN) atomicMax(&bb[3 + a], h);
				}
			}
			__syncthreads();
			glob = true;
			if (bb[0] > bb[3]) break;  // (no ray of the stretch is walked at all)
			// the box as the host makes the scan's grid (makeGrid: one block of padding, even base), x moved down to a
			// multiple of 32 cells inside the scan's grid so that words map onto words; never beyond the scan's grid
			Grid sub = grid_all;
			bool ok = true;
			u64 rows = 1;
			for (int a = 0; a < 3; ++a) {
				i32 l = (bb[a] - 2) & ~1, h = bb[3 + a] + 2;
				if (0 == a) l = grid_all.base[0] + (((l - grid_all.base[0]) >> 5) << 5);
				l = max(l, grid_all.base[a]);
				h = min(h, grid_all.base[a] + 2 * grid_all.nb[a] - 1);
				if (h < l) ok = false;
				sub.base[a] = l;
				sub.nb[a] = (h - l) / 2 + 1;
				boxo[a] = l - grid_all.base[a];
				if (a) rows *= 2ull * (u64)sub.nb[a];
			}
			const u64 bytes = (((u64)(gridRowBits(sub) >> 3) * rows) + 15ull) & ~15ull;
			if (ok && bytes <= (u64)box_limit_bytes) {
				sub.bytes = bytes;
				gr = sub;
				glob = false;
				break;
			}
			if (pe - ps <= 32u) break;
			pe = ps + (pe - ps + 1u) / 2u;
		}
		if (glob && pe > ps && 0 == threadIdx.x) {
			atomicAdd(&ctl->dbg[40], 1ull);  // (diagnostics: passes marked through the filter)
			ctl->dbg[41] = ((u64)(u32)(bb[3] - bb[0]) << 40) | ((u64)(u32)(bb[4] - bb[1]) << 20) | (u64)(u32)(bb[5] - bb[2]);
			ctl->dbg[42] = ((u64)ps << 32) | pe;
			ctl->dbg[43] = mine;
		}
	}
	const u32 lds_words = glob ? (1u << UFO_CAST_FILT_LOG2) : (u32)(gr.bytes >> 2);
	{
		uint4* l4 = reinterpret_cast<uint4*>(lds);
		const u32 fill = glob ? 0xFFFFFFFFu : 0u;  // (filter tags: no cell has this index)
		for (u32 j = threadIdx.x; j < (lds_words >> 2); j += blockDim.x) l4[j] = make_uint4(fill, fill, fill, fill);
	}
	const u32 rowBits = gridRowBits(gr), planeBits = rowBits * 2u * (u32)gr.nb[1];
	// in rounds of BATCH
	for (u32 base = ps; base < pe; base += BATCH) {
		__syncthreads();  // previous round's queue and constants are no longer read (also orders the LDS zeroing)
		const u32 t = threadIdx.x;
		const bool have = t < BATCH && base + t < pe;
		// ---- 1. one lane per ray: clip, keys, computeRayInit ----
		u32 l1 = 0, dmax = 0, ax = 0, status = 0, lin0 = 0;
		if (have) {
			RayState r;
			raySetup(g, sensor, depth, gr, ray_end[rayIndex(base + t)], r);
			status = r.status;
			if (1 == r.status) {
				err |= castMarkCheckedDyn(glob, gr, lds, slabs, rowBits, planeBits, r.start[0], r.start[1], r.start[2], lim, &oob);
				steps += 1;
			} else if (3 == r.status) {
				// clipped ray (rare): sequential walk with every step checked, by this lane alone
				i32 cx = r.start[0], cy = r.start[1], cz = r.start[2];
				double ax_ = r.tm[0], ay_ = r.tm[1], az_ = r.tm[2];
				const u64 budget = 3ull * (1ull << g.L) + 8;
				u64 st = 0;
				bool more;
				do {
					if (++st > budget) {
						err |= ERR_RUNAWAY;
						break;
					}
					err |= castMarkCheckedDyn(glob, gr, lds, slabs, rowBits, planeBits, cx, cy, cz, lim, &oob);
					if (ax_ <= ay_) {
						if (ax_ <= az_) {
							cx += r.s[0];
							ax_ += r.td[0];
						} else {
							cz += r.s[2];
							az_ += r.td[2];
						}
					} else {
						if (ay_ <= az_) {
							cy += r.s[1];
							ay_ += r.td[1];
						} else {
							cz += r.s[2];
							az_ += r.td[2];
						}
					}
					more = (cx != r.goal[0] || cy != r.goal[1] || cz != r.goal[2]) && (fmin(fmin(ax_, ay_), az_) <= r.dist);
				} while (more);
				steps += st > budget ? budget : st;
			} else if (2 == r.status) {
				const u32 dxn = (u32)abs((i32)(r.gpk & 1023u) - (i32)(r.pk0 & 1023u));
				const u32 dyn = (u32)abs((i32)((r.gpk >> 10) & 1023u) - (i32)((r.pk0 >> 10) & 1023u));
				const u32 dzn = (u32)abs((i32)(r.gpk >> 20) - (i32)(r.pk0 >> 20));
				ax = (dxn >= dyn && dxn >= dzn) ? 0u : (dyn >= dzn ? 1u : 2u);
				dmax = ax == 0 ? dxn : (ax == 1 ? dyn : dzn);
				l1 = dxn + dyn + dzn;
				lin0 = pkToLin(r.pk0, rowBits, planeBits);
				RayConst c;
				c.td[0] = r.td[0];
				c.td[1] = r.td[1];
				c.td[2] = r.td[2];
				c.dist = r.dist;
				c.dl[0] = r.s[0];
				c.dl[1] = (i32)r.s[1] * (i32)rowBits;
				c.dl[2] = (i32)r.s[2] * (i32)planeBits;
				c.glin = pkToLin(r.gpk, rowBits, planeBits);
				rc[t] = c;
				hd[t].tm[0] = r.tm[0];
				hd[t].tm[1] = r.tm[1];
				hd[t].tm[2] = r.tm[2];
			}
		}
		// ---- 2. segment length for this round: about K steps, and the queue must hold every segment ----
		u32 tot = l1, cntr = (2 == status) ? 1u : 0u;
		for (int o = 32; o > 0; o >>= 1) {
			tot += __shfl_xor(tot, o);
			cntr += __shfl_xor(cntr, o);
		}
		if (0 == lane && wave < 8u) {
			sh[wave] = tot;
			sh[16 + wave] = cntr;
		}
		__syncthreads();
		u32 total = 0, nray2 = 0;
		for (u32 wv = 0; wv < nwaves; ++wv) {
			total += sh[wv];
			nray2 += sh[16 + wv];
		}
		// A ray of l1 steps is cut every w = floor(dmax*K/l1) pops of its dominant axis. With x = dmax*K/l1 >= K/3 (l1 <= 3 dmax)
		// w >= x - 1, so the ray has at most dmax/(x-1) + 1 = l1/(K - l1/dmax) + 1 <= l1/(K-3) + 1 segments: the queue holds
		// them all if K >= total/room + 3. (The bound used to be 2*l1/K + 1, i.e. K >= 2*total/room: segments twice as long
		// as the queue needs when it is the queue that decides -- long rays, k_cast<2>'s small queue.)
		u32 K = k_min;
		{
			const u32 room = QCAP - nray2;  // >= QCAP - BATCH > 0
			const u32 need = (total + room - 1u) / room + 3u;
			K = max(K, need);
		}
		u32 w = 1, nseg = 0;
		if (2 == status) {
			w = (u32)(((u64)dmax * K) / l1);
			if (w < 1u) w = 1u;
			nseg = (dmax + w - 1u) / w;  // >= 1 (start and goal differ)
		}
		__syncthreads();  // sh[] is reused below
		u32 incl = nseg;
		for (int o = 1; o < 64; o <<= 1) {
			const u32 v = __shfl_up(incl, o);
			if ((int)lane >= o) incl += v;
		}
		if (63u == lane && wave < 8u) sh[wave] = incl;
		__syncthreads();
		u32 off = incl - nseg, nsegs = 0;
		for (u32 wv = 0; wv < nwaves; ++wv) {
			const u32 v = sh[wv];
			if (wv < wave) off += v;
			nsegs += v;
		}
		if (t < BATCH) {
			hd[t].lin0 = lin0;
			hd[t].ax = ax;
			hd[t].w = w;
			hd[t].nseg = nseg;
			hd[t].off = off;
		}
		for (u32 si = threadIdx.x; si < nsegs; si += blockDim.x) q[si].lin = 0;  // cut cells are summed up from two lanes
		__syncthreads();
		// ---- 3. cut states from the three independent addition chains (k_dda_seg), two lanes per ray: each
		//         owns one of the non-dominant axes and runs the dominant chain a* alongside ----
		// after k0 = j*w pops of a*: element A[k0-1] (= v) was popped and t_max_a* = A[k0]; of the other axis the
		// elements before v were popped (strictly smaller, or equal when the axis has priority: the lower axis
		// index wins ties, VEC3:244-251) -- their count moves the cut's cell.
		for (u32 idx = threadIdx.x; idx < 2u * BATCH; idx += blockDim.x) {
			const u32 role = idx >= BATCH ? 1u : 0u, ry = idx - role * BATCH;
			const RayHdr h = hd[ry];
			if (0 == h.nseg) continue;
			const RayConst c = rc[ry];
			if (0 == role) {
				SegRec rec;
				rec.tm[0] = h.tm[0];
				rec.tm[1] = h.tm[1];
				rec.tm[2] = h.tm[2];
				rec.lin = h.lin0;
				rec.end = c.glin;
				rec.ray = ry | 0x80000000u;
				rec.pad = 0;
				q[h.off] = rec;
			}
			const u32 axd = h.ax;
			const u32 b = (0 == role) ? (axd == 0 ? 1u : 0u) : (axd == 2 ? 1u : 2u);
			const bool pri = b < axd;
			double ta = axd == 0 ? h.tm[0] : (axd == 1 ? h.tm[1] : h.tm[2]), v = ta;
			const double tda = axd == 0 ? c.td[0] : (axd == 1 ? c.td[1] : c.td[2]);
			const i32 da = axd == 0 ? c.dl[0] : (axd == 1 ? c.dl[1] : c.dl[2]);
			double tb = b == 0 ? h.tm[0] : (b == 1 ? h.tm[1] : h.tm[2]);
			const double dbt = b == 0 ? c.td[0] : (b == 1 ? c.td[1] : c.td[2]);
			const i32 dbl = b == 0 ? c.dl[0] : (b == 1 ? c.dl[1] : c.dl[2]);
			u32 ka = 0, cb = 0;
			for (u32 j = 1; j < h.nseg; ++j) {
				const u32 k0 = j * h.w;  // < dmax
				for (; ka < k0; ++ka) {
					v = ta;
					ta = ta + tda;
				}
				// pop while the condition holds, four candidates at a time (same sequence of additions)
				while (cb < 2048u) {
					const double s1 = tb + dbt, s2 = s1 + dbt, s3 = s2 + dbt;
					const bool c0 = pri ? (tb <= v) : (tb < v);
					if (!c0) break;
					const bool c1 = pri ? (s1 <= v) : (s1 < v), c2 = pri ? (s2 <= v) : (s2 < v), c3 = pri ? (s3 <= v) : (s3 < v);
					if (c1 & c2 & c3) {
						tb = s3 + dbt;
						cb += 4u;
					} else {
						tb = c1 ? (c2 ? s3 : s2) : s1;
						cb += 1u + (c1 ? (c2 ? 2u : 1u) : 0u);
						break;
					}
				}
				if (cb >= 2048u) err |= ERR_RUNAWAY;  // (guard of the pop loop: cannot trip inside a grid of < 1024 cells per axis)
				SegRec* o = &q[h.off + j];
				if (0 == role) {
					o->tm[axd] = ta;
					o->end = c.glin;
					o->ray = ry;
				}
				o->tm[b] = tb;
				atomicAdd(&o->lin, (0 == role ? h.lin0 + (u32)((i32)k0 * da) : 0u) + (u32)((i32)cb * dbl));
			}
		}
		__syncthreads();
		// 3c. a segment stops where the next one of its ray starts
		for (u32 si = threadIdx.x; si + 1u < nsegs; si += blockDim.x)
			if (!(q[si + 1u].ray & 0x80000000u)) q[si].end = q[si + 1u].lin;
		__syncthreads();
		// ---- 4. every lane walks segments ----
		for (u32 si = threadIdx.x; si < nsegs; si += blockDim.x) {
			const SegRec rec = q[si];
			const RayConst c = rc[rec.ray & 0x7FFFFFFFu];
			double tmx = rec.tm[0], tmy = rec.tm[1], tmz = rec.tm[2];
			const double tdx = c.td[0], tdy = c.td[1], tdz = c.td[2];
			const long long idist = __double_as_longlong(c.dist);
			const i32 dlx = c.dl[0], dly = c.dl[1], dlz = c.dl[2];
			const u32 end = rec.end;
			u32 lin = rec.lin;
			// loop condition at the segment start (OMB:1300); segment 0 starts the do-while unconditionally.
			// t_max values and dist are non-negative doubles: their order is the order of their bit patterns
			bool go = (0 != (rec.ray & 0x80000000u)) ||
			          ((lin != c.glin) && ((__double_as_longlong(tmx) <= idist) | (__double_as_longlong(tmy) <= idist) |
			                               (__double_as_longlong(tmz) <= idist)));
			const u32 lin_first = lin;
			u32 cnt = 0;  // uniform guard only
			while (go) {
				++cnt;
				if (glob) castMark<true>(lds, slabs, lin);
				else castMark<false>(lds, slabs, lin);
				const bool cxy = tmx <= tmy, cxz = tmx <= tmz, cyz = tmy <= tmz;
				const bool selx = cxy & cxz;
				const bool sely = !cxy & cyz;
				const bool selz = !(selx | sely);
				lin += (u32)(selx ? dlx : (sely ? dly : dlz));
				const double nx = tmx + tdx, ny = tmy + tdy, nz = tmz + tdz;
				tmx = selx ? nx : tmx;
				tmy = sely ? ny : tmy;
				tmz = selz ? nz : tmz;
				const bool more =
				    (__double_as_longlong(tmx) <= idist) | (__double_as_longlong(tmy) <= idist) | (__double_as_longlong(tmz) <= idist);
				go = (lin != end) & more & (cnt < 4096u);
			}
			if (cnt >= 4096u) err |= ERR_RUNAWAY;  // (a segment is ~K steps by construction: the guard cannot trip)
			// every step moves one cell along one axis and never turns back: steps = L1 distance covered
			const u32 ny2 = 2u * (u32)gr.nb[1];
			const u32 r0 = lin_first / rowBits, r1 = lin / rowBits;
			const i32 ddx = (i32)(lin % rowBits) - (i32)(lin_first % rowBits), ddy = (i32)(r1 % ny2) - (i32)(r0 % ny2),
			          ddz = (i32)(r1 / ny2) - (i32)(r0 / ny2);
			steps += (u32)(abs(ddx) + abs(ddy) + abs(ddz));
		}
	}
	__syncthreads();
	if (0 == MODE) {
		const uint4* l4 = reinterpret_cast<const uint4*>(lds);
		uint4* out4 = reinterpret_cast<uint4*>(slabs) + (size_t)blockIdx.x * (lds_words >> 2);
		const u32 n4 = lds_words >> 2;
		for (u32 j = threadIdx.x; j < n4; j += blockDim.x) out4[j] = l4[j];
	} else if (!glob) {
		// the box goes into the scan's grid: a row of the box is a stretch of a row of the grid, word for word
		const u32 rowW = rowBits >> 5, ny = 2u * (u32)gr.nb[1];
		const u32 growW = gridRowBits(grid_all) >> 5, gny = 2u * (u32)grid_all.nb[1];
		for (u32 j = threadIdx.x; j < lds_words; j += blockDim.x) {
			const u32 wv = lds[j];
			if (0 == wv) continue;
			const u32 wx = j % rowW, r = j / rowW;
			const u32 ly = r % ny, lz = r / ny;
			if (lz >= 2u * (u32)gr.nb[2]) continue;  // (padding behind the last row)
			const size_t gi = ((size_t)(lz + (u32)boxo[2]) * gny + (ly + (u32)boxo[1])) * growW + wx + ((u32)boxo[0] >> 5);
			__hip_atomic_fetch_or(&slabs[gi], wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
	if (2 != MODE || pe >= mine) break;
	ps = pe;
	}  // passes
	blockStoreSteps(steps, steps_part);
	if (0 != MODE && 0 == threadIdx.x && steps_part[blockIdx.x]) atomicAdd(&ctl->n_steps, steps_part[blockIdx.x]);  // (nobody merges afterwards)
	if (oob) atomicAdd(&ctl->n_oob, oob);
	if (err) atomicOr(&ctl->err, err);
}

// OR the per-workgroup slabs of the ray kernels (LDS-grid mode) into grid M. Workgroup = 64 columns of
// 16 bytes x 16 slab lanes: every thread ORs the slabs s = lane, lane+16, ... of its column (independent
// loads, 1 KiB contiguous per slab per wave), the 16 partial results are combined through LDS.
// steps_part (optional): per-workgroup step counts of k_walk / k_cast, folded into ScanCtl::n_steps here --
// one atomic per wave on that single word would cost the ray kernel ~12 ns each, serialised.
__global__ __launch_bounds__(1024) void k_merge_slabs(const uint4* __restrict__ slabs, u32 n_slabs, u32 n4, uint4* __restrict__ grid,
                                                      const unsigned long long* __restrict__ steps_part, ScanCtl* ctl)
{
	__shared__ uint4 part[16][64];
	const u32 col = threadIdx.x & 63u, sl = threadIdx.x >> 6;
	if (steps_part && 0 == blockIdx.x && threadIdx.x < 64u) {
		unsigned long long v = 0;
		for (u32 s = threadIdx.x; s < n_slabs; s += 64u) v += steps_part[s];
		for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
		if (0 == threadIdx.x && v) atomicAdd(&ctl->n_steps, v);
	}
	for (u32 j0 = blockIdx.x * 64u; j0 < n4; j0 += gridDim.x * 64u) {
		const u32 j = j0 + col;
		uint4 acc = make_uint4(0, 0, 0, 0);
		if (j < n4) {
			for (u32 s = sl; s < n_slabs; s += 16u) {
				const uint4 a = slabs[(size_t)s * n4 + j];
				acc.x |= a.x;
				acc.y |= a.y;
				acc.z |= a.z;
				acc.w |= a.w;
			}
		}
		part[sl][col] = acc;
		__syncthreads();
		if (0 == sl && j < n4) {
			for (u32 k = 1; k < 16u; ++k) {
				const uint4 a = part[k][col];
				acc.x |= a.x;
				acc.y |= a.y;
				acc.z |= a.z;
				acc.w |= a.w;
			}
			grid[j] = acc;
		}
		__syncthreads();
	}
}

// ------------------------------------------------------------------------------------------------
// K3 extract: non-zero bytes of the grids -> update list, one entry per touched node block.
// ------------------------------------------------------------------------------------------------
// MERGED (insert depth 0 only): one list for the whole scan -- a block with misses also carries the mask of
// its hit children (looked up in the hit-block hash, which is flagged so that k_extract_hits skips it).
template <bool MERGED>
__global__ __launch_bounds__(256) void k_extract(MapGeom g, Grid gr, const u32* __restrict__ grid, u32 which,
                                                 Entry* __restrict__ entries, u32 cap, ScanCtl* ctl, HitBlocks hb)
{
	const u64 nwords = gr.bytes >> 2;
	const u32 level = gr.depth + 1;
	const u64 stride = (u64)gridDim.x * blockDim.x;
	// uniform trip count so that every lane reaches the wave-aggregated append
	const u64 iters = (nwords + stride - 1) / stride;
	// The list's counter is one word: an atomic per wave and grid word serialises at ~12 ns each (0.5 M of them = 6 ms on
	// the 134 MB grid of C3 at insert depth 0, where reading the grid takes 0.05). A lane reads SIXTEEN words, the wave
	// reserves room for all their entries with one atomic, then the entries are written.
	constexpr u32 CH = 16;
	u64 w0 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	for (u64 it0 = 0; it0 < iters; it0 += CH, w0 += stride * CH) {
		u32 mw[CH];
		u32 total = 0;
#pragma unroll
		for (u32 q = 0; q < CH; ++q) {
			const u64 wq = w0 + stride * q;
			const u32 mq = (it0 + q < iters && wq < nwords) ? grid[wq] : 0u;
			mw[q] = mq;
			total += ((mq & 0xFFu) ? 1u : 0u) + ((mq & 0xFF00u) ? 1u : 0u) + ((mq & 0xFF0000u) ? 1u : 0u) + ((mq & 0xFF000000u) ? 1u : 0u);
		}
		if (0 == __ballot(total != 0)) continue;
		u32 pos = waveAppendN(&ctl->n_entries[which], total);
		if (0 == total) continue;
#pragma unroll
		for (u32 q = 0; q < CH; ++q) {
		const u32 m = mw[q];
		if (0 == m) continue;
		const u64 w = w0 + stride * q;
		for (u32 b = 0; b < 4; ++b) {
			u32 mb = (m >> (8 * b)) & 0xFF;
			if (mb == 0) continue;
			u32 my = pos++;
			if (my >= cap) continue;
			u64 idx = w * 4 + b;
			u64 bx = idx % (u64)gr.nb[0];
			u64 r = idx / (u64)gr.nb[0];
			u64 by = r % (u64)gr.nb[1];
			u64 bz = r / (u64)gr.nb[1];
			// absolute block coordinate = (cell >> 1); cells are key >> depth (key includes the +M offset)
			u32 ax = (u32)((gr.base[0] >> 1) + (i32)bx), ay = (u32)((gr.base[1] >> 1) + (i32)by),
			    az = (u32)((gr.base[2] >> 1) + (i32)bz);
			u64 p = morton3(ax, ay, az) & ((1ULL << (3 * (g.L - level))) - 1ULL);
			Entry e;
			e.lk = (1ULL << (3 * (g.L - level))) | p;
			e.hit = which ? 0 : (u8)mb;
			e.miss = which ? (u8)mb : 0;
			e.level = (u8)level;
			e.c_last = (u8)(31 - __clz((int)mb));  // misses: ascending code order -> highest child
			e.t_last = 0;
			if (MERGED) {
				e.miss = (u8)mb;
				const u32 hs = hitBlocksFind(hb, p);
				if (hs != 0xFFFFFFFFu) {
					const u32 hm = hb.mask[hs];
					e.hit = (u8)hm;
					hb.mask[hs] = hm | UFO_HB_TAKEN;  // this thread is the only one that looks this block up
				} else {
					e.hit = 0;
				}
			}
			entries[my] = e;
		}
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Grid::layout 2: NO dense grid. The reference's CodeMap has no size bound (code.h:568-785); a scan whose bounding box
// would need more scratch than ufomap_map_set_scratch_limit allows (a few far returns in an otherwise small scan: the
// box is what grows, not the number of cells) keeps the ray cells in a hash set of node blocks instead: key = Morton
// code of the block at level depth+1, value = mask of its marked children -- the update-list record, deduplicated as
// it is produced. One lane per ray, every step is one insert-or-OR; the host doubles the set and repeats the walk if it
// fills up (ERR_ENTRIES). Slower per step than any of the grid kernels (a device-scope CAS/OR per step, ~17 per ns for
// the whole chip), but bounded by the cells the rays touch, not by the box they span.
// ------------------------------------------------------------------------------------------------
struct MissSet {
	u64* keys;  // ~0 = empty
	u32* mask;
	u32 cap_mask;
	u32* count;  // occupied slots
};
__device__ inline u32 missSetMark(const MissSet& ms, i32 cx, i32 cy, i32 cz, u32 lim, u32* oob)
{
	if ((u32)cx >= lim || (u32)cy >= lim || (u32)cz >= lim) {
		++*oob;  // key outside [0, 2^L): dropped, as gridMark does
		return 0;
	}
	const u64 key = morton3((u32)cx >> 1, (u32)cy >> 1, (u32)cz >> 1);
	const u32 bit = 1u << ((cx & 1) | ((cy & 1) << 1) | ((cz & 1) << 2));
	u32 s = hash64(key) & ms.cap_mask;
	const u32 max_probe = (ms.cap_mask >> 1) + 1u;
	for (u32 probe = 0; probe < max_probe; ++probe) {
		u64 k = __hip_atomic_load(&ms.keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (k == ~0ULL) {
			// (a load factor above 1/2 counts as full: the host doubles the set and repeats the walk)
			if (__hip_atomic_load(ms.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (ms.cap_mask >> 1)) return ERR_ENTRIES;
			const u64 prev = atomicCAS((unsigned long long*)&ms.keys[s], ~0ULL, (unsigned long long)key);
			if (prev == ~0ULL) atomicAdd(ms.count, 1u);
			k = (prev == ~0ULL) ? key : prev;
		}
		if (k == key) {
			if (!(__hip_atomic_load(&ms.mask[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(&ms.mask[s], bit);
			return 0;
		}
		s = (s + 1) & ms.cap_mask;
	}
	return ERR_ENTRIES;
}

__global__ __launch_bounds__(256) void k_dda_set(MapGeom g, D3 sensor, u32 depth, Grid gr, MissSet ms, const D3* __restrict__ ray_end,
                                                 const ScanCtl* ctl_in, ScanCtl* ctl)
{
	const u32 n = ctl_in->n_rays;
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	const u32 lim = 1u << (g.L - depth);
	unsigned long long steps = 0;
	u32 err = 0, oob = 0;
	if (i < n) {
		RayState r;
		raySetup(g, sensor, depth, gr, ray_end[i], r);
		if (1 == r.status) {
			err |= missSetMark(ms, r.start[0], r.start[1], r.start[2], lim, &oob);
			steps = 1;
		} else if (0 != r.status) {
			// freeSpaceNormal (OMB:1261-1301) step by step, every cell through the set
			i32 cx = r.start[0], cy = r.start[1], cz = r.start[2];
			double tx = r.tm[0], ty = r.tm[1], tz = r.tm[2];
			const u64 budget = 3ull * (1ull << g.L) + 8;
			bool more;
			do {
				if (++steps > budget) {
					err |= ERR_RUNAWAY;
					break;
				}
				err |= missSetMark(ms, cx, cy, cz, lim, &oob);
				if (err & ERR_ENTRIES) break;  // the walk is repeated with a larger set
				if (tx <= ty) {
					if (tx <= tz) {
						cx += r.s[0];
						tx += r.td[0];
					} else {
						cz += r.s[2];
						tz += r.td[2];
					}
				} else {
					if (ty <= tz) {
						cy += r.s[1];
						ty += r.td[1];
					} else {
						cz += r.s[2];
						tz += r.td[2];
					}
				}
				more = (cx != r.goal[0] || cy != r.goal[1] || cz != r.goal[2]) && (fmin(fmin(tx, ty), tz) <= r.dist);
			} while (more);
			if (steps > budget) steps = budget;
		}
	}
	waveAddU64(&ctl->n_steps, steps);
	if (oob) atomicAdd(&ctl->n_oob, oob);
	if (err) atomicOr(&ctl->err, err);
}

// the set's records -> update list (k_extract for Grid::layout 2); MERGED as there
template <bool MERGED>
__global__ __launch_bounds__(256) void k_extract_set(MapGeom g, Grid gr, MissSet ms, u32 which, Entry* __restrict__ entries, u32 cap, ScanCtl* ctl,
                                                     HitBlocks hb)
{
	const u32 level = gr.depth + 1;
	const u32 nslots = ms.cap_mask + 1u;
	const u32 stride = gridDim.x * blockDim.x;
	const u32 iters = (nslots + stride - 1) / stride;  // uniform trip count: every lane reaches the wave-aggregated append
	u32 s = blockIdx.x * blockDim.x + threadIdx.x;
	for (u32 it = 0; it < iters; ++it, s += stride) {
		const bool have = s < nslots && ms.keys[s] != ~0ULL && 0 != (ms.mask[s] & 0xFFu);
		const u32 my = waveAppend(&ctl->n_entries[which], have);
		if (!have || my >= cap) continue;
		const u64 p = ms.keys[s] & ((1ULL << (3 * (g.L - level))) - 1ULL);
		const u32 mb = ms.mask[s] & 0xFFu;
		Entry e;
		e.lk = (1ULL << (3 * (g.L - level))) | p;
		e.hit = 0;
		e.miss = (u8)mb;
		e.level = (u8)level;
		e.c_last = (u8)(31 - __clz((int)mb));  // misses: ascending code order -> highest child
		e.t_last = 0;
		if (MERGED) {
			const u32 hs = hitBlocksFind(hb, p);
			if (hs != 0xFFFFFFFFu) {
				const u32 hm = hb.mask[hs];
				e.hit = (u8)hm;
				hb.mask[hs] = hm | UFO_HB_TAKEN;
			}
		}
		entries[my] = e;
	}
}
// the set's cells as codes (stage-level parity export)
__global__ __launch_bounds__(256) void k_set_codes(MissSet ms, u64* __restrict__ codes, u32 cap, ScanCtl* ctl)
{
	const u32 nslots = ms.cap_mask + 1u;
	for (u32 s = blockIdx.x * blockDim.x + threadIdx.x; s < nslots; s += gridDim.x * blockDim.x) {
		const u64 k = ms.keys[s];
		if (k == ~0ULL) continue;
		u32 m = ms.mask[s] & 0xFFu;
		while (m) {
			const u32 c = (u32)__ffs(m) - 1u;
			m &= m - 1u;
			const u32 pos = atomicAdd(&ctl->n_codes, 1u);
			if (pos < cap) codes[pos] = (k << 3) | (u64)c;
		}
	}
}

// grid bits -> codes (shifted by 3*depth) for the stage-level parity exports
__global__ __launch_bounds__(256) void k_grid_codes(MapGeom g, Grid gr, const u32* __restrict__ grid,
                                                    u64* __restrict__ codes, u32 cap, ScanCtl* ctl)
{
	u64 nwords = gr.bytes >> 2;
	for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (u64)gridDim.x * blockDim.x) {
		u32 m = grid[w];
		while (m) {
			u32 bit = __ffs(m) - 1;
			m &= m - 1;
			u64 idx = w * 4 + (bit >> 3);
			u32 c = bit & 7;
			u64 bx = idx % (u64)gr.nb[0];
			u64 r = idx / (u64)gr.nb[0];
			u64 by = r % (u64)gr.nb[1];
			u64 bz = r / (u64)gr.nb[1];
			u32 x = (u32)(gr.base[0] + 2 * (i32)bx + (i32)(c & 1)), y = (u32)(gr.base[1] + 2 * (i32)by + (i32)((c >> 1) & 1)),
			    z = (u32)(gr.base[2] + 2 * (i32)bz + (i32)((c >> 2) & 1));
			// cell coordinate = key >> depth; Code(key) >> 3*depth == morton(cell) for keys < 2^21
			u64 code = morton3(x, y, z);
			u32 pos = atomicAdd(&ctl->n_codes, 1u);
			if (pos < cap) codes[pos] = code;
		}
	}
}

// ------------------------------------------------------------------------------------------------
// K3 / stage export for Grid::layout 1 (one bit per cell): a thread takes one 32-cell word of an even row
// (y, z even) together with the same word of rows (y+1, z), (y, z+1), (y+1, z+1) = 16 node blocks.
// ------------------------------------------------------------------------------------------------
template <bool MERGED>
__global__ __launch_bounds__(256) void k_extract_bits(MapGeom g, Grid gr, const u32* __restrict__ grid, u32 which,
                                                      Entry* __restrict__ entries, u32 cap, ScanCtl* ctl, HitBlocks hb)
{
	const u32 rowW = gridRowBits(gr) >> 5;
	const u32 nby = (u32)gr.nb[1], nbz = (u32)gr.nb[2];
	const u64 nitems = (u64)rowW * nby * nbz * 16u;  // one thread per node block; 16 consecutive threads share four words
	const u32 level = gr.depth + 1;
	const u64 stride = (u64)gridDim.x * blockDim.x;
	const u64 iters = (nitems + stride - 1) / stride;  // uniform trip count: every lane reaches the wave-aggregated append
	u64 it_ = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	for (u64 it = 0; it < iters; ++it, it_ += stride) {
		u32 mb = 0, bx = 0, by = 0, bz = 0;
		if (it_ < nitems) {
			const u32 j2 = ((u32)it_ & 15u) * 2u;
			const u64 wi = it_ >> 4;
			const u32 wx = (u32)(wi % rowW);
			const u64 r = wi / rowW;
			by = (u32)(r % nby);
			bz = (u32)(r / nby);
			bx = wx * 16u + (j2 >> 1);
			const u64 row00 = ((u64)(2 * bz) * (2 * nby) + 2 * by) * rowW + wx;
			const u32 w00 = grid[row00], w10 = grid[row00 + rowW], w01 = grid[row00 + (u64)2 * nby * rowW],
			          w11 = grid[row00 + (u64)2 * nby * rowW + rowW];
			mb = ((w00 >> j2) & 3u) | (((w10 >> j2) & 3u) << 2) | (((w01 >> j2) & 3u) << 4) | (((w11 >> j2) & 3u) << 6);
		}
		const bool have = 0 != mb;
		const u32 my = blockAppend(&ctl->n_entries[which], have);  // one atomic per workgroup (uniform trip count)
		if (!have || my >= cap) continue;
		u32 ax = (u32)((gr.base[0] >> 1) + (i32)bx), ay = (u32)((gr.base[1] >> 1) + (i32)by), az = (u32)((gr.base[2] >> 1) + (i32)bz);
		u64 p = morton3(ax, ay, az) & ((1ULL << (3 * (g.L - level))) - 1ULL);
		Entry e;
		e.lk = (1ULL << (3 * (g.L - level))) | p;
		e.hit = which ? 0 : (u8)mb;
		e.miss = which ? (u8)mb : 0;
		e.level = (u8)level;
		e.c_last = (u8)(31 - __clz((int)mb));  // misses: ascending code order -> highest child
		e.t_last = 0;
		if (MERGED) {
			e.miss = (u8)mb;
			const u32 hs = hitBlocksFind(hb, p);
			if (hs != 0xFFFFFFFFu) {
				const u32 hm = hb.mask[hs];
				e.hit = (u8)hm;
				hb.mask[hs] = hm | UFO_HB_TAKEN;
			} else {
				e.hit = 0;
			}
		}
		entries[my] = e;
	}
}

__global__ __launch_bounds__(256) void k_grid_codes_bits(MapGeom g, Grid gr, const u32* __restrict__ grid, u64* __restrict__ codes,
                                                         u32 cap, ScanCtl* ctl)
{
	const u32 rowW = gridRowBits(gr) >> 5;
	const u32 ny = 2u * (u32)gr.nb[1];
	const u64 nwords = (u64)rowW * ny * 2u * (u32)gr.nb[2];
	for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (u64)gridDim.x * blockDim.x) {
		u32 m = grid[w];
		const u32 wx = (u32)(w % rowW);
		const u64 r = w / rowW;
		const u32 ly = (u32)(r % ny), lz = (u32)(r / ny);
		while (m) {
			const u32 bit = __ffs(m) - 1;
			m &= m - 1;
			const u32 x = (u32)(gr.base[0] + (i32)(wx * 32u + bit)), y = (u32)(gr.base[1] + (i32)ly), z = (u32)(gr.base[2] + (i32)lz);
			const u64 code = morton3(x, y, z);
			const u32 pos = atomicAdd(&ctl->n_codes, 1u);
			if (pos < cap) codes[pos] = code;
		}
	}
}

// ------------------------------------------------------------------------------------------------
// early_stopping > 0 (occupancy_map_base.h:1289-1298, 1327-1333; the server's dynamic-reconfigure range is 0 .. 10): a ray
// ends once that many cells IN A ROW were in the scan's set already -- put there by a ray cast EARLIER (the reference casts
// the rays one after the other, in the order of the cloud), or, with fixed-step casting, by the ray's own step before.
// Which cells a ray visits therefore depends on where every earlier ray stopped. Exact parallel form: a ray's stop is a
// function of the stops of the rays before it alone, so the scan's stops are the unique fixed point of
//     first[c] = lowest rank among the rays that visit cell c within their current stop   (k_es_mark: atomicMin)
//     stop[r]  = where ray r stops when "in the set already" means first[c] < rank(r)      (k_es_stops)
// Iterated from "nobody stops": after round k the k lowest-ranked rays are final (induction on the rank), in practice the
// estimates bracket the answer and a LiDAR sweep settles in a handful of rounds; the host stops when no stop moved. Rank =
// index of the ray's point in the cloud (k_select appends workgroup by workgroup, not in order). Then the visited cells go
// into grid M (k_es_mark with the grid) and the update proceeds as without early stopping.
// ------------------------------------------------------------------------------------------------
struct EsArgs {
	u32* first;        // [cells of the ray box] lowest rank that visits the cell (0xFFFFFFFF: nobody); nullptr: the sparse form below
	// Round 6: a ray box whose dense array does not fit the scratch limit (a 2 mm frame at insert depth 0: 4e9 cells = 15 GB) keeps "who
	// visits a cell first" in a hash of the cells the rays DO visit (key = the cell's index in the box, open addressing; the host doubles
	// it and repeats the round when it fills up) -- bounded by the scan's ray cells, as the reference's CodeMap is (CODE:568-785)
	u64* hkeys;        // [hmask + 1] cell index (~0: empty)
	u32* hvals;        // [hmask + 1] lowest rank
	u32 hmask;
	u32* stop;         // [rays] cells the ray visits (0xFFFFFFFF: to its end)
	const u32* rank;   // [rays] the ray's point index
	u32 early;         // early_stopping
	u32 simple;        // fixed-step casting
};
// the cells of one ray, in the order the reference visits them: visit(cx, cy, cz) returns false to end the ray; returns the
// number of cells visited. (The checked sequential walk: this path is about exactness, not speed.)
template <typename V>
__device__ inline u32 esWalk(const MapGeom& g, const D3& sensor, u32 depth, D3 to, bool simple, u32* err, V&& visit)
{
	D3 from = sensor;
	if (!moveLineInside(g, from, to)) return 0;
	D3 cur = to, end = from;
	D3 dir = end - cur;
	const double dist = norm(dir);
	dir = dir / dist;
	const u64 budget = 3ull * (1ull << g.L) + 8;
	u32 n = 0;
	if (simple) {
		const double ns = nodeSize(g, depth);
		const int num_steps = (int)(dist / ns);
		if (num_steps < 0 || (u64)num_steps > budget) {
			*err |= ERR_RUNAWAY;
			return 0;
		}
		const D3 stepv = dir * ns;
		for (int s = 0; s <= num_steps; ++s) {
			++n;
			if (!visit((i32)(toKey1(g, cur.x, depth) >> depth), (i32)(toKey1(g, cur.y, depth) >> depth), (i32)(toKey1(g, cur.z, depth) >> depth))) break;
			cur = cur + stepv;
		}
		return n;
	}
	const u32 kx = toKey1(g, cur.x, depth), ky = toKey1(g, cur.y, depth), kz = toKey1(g, cur.z, depth);
	const u32 ex = toKey1(g, end.x, depth), ey = toKey1(g, end.y, depth), ez = toKey1(g, end.z, depth);
	i32 cx = (i32)(kx >> depth), cy = (i32)(ky >> depth), cz = (i32)(kz >> depth);
	if (kx == ex && ky == ey && kz == ez) {
		(void)visit(cx, cy, cz);
		return 1;
	}
	const i32 gx = (i32)(ex >> depth), gy = (i32)(ey >> depth), gz = (i32)(ez >> depth);
	const double node_size = nodeSize(g, depth), half = g.hs[depth];
	double bx = toCoord1(g, kx, depth) - cur.x, by = toCoord1(g, ky, depth) - cur.y, bz = toCoord1(g, kz, depth) - cur.z;
	i32 sx, sy, sz;
	double tdx, tdy, tdz, tmx, tmy, tmz;
#define UFO_AXIS_INIT(d, b, s, td, tm)                 \
	if (0 < d) {                                        \
		s = 1;                                          \
		b += half;                                      \
		td = node_size / fabs(d);                       \
		tm = b / d;                                     \
	} else if (0 > d) {                                 \
		s = -1;                                         \
		b -= half;                                      \
		td = node_size / fabs(d);                       \
		tm = b / d;                                     \
	} else {                                            \
		s = 0;                                          \
		td = 1.7976931348623157e308;                    \
		tm = 1.7976931348623157e308;                    \
	}
	UFO_AXIS_INIT(dir.x, bx, sx, tdx, tmx)
	UFO_AXIS_INIT(dir.y, by, sy, tdy, tmy)
	UFO_AXIS_INIT(dir.z, bz, sz, tdz, tmz)
#undef UFO_AXIS_INIT
	bool go;
	do {
		if ((u64)++n > budget) {
			*err |= ERR_RUNAWAY;
			break;
		}
		if (!visit(cx, cy, cz)) break;
		if (tmx <= tmy) {
			if (tmx <= tmz) {
				cx += sx;
				tmx += tdx;
			} else {
				cz += sz;
				tmz += tdz;
			}
		} else {
			if (tmy <= tmz) {
				cy += sy;
				tmy += tdy;
			} else {
				cz += sz;
				tmz += tdz;
			}
		}
		go = (cx != gx || cy != gy || cz != gz) && (fmin(fmin(tmx, tmy), tmz) <= dist);
	} while (go);
	return n;
}
// index of a cell in the dense array over the ray box (cells, x fastest); false: outside (keys beyond the map's range are
// dropped by the marking as everywhere else, cells outside the box cannot happen)
__device__ inline bool esCell(const Grid& gr, i32 cx, i32 cy, i32 cz, u32 lim, u64* idx, u32* err)
{
	if ((u32)cx >= lim || (u32)cy >= lim || (u32)cz >= lim) return false;
	const i32 lx = cx - gr.base[0], ly = cy - gr.base[1], lz = cz - gr.base[2];
	if ((u32)lx >= 2u * (u32)gr.nb[0] || (u32)ly >= 2u * (u32)gr.nb[1] || (u32)lz >= 2u * (u32)gr.nb[2]) {
		*err |= ERR_GRID_OOB;
		return false;
	}
	*idx = (u64)lx + 2ull * (u64)gr.nb[0] * ((u64)ly + 2ull * (u64)gr.nb[1] * (u64)lz);
	return true;
}
__device__ inline u32 esHash(u64 k)
{
	k ^= k >> 33;
	k *= 0xff51afd7ed558ccdULL;
	k ^= k >> 29;
	return (u32)k;
}
// first[idx] = min(first[idx], rank)
__device__ inline void esFirstMin(const EsArgs& a, u64 idx, u32 rank, u32* err)
{
	if (a.first) {
		atomicMin(&a.first[idx], rank);
		return;
	}
	u32 s = esHash(idx) & a.hmask;
	for (u32 probe = 0; probe <= a.hmask; ++probe) {
		u64 k = __hip_atomic_load(&a.hkeys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (k == ~0ull) {
			const u64 prev = atomicCAS((unsigned long long*)&a.hkeys[s], ~0ull, (unsigned long long)idx);
			k = (prev == ~0ull) ? idx : prev;
		}
		if (k == idx) {
			atomicMin(&a.hvals[s], rank);
			return;
		}
		if (probe >= 256u) break;  // (the set is filling up: the host doubles it and repeats the round)
		s = (s + 1u) & a.hmask;
	}
	*err |= ERR_ENTRIES;
}
// first[idx] (0xFFFFFFFF: nobody visits the cell)
__device__ inline u32 esFirstGet(const EsArgs& a, u64 idx)
{
	if (a.first) return a.first[idx];
	u32 s = esHash(idx) & a.hmask;
	for (u32 probe = 0; probe <= 256u; ++probe) {
		const u64 k = __hip_atomic_load(&a.hkeys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (k == idx) return a.hvals[s];
		if (k == ~0ull) return 0xFFFFFFFFu;
		s = (s + 1u) & a.hmask;
	}
	return 0xFFFFFFFFu;  // (cannot happen in a round whose marking pass did not overflow: an insert gives up after as many probes)
}
// every ray marks the cells it visits within its current stop: first[c] = min rank; grid != nullptr (the final pass): the
// cells go into grid M as well, the visits are the scan's step count
__global__ __launch_bounds__(256) void k_es_mark(MapGeom g, D3 sensor, u32 depth, Grid gr, EsArgs a, const D3* __restrict__ ray_end, const ScanCtl* ctl_in, ScanCtl* ctl,
                                                 u32* __restrict__ grid)
{
	const u32 n = ctl_in->n_rays;
	const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
	unsigned long long steps = 0;
	u32 err = 0, oob = 0;
	if (r < n) {
		const u32 lim = 1u << (g.L - depth), rank = a.rank[r], stop = a.stop[r];
		u32 k = 0;
		steps = esWalk(g, sensor, depth, ray_end[r], 0 != a.simple, &err, [&](i32 cx, i32 cy, i32 cz) {
			u64 idx;
			if (esCell(gr, cx, cy, cz, lim, &idx, &err)) {
				esFirstMin(a, idx, rank, &err);
				if (grid) (void)gridMark(gr, grid, cx, cy, cz, lim, &oob);
			} else if (grid && ((u32)cx >= lim || (u32)cy >= lim || (u32)cz >= lim)) {
				++oob;
			}
			return ++k < stop;
		});
	}
	if (grid) {
		waveAddU64(&ctl->n_steps, steps);
		if (oob) atomicAdd(&ctl->n_oob, oob);
	}
	if (err) atomicOr(&ctl->err, err);
}
// every ray walks its whole path against `first` and finds where it stops; *changed counts the rays whose stop moved
__global__ __launch_bounds__(256) void k_es_stops(MapGeom g, D3 sensor, u32 depth, Grid gr, EsArgs a, const D3* __restrict__ ray_end, const ScanCtl* ctl_in, ScanCtl* ctl,
                                                  u32* changed)
{
	const u32 n = ctl_in->n_rays;
	const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
	u32 err = 0;
	bool moved = false;
	if (r < n) {
		const u32 lim = 1u << (g.L - depth), rank = a.rank[r];
		u32 row = 0;
		u64 prev = ~0ull;
		bool stopped = false;
		const u32 visited = esWalk(g, sensor, depth, ray_end[r], 0 != a.simple, &err, [&](i32 cx, i32 cy, i32 cz) {
			u64 idx = ~0ull;
			bool already = false;
			if (esCell(gr, cx, cy, cz, lim, &idx, &err)) already = esFirstGet(a, idx) < rank || (a.simple && idx == prev);
			prev = idx;
			if (!already) {
				row = 0;
				return true;
			}
			if (++row >= a.early) {
				stopped = true;
				return false;
			}
			return true;
		});
		const u32 stop = stopped ? visited : 0xFFFFFFFFu;
		moved = stop != a.stop[r];
		a.stop[r] = stop;
	}
	const u32 c = (u32)__popcll(__ballot(moved));
	if (0 == (threadIdx.x & 63u) && c) atomicAdd(changed, c);
	if (err) atomicOr(&ctl->err, err);
}
// insert depth > 0: the ray of a cell is its FIRST point's (the cloud's order decides the ranks): every candidate point
// registers under its cell with atomicMin before k_select lets the winner cast (without early stopping whichever point
// creates the entry casts: all rays into one cell are identical)
template <bool DISCRETE>
__global__ __launch_bounds__(256) void k_es_raycells(MapGeom g, D3 sensor, u32 n, u32 depth, HitHash hh, const D3* __restrict__ pt_end, const u8* __restrict__ pt_flag,
                                                     const u32* __restrict__ pt_slot, ScanCtl* ctl)
{
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n || !DISCRETE || 0 == depth) return;
	const u8 flag = pt_flag[i];
	bool cast = 0 != (flag & PF_CAST);
	if (flag & PF_HITCAND) {
		const u32 s = pt_slot[i];
		if (!(s != NONE && hh.minidx[s] == i)) cast = false;  // OMB:358-360: dropped entirely, no ray
	}
	if (!cast) return;
	D3 cur = sensor, e2 = pt_end[i];
	if (!moveLineInside(g, cur, e2)) return;
	const u32 k0 = toKey1(g, e2.x, depth), k1 = toKey1(g, e2.y, depth), k2 = toKey1(g, e2.z, depth);
	(void)hitHashInsert(hh, morton3(k0, k1, k2) | (1ULL << 63), i, &ctl->err);
}
}  // namespace ufo
