// fast_kernels.h -- the steady-state pipeline of a depth-0 scan on a predicted ray grid (plain and colour maps):
//
//   prep stream   k_fhits  (k_signal)                                   (needs only the cloud; keeps the points for a repeat)
//   scan stream            (k_done_gate) k_fcast                        (never touches the map; k_done_gate = the end of the
//                                                                        previous scan half + this one's gate, one launch)
//   map stream                               (k_claim) k_fmerge  k_tile  k_ftail     ONE walk of the tree for every scan
//                                                                                     that has queued up by then
//   a ray grid beyond LDS (up to 65 536 tiles): k_fselect + k_cast<2> (scan_kernels.h) instead of k_fcast, and k_up --
//   level 4 of the tree in parallel -- between k_tile and k_ftail; a synchronous call with nothing in flight: the five
//   kernels on the map stream alone, no hand-over kernels
//
// four working launches per scan on the streams that matter, and the three of the map stream are shared by all the scans
// a walk takes -- where the general path (scan_kernels.h / map_kernels.h: classify, select, reduce_boxes, hitmark, cast,
// merge_slabs, extract x2, ensure, init_new, apply_leaf, propagate x2, propagate_tail) needs fourteen per scan. What makes
// that possible:
//   * the scan runs on the ray grid PREDICTED from the previous scans (host_fast_path.inl: predictGrid), so every array of
//     the scan has a dense, known geometry: "first point in a voxel wins" (CodeSet `indices_`, occupancy_map_base.h:295,
//     358-360) is one atomicMin on a dense u32 array over the grid's cells instead of a hash insert, and nothing has to
//     be compacted into lists between kernels;
//   * k_fcast takes a point's ray from the 32-byte record k_fhits left for it and drops the points that lost their voxel
//     itself, instead of reading a ray list that a separate compaction kernel wrote;
//   * what a scan hands to the tree update is two bit grids (ray cells, hit voxels) and a bitmap of the depth-3 tiles they
//     touch -- 0.2 MB, the same whether the update runs on this GPU or on seven others (ufomap_map_insert_batch);
//   * the tree update is TILED and BATCHED: one wavefront owns one depth-3 node (8x8x8 voxels: 64 level-1 node blocks, 8
//     level-2 blocks, 1 level-3 block) and does everything the reference's updateValue does beneath it -- createNode with
//     inheritance, updateOccupancy for hits then misses, updateNode / pruning on the way up (occupancy_map_base.h:
//     1063-1224, octree.h:997-1162) -- in registers and cross-lane operations, for B scans in order, reading each block's
//     record once and writing it once; the few hundred node blocks above depth 3 are finished by ONE workgroup that
//     holds them in LDS (k_ftail), where a level costs a barrier instead of a round trip to HBM;
//   * which scans a walk takes is decided on the device when the walk starts (k_claim): whatever has queued up.
// Semantics are those of the general path (map_kernels.h: "last-update chain"), which stays in place for everything
// else -- first scans, insert depth > 0, simple ray casting, more than 1022 cells per axis -- and doubles as the on-GPU
// cross-check of this file (tests run both on the same scans).
#pragma once
#include "map_kernels.h"

#define UFO_BIG_MAX_TILES 65536u  // depth-3 tiles of a ray grid beyond LDS (k_fselect / k_cast<2> / k_up); its level-4 cells: <= UFO_FAST_MAX_TILES
#define UFO_FAST_MAX_TILES 8192u  // depth-3 tiles of a ray grid that takes the fast path (the host keeps larger grids off it)

namespace ufo
{
// cross-lane moves by DPP (a VALU operand modifier, a few clocks) instead of ds_bpermute (an LDS-unit operation, ~120 clocks): see grpMax below
template <int CTRL>
__device__ __forceinline__ u32 dppU(u32 v)
{
	return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ float dppF(float v)
{
	return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dppD(double v)
{
	const unsigned long long b = (unsigned long long)__double_as_longlong(v);
	const u32 lo = dppU<CTRL>((u32)b), hi = dppU<CTRL>((u32)(b >> 32));
	return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
#define UFO_DPP_X1 0xB1   // quad_perm [1, 0, 3, 2]
#define UFO_DPP_X2 0x4E   // quad_perm [2, 3, 0, 1]
#define UFO_DPP_HM 0x141  // row_half_mirror
#define UFO_DPP_R8 0x128  // row_ror:8
// Geometry shared by these kernels: the predicted bit grid (Grid::layout 1) and the depth-3 tiles that cover it.
struct FastGeo {
	Grid gr;
	u32 rowBits, planeBits;  // bits per row of cells (x), per plane (x, y): a cell's index is lx + ly*rowBits + lz*planeBits
	i32 tbase[3];            // absolute tile coordinate (cell >> 3) of tile (0, 0, 0)
	u32 nt[3];               // tiles per axis
	u32 ntiles;
	u32 ncells;              // entries of the first-point array (planeBits * 2*nb[2])
	u32 tl;                  // level of the "tiles" tbase / nt / ntiles describe: 3 (k_tile's); k_ftail after k_up: 4 (FastGeo of the level-4 cells)
};

// One input point through the head loop of insertPointCloud (occupancy_map_base.h:281-303) or
// insertPointCloudDiscrete (354-398; colour variant occupancy_map_color.h:195-233) at insert depth 0, up to but not
// including "first point in the voxel wins". Same arithmetic, same order as k_classify / k_select (scan_kernels.h).
struct PointRay {
	D3 end;       // ray end as freeSpace gets it (discrete: centre of the end's voxel)
	u32 cell;     // index of the hit voxel in the grid (hit candidates only)
	bool cast;    // a ray is cast unless the point loses its voxel to an earlier point (discrete mode)
	bool hitcand; // the point lies in range: its voxel receives a hit if the point is the first one in it
	bool odd;     // needs the general path: clipped at the map cube, or outside the predicted grid
	u32 oct;      // octant of the ray's end cell as seen from the sensor's cell: bit a set = end cell >= sensor cell on axis a (k_fcast4)
	u32 l1;       // ... and the ray's length in cells (sum over the axes)
};
template <bool DISCRETE>
__device__ inline PointRay pointRay(const MapGeom& g, const FastGeo& fg, const D3& sensor, const double* __restrict__ xyz, const Ingest& ing,
                                    u32 i, double max_range, u32 color_variant, double amn[3], double amx[3])
{
	PointRay r;
	r.cast = r.hitcand = r.odd = false;
	r.cell = 0;
	r.oct = 0;
	r.l1 = 0;
	D3 end;
	if (!loadPoint(xyz, ing, i, &end)) {
		r.end = end;
		return r;
	}
	const double h = g.hs[g.L];
	// the fast path handles segments that lie inside the map cube (no clipping, moveLineInside leaves them alone)
	if (!inBBX(sensor, h) || !inBBX(end, h)) {
		r.odd = true;
		r.end = end;
		return r;
	}
	D3 hit_at = end;
	if (DISCRETE) {
		const double sq_max = max_range * max_range;
		double dsq = sqnorm(end - sensor);
		if (0 > max_range || dsq < sq_max) {
			r.hitcand = true;
		} else {
			D3 c{toCoord1(g, toKey1(g, end.x, 0), 0), toCoord1(g, toKey1(g, end.y, 0), 0), toCoord1(g, toKey1(g, end.z, 0), 0)};
			D3 dir = c - sensor;
			if (color_variant) {
				dsq = sqnorm(dir);
				if (0 <= max_range && dsq > sq_max) {
					dir = dir / sqrt(dsq);
					end = sensor + (dir * max_range);
				}
			} else {
				const double dist = norm(dir);
				dir = dir / dist;
				if (0 <= max_range && dist > max_range) end = sensor + (dir * max_range);
			}
		}
		// OMB:371-398: the ray ends at the centre of its end's voxel
		const u32 k0 = toKey1(g, end.x, 0), k1 = toKey1(g, end.y, 0), k2 = toKey1(g, end.z, 0);
		const D3 ec{toCoord1(g, k0, 0), toCoord1(g, k1, 0), toCoord1(g, k2, 0)};
		const D3 cc{toCoord1(g, toKey1(g, sensor.x, 0), 0), toCoord1(g, toKey1(g, sensor.y, 0), 0), toCoord1(g, toKey1(g, sensor.z, 0), 0)};
		const double t = g.hs[0];
		for (int a = 0; a < 3; ++a) {
			amn[a] = fmin(ec[a] - t, cc[a] - t);
			amx[a] = fmax(ec[a] + t, cc[a] + t);
		}
		end = ec;
	} else {
		D3 dir = end - sensor;
		const double dist = norm(dir);
		if (0 > max_range || dist <= max_range) {
			r.hitcand = true;
		} else {
			dir = dir / dist;
			end = sensor + (dir * max_range);
			if (!inBBX(end, h)) r.odd = true;
		}
		for (int a = 0; a < 3; ++a) {
			amn[a] = fmin(end[a], sensor[a]);
			amx[a] = fmax(end[a], sensor[a]);
		}
	}
	r.cast = true;
	r.end = end;
	// both end cells inside the interior of the predicted grid (one block of padding) and inside the key range: every
	// cell of the walk is, then (a DDA path is monotone per axis)
	const i32 lim = (i32)((1u << g.L) - 1u);
	const i32 mx[3] = {2 * fg.gr.nb[0] - 2, 2 * fg.gr.nb[1] - 2, 2 * fg.gr.nb[2] - 2};
	i32 le[3], ls[3];
	for (int a = 0; a < 3; ++a) {
		const i32 ke = (i32)toKey1(g, end[a], 0), ks = (i32)toKey1(g, sensor[a], 0);
		if (ke < 0 || ke > lim || ks < 0 || ks > lim) r.odd = true;
		le[a] = ke - fg.gr.base[a];
		ls[a] = ks - fg.gr.base[a];
		if (le[a] < 1 || le[a] > mx[a] || ls[a] < 1 || ls[a] > mx[a]) r.odd = true;
		r.oct |= (le[a] >= ls[a] ? 1u : 0u) << a;
		r.l1 += (u32)abs(le[a] - ls[a]);
	}
	if (r.hitcand && !r.odd) {
		// the hit voxel: the original point's (== the ray end's voxel in both modes; continuous mode: hit_at == end)
		i32 lh[3];
		for (int a = 0; a < 3; ++a) {
			lh[a] = (i32)toKey1(g, hit_at[a], 0) - fg.gr.base[a];
			if (lh[a] < 1 || lh[a] > mx[a]) r.odd = true;
		}
		r.cell = (u32)lh[0] + (u32)lh[1] * fg.rowBits + (u32)lh[2] * fg.planeBits;
	}
	return r;
}

// ------------------------------------------------------------------------------------------------
// F1: first point in the voxel (atomicMin of the point index on the dense cell array), bounding boxes, validity of
// the predicted grid for this scan (ERR_SPEC: the host repeats the scan on the general path).
// ------------------------------------------------------------------------------------------------
// What k_fhits found out about a point, for k_fcast (which would otherwise run the same ~100 double-precision operations
// per point again): 32 bytes per point, written once, read once.
struct PointRec {
	D3 end;     // ray end as freeSpace gets it
	u32 cell;   // hit voxel's index in the grid (hit candidates only)
	u32 flags;  // 1 cast, 2 hit candidate, 4 odd
};
// BIN (round 6, k_fcast4): the records of the points that may cast a ray leave the kernel SORTED BY OCTANT inside every workgroup's
// 256-point stretch (a counting sort on ballots), with the stretch's eight counts beside them -- a workgroup of the ray kernel takes
// the rays of ONE octant, whose cells lie in one sub-box of the grid. The record carries the point's index (flags | index << 3).
template <bool DISCRETE, bool BIN = false>
__global__ __launch_bounds__(256) void k_fhits(MapGeom g, FastGeo fg, D3 sensor, const double* __restrict__ xyz, u32 n, double max_range,
                                               u32 color_variant, u32* __restrict__ first, BoxPartial* __restrict__ part, ScanCtl* ctl,
                                               Ingest ing, PointRec* __restrict__ recs, double* __restrict__ keep, const uint8_t* __restrict__ rgb_in,
                                               uint8_t* __restrict__ rgb_keep, uint4* __restrict__ bcnt = nullptr, u32* __restrict__ bwgt = nullptr)
{
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	double amn[3] = {1e300, 1e300, 1e300}, amx[3] = {-1e300, -1e300, -1e300};
	i32 ck[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, ek[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
	bool odd = false;
	PointRec pr;
	pr.flags = 0;
	u32 boct = 8u;  // BIN: the octant the point's record goes to (8: no record -- the point casts no ray)
	u32 bl1 = 0;    // ... and its ray's length in cells
	if (i < n) {
		if (keep) {
			// The caller's cloud is read by THIS kernel only: the point as it enters the head loop (map frame, float64; NaN for a
			// point the ingest drops) is kept in the hand-over set, so that a scan that has to be repeated -- it did not fit
			// its predicted grid -- never goes back to a buffer the caller may have reused since the call returned.
			D3 pt;
			if (!loadPoint(xyz, ing, i, &pt)) pt = D3{__builtin_nan(""), __builtin_nan(""), __builtin_nan("")};
			keep[3 * (size_t)i] = pt.x;
			keep[3 * (size_t)i + 1] = pt.y;
			keep[3 * (size_t)i + 2] = pt.z;
		}
		if (rgb_keep) {  // (the colours likewise: read by the tree update, long after the call has returned)
			rgb_keep[3 * (size_t)i] = rgb_in[3 * (size_t)i];
			rgb_keep[3 * (size_t)i + 1] = rgb_in[3 * (size_t)i + 1];
			rgb_keep[3 * (size_t)i + 2] = rgb_in[3 * (size_t)i + 2];
		}
		const PointRay r = pointRay<DISCRETE>(g, fg, sensor, xyz, ing, i, max_range, color_variant, amn, amx);
		odd = r.odd;
		pr.end = r.end;
		pr.cell = r.cell;
		pr.flags = (r.cast ? 1u : 0u) | (r.hitcand ? 2u : 0u) | (r.odd ? 4u : 0u);
		if (BIN) {
			pr.flags |= i << 3;
			if (r.cast && !r.odd) {
				boct = r.oct;
				bl1 = min(r.l1, 4095u);
			}
		} else recs[i] = pr;
		if (!r.odd) {
			if (r.hitcand) atomicMin(&first[r.cell], i);
			if (r.cast) {
				for (int a = 0; a < 3; ++a) {
					const i32 ka = (i32)toKey1(g, r.end[a], 0), kb = (i32)toKey1(g, sensor[a], 0);
					ck[a] = min(ka, kb);
					ek[a] = max(ka, kb);
				}
			}
		}
	}
	if (__ballot(odd) && 0 == (threadIdx.x & 63u)) atomicOr(&ctl->err, ERR_SPEC);
	if (BIN) {
		__shared__ u32 wcnt[4][8], wsum[8];
		const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
		if (threadIdx.x < 8u) wsum[threadIdx.x] = 0;
		__syncthreads();
		// (the octant's share of the ray kernel's WORK: the sum of squared lengths -- a point far away casts a long ray AND is likely to
		// be the first of its voxel; tracks the octants' true DDA steps within ~10 % on LiDAR scans where the counts are off by 3.6x)
		if (boct < 8u) atomicAdd(&wsum[boct], bl1 * bl1);
		u64 mine = 0;
#pragma unroll
		for (u32 o = 0; o < 8u; ++o) {
			const u64 mo = __ballot(boct == o);
			if (boct == o) mine = mo;
			if (0 == lane) wcnt[wave][o] = (u32)__popcll(mo);
		}
		__syncthreads();
		u32 tot[8];
#pragma unroll
		for (u32 o = 0; o < 8u; ++o) tot[o] = wcnt[0][o] + wcnt[1][o] + wcnt[2][o] + wcnt[3][o];
		if (boct < 8u) {
			u32 pos = (u32)__popcll(mine & ((1ULL << lane) - 1ULL));
#pragma unroll
			for (u32 o = 0; o < 8u; ++o) {
				if (o < boct) pos += tot[o];
				if (o == boct)
					for (u32 w = 0; w < 4u; ++w)
						if (w < wave) pos += wcnt[w][o];
			}
			recs[(size_t)blockIdx.x * 256u + pos] = pr;
		}
		if (0 == threadIdx.x) bcnt[blockIdx.x] = make_uint4(tot[0] | (tot[1] << 16), tot[2] | (tot[3] << 16), tot[4] | (tot[5] << 16), tot[6] | (tot[7] << 16));
		if (threadIdx.x < 8u) bwgt[8u * blockIdx.x + threadIdx.x] = wsum[threadIdx.x];
		__syncthreads();  // (blockBoxReduce's shared arrays follow)
	}
	i32 none_lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, none_hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
	blockBoxReduce(part, 7u, amn, amx, none_lo, none_hi, ck, ek);  // (folded by k_fmerge's last workgroup)
}

// ------------------------------------------------------------------------------------------------
// F2b (ray grids beyond LDS): "first point in the voxel wins" and the compaction of the surviving ray ends -- what
// k_fcast's prologue does per workgroup -- as a launch of its own, in the form the ray kernel of such grids reads
// (k_cast<2>, scan_kernels.h: the ray list in cloud order + where every 256-point stretch of the cloud starts in it).
// ------------------------------------------------------------------------------------------------
template <bool DISCRETE>
__global__ __launch_bounds__(256) void k_fselect(u32 n, u32* __restrict__ first, const PointRec* __restrict__ recs, D3* __restrict__ ray_end,
                                                 u32* __restrict__ blk_range, unsigned long long* __restrict__ parts, u32* __restrict__ gridH, u32 clean_first,
                                                 const ScanCtl* ctl)
{
	// No word that every workgroup would have to add to (each such atomic is ~12 ns, one after the other): a stretch's rays
	// go to the stretch's own 256 slots of the ray list, the counts to per-stretch partials (folded by k_fmerge).
	if (ctl->err) {  // (k_fhits: the scan does not fit the predicted grid; uniform exit, the scan will be repeated)
		if (0 == threadIdx.x) {
			blk_range[2u * blockIdx.x] = 0;
			blk_range[2u * blockIdx.x + 1u] = 0;
			parts[blockIdx.x] = 0;
		}
		return;
	}
	__shared__ u32 wcnt[4];
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	bool cast = false, winner = false;
	D3 end{0, 0, 0};
	if (i < n) {
		const PointRec r = recs[i];
		const bool odd = 0 != (r.flags & 4u);
		cast = (r.flags & 1u) && !odd;
		if ((r.flags & 2u) && !odd) {
			winner = first[r.cell] == i;
			if (DISCRETE && !winner) cast = false;  // OMB:358-360: dropped entirely, no ray
			if (winner) {
				// the voxel receives a hit (OMB:295, 358-360): its bit in the scan's hit grid; and the first-point array is left
				// clean for the set's next scan (a point of the same voxel that looks later finds "none", which is not its index
				// either) -- unless the tree update needs the first points for their colours and cleans up itself (k_tile)
				atomicOr(&gridH[r.cell >> 5], 1u << (r.cell & 31u));
				if (clean_first) first[r.cell] = 0xFFFFFFFFu;
			}
		}
		end = r.end;
	}
	const u64 mask = __ballot(cast);
	const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	if (0 == lane) wcnt[wave] = (u32)__popcll(mask);
	const u32 hcount = (u32)__syncthreads_count(winner ? 1 : 0);  // (a barrier: wcnt is complete)
	u32 off = 0, rcount = 0;
	for (u32 w = 0; w < 4u; ++w) {
		if (w < wave) off += wcnt[w];
		rcount += wcnt[w];
	}
	const u32 base = blockIdx.x * blockDim.x;
	if (cast) ray_end[base + off + (u32)__popcll(mask & ((1ULL << lane) - 1ULL))] = end;
	if (0 == threadIdx.x) {
		blk_range[2u * blockIdx.x] = base;
		blk_range[2u * blockIdx.x + 1u] = rcount;
		parts[blockIdx.x] = (unsigned long long)rcount | ((unsigned long long)hcount << 32);
	}
}

// Fold the per-workgroup bounding boxes of k_fhits (cell box of the rays: predicts the next grid; change AABB, OMB:305-308,
// 388-398) into the control block; called by ONE workgroup. (Atomics from every workgroup of k_fhits on these 12 words,
// even guarded by a load, tripled that kernel's time.)
__device__ inline void foldBoxes(const BoxPartial* __restrict__ boxes, u32 nboxes, ScanCtl* ctl)
{
	// The LAST workgroup of the launch does not merge: it folds the per-workgroup bounding boxes of k_fhits (cell box of
	// the rays: predicts the next grid; change AABB, OMB:305-308, 388-398) into the control block, beside the others.
	// (Atomics from every workgroup of k_fhits on these 12 words, even guarded by a load, tripled that kernel's time.)
	__shared__ double rd[16][6];
	__shared__ i32 ri[16][6];
	double amn[3] = {1e300, 1e300, 1e300}, amx[3] = {-1e300, -1e300, -1e300};
	i32 mmn[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, mmx[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
	for (u32 i = threadIdx.x; i < nboxes; i += blockDim.x) {
		const BoxPartial& p = boxes[i];
		for (int a = 0; a < 3; ++a) {
			amn[a] = fmin(amn[a], p.aabb_min[a]);
			amx[a] = fmax(amx[a], p.aabb_max[a]);
			mmn[a] = min(mmn[a], p.mb_min[a]);
			mmx[a] = max(mmx[a], p.mb_max[a]);
		}
	}
	const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	for (int a = 0; a < 3; ++a) {
		const double l = waveMinD(amn[a]), h = waveMaxD(amx[a]);
		const i32 il = waveMinI(mmn[a]), ih = waveMaxI(mmx[a]);
		if (0 == lane) {
			rd[wave][a] = l;
			rd[wave][3 + a] = h;
			ri[wave][a] = il;
			ri[wave][3 + a] = ih;
		}
	}
	__syncthreads();
	if (threadIdx.x < 6u) {
		const u32 a = threadIdx.x % 3u, k = threadIdx.x;
		const bool is_max = k >= 3u;
		const u32 nw = (blockDim.x + 63u) >> 6;
		double d = rd[0][k];
		i32 v = ri[0][k];
		for (u32 w = 1; w < nw; ++w) {
			d = is_max ? fmax(d, rd[w][k]) : fmin(d, rd[w][k]);
			v = is_max ? max(v, ri[w][k]) : min(v, ri[w][k]);
		}
		if (is_max) {
			ctl->mb_max[a] = v;
			ctl->hb_max[a] = v;
			if (d > -1e299) ctl->aabb_max[a] = encD(d);
		} else {
			ctl->mb_min[a] = v;
			ctl->hb_min[a] = v;
			if (d < 1e299) ctl->aabb_min[a] = encD(d);
		}
	}
}

// ---- octant sub-boxes of a ray grid around the sensor's cell (k_fcast4, below) ----
struct OctGeo {
	u32 s[3];            // the sensor's cell, local to the ray grid
	u32 xw0[2], nxw[2];  // per x bit of the octant: first 32-bit word of a sub-row, words per sub-row
	u32 y0[2], ny[2], z0[2], nz[2];
	u32 wmax4;           // the largest sub-box, in 16-byte units
};
__host__ __device__ inline u32 octWords4(const OctGeo& og, u32 o) { return (og.nxw[o & 1u] * og.ny[(o >> 1) & 1u] * og.nz[o >> 2] + 3u) >> 2; }
__host__ __device__ inline bool makeOctGeo(const MapGeom& g, const FastGeo& fg, const D3& sensor, OctGeo* og)
{
	const u32 n[3] = {2u * (u32)fg.gr.nb[0], 2u * (u32)fg.gr.nb[1], 2u * (u32)fg.gr.nb[2]};
	for (int a = 0; a < 3; ++a) {
		const long long c = (long long)toKey1(g, sensor[a], 0) - (long long)fg.gr.base[a];
		if (c < 0 || c >= (long long)n[a]) return false;  // (the scan will be flagged: the sensor lies outside its predicted grid)
		og->s[a] = (u32)c;
	}
	og->xw0[0] = 0;
	og->nxw[0] = (og->s[0] >> 5) + 1u;
	og->xw0[1] = og->s[0] >> 5;
	og->nxw[1] = ((n[0] - 1u) >> 5) - og->xw0[1] + 1u;
	og->y0[0] = 0;
	og->ny[0] = og->s[1] + 1u;
	og->y0[1] = og->s[1];
	og->ny[1] = n[1] - og->s[1];
	og->z0[0] = 0;
	og->nz[0] = og->s[2] + 1u;
	og->z0[1] = og->s[2];
	og->nz[1] = n[2] - og->s[2];
	og->wmax4 = 0;
	for (u32 o = 0; o < 8u; ++o) og->wmax4 = max(og->wmax4, octWords4(*og, o));
	return true;
}
struct OctTab {  // left by workgroup 0 of the ray kernel for k_fmerge
	OctGeo og;
	u32 wg_start[9];   // workgroups wg_start[o] .. wg_start[o + 1] - 1 took octant o
	u32 slab_off4[9];  // where the first of them stored its sub-box, in 16-byte units
};
// ---- scans as the tree update sees them (who applies which scan: see k_claim below) ----
#define UFO_RING 16u       // scans in flight per handle (a power of two, > the number of hand-over sets)
#define UFO_BATCH_MAX 16u  // scans per walk
#define UFO_XSLOT_CTL 1024u  // a rank's exchange slot of a batch step: [control block | tile bitmap at UFO_XSLOT_CTL | bit grids at UFO_XSLOT_HDR]
#define UFO_XSLOT_HDR 2048u
struct ScanDesc {  // a scan as the tree update sees it: written into the ring when its scan half ends (k_scan_done)
	const uint4* slabs;                  // the ray kernel's per-workgroup copies of the ray grid ...
	const unsigned long long* parts;     // ... and step / ray / hit counts (merged by the walk that takes the scan)
	u32* gridM;                          // ray cells of the scan (bit grid, Grid::layout 1; written by k_fmerge)
	u32* gridH;                          // hit voxels of the scan (same layout; written by k_fmerge from `first`)
	u32* first;                          // first point of every cell (k_fhits' atomicMin; 0xFFFFFFFF: none), left clean by k_fmerge
	u32* tile_bits;                      // depth-3 tiles of the grid that hold a ray cell (written by k_fmerge, cleared by k_ftail)
	ScanCtl* ctl;                        // control block (err: the scan half flagged the scan; the walk stands back)
	ScanCtl* host_result;                // where the finished control block goes (pinned), followed by the word the host polls
	const BoxPartial* boxes;             // k_fhits' per-workgroup bounding boxes
	unsigned long long done_value;       // value of that word: the integration's running number
	unsigned long long fseq;             // running number among the fast-path scans
	u32 n_slabs, nboxes;
	u32 geo;                             // scans with equal geo may share a walk (same ray grid, consecutive updates of the map)
	u32 pad;
	const uint8_t* rgb;                  // colour maps: the points' colours (3 bytes each; nullptr: a cloud without colours). The walk
	                                     // then reads `first` itself (the colour of a voxel's first point, OMC.h:195-233) and cleans it
	const OctTab* oct;                   // k_fcast4: the slabs are octant sub-boxes -- the table its first workgroup left (nullptr: copies of the whole grid)
};
struct Pipe {
	unsigned long long scan_done;  // fast-path number of the newest scan whose scan half has finished (the scan stream works them off in order)
	unsigned long long claimed;    // ... of the newest scan a walk has taken
	struct Slot {
		unsigned long long first;  // the slot of scan f (slot[f & 15]) applies scans first .. first + B - 1
		u32 B, pad;                // (B == 0: scan f went with an earlier walk)
	} slot[UFO_RING];
	u32 wstat[UFO_RING];           // wstat[f & 15]: 0 = the walk that took scan f applied it; else it stood back / failed
	ScanDesc ring[UFO_RING];
	unsigned long long* ts;        // developer aid (option "tstamps"): device clock at the pipeline's hand-overs, 8 words per scan
	u32 merge_arrivals, pad_;      // k_fmerge_batch: workgroups of the launch that have finished (left at 0 by the last one)
};
// [0] k_signal (first-point pass done)  [1] gate entered  [2] gate open  [3] scan half published  [4] k_claim entered
// [5] k_claim done  [6] k_ftail done  [7] scans the walk took; 100 MHz clock
#define UFO_TS_SCANS 4096u
__device__ __forceinline__ void tsMark(unsigned long long* ts, unsigned long long f, u32 k, unsigned long long v)
{
	if (ts) ts[(f & (UFO_TS_SCANS - 1u)) * 8u + k] = v;
}
// ------------------------------------------------------------------------------------------------
// F2' (round 5; round 3's k_fcast -- k_cast's round structure fed from the cloud, removed in round 6 -- ran its phases one behind the other):
// the phases in front of the walk FUSED. k_fcast ran head loop -> (barrier) -> set-up, one lane
// per ray -> (barrier) -> segment sizes, prefix sums -> (2 barriers) -> dominant chains, one lane per ray -> (barrier) -> the
// other axes, two lanes per ray -> (2 barriers) -> walk: a lone workgroup per CU (148 KB of LDS), so every phase is the latency
// of its own dependent chain with three of sixteen waves busy, and the ray ends go through global memory between the first two.
// Measured by the kernel's own stamps (workgroup 0, 16 cm LiDAR scan, scripts/dev/dev_fcast.py): head loop 5.1 us, set-up 2.3,
// sizes 1.5, dominant chains 2.4, other axes + fix-ups 5.2 = 16.4 us in front of a walk of 7.1.
// Here the lane that looks at a point does everything for the point's ray -- the winner test, clip / keys / computeRayInit, the
// three addition chains side by side in its registers (k_vcut's loop, vol_kernels.h: the dominant axis up to the cut, the two
// others while their elements precede the cut's) -- and puts the segments into the LDS queue through one returning LDS atomic:
// the rays of a 64-point stretch run in all of the workgroup's waves at once, nothing is handed from phase to phase, and ONE
// barrier stands between the cuts and the walk. Rays that find the queue full wait for the next round (a round = cuts, barrier,
// walk, barrier); the steady state needs one. Same cells, same step count: the cut states are the same sums.
// ------------------------------------------------------------------------------------------------
template <bool DISCRETE>
__global__ __launch_bounds__(1024) void k_fcast2(MapGeom g, FastGeo fg, D3 sensor, u32 n, const u32* __restrict__ first, u32* __restrict__ slabs, u32 K,
                                                const ScanCtl* ctl_in, ScanCtl* ctl, unsigned long long* __restrict__ steps_part, const PointRec* __restrict__ recs,
                                                u32 rcap, u32 qcap, u32 prio, Pipe* solo, ScanDesc solo_desc)
{
	if (solo && 0 == (threadIdx.x | blockIdx.x)) {
		solo->ring[0] = solo_desc;
		solo->slot[0].first = 0;
		solo->slot[0].B = 1;
	}
	if (prio >= 3u) __builtin_amdgcn_s_setprio(3);
	else if (2u == prio) __builtin_amdgcn_s_setprio(2);
	else if (1u == prio) __builtin_amdgcn_s_setprio(1);
	extern __shared__ __attribute__((aligned(16))) u32 lds[];
	const u32 err_in = ctl_in->err;  // (looked at once the LDS grid has been cleared: the load is in flight meanwhile)
	if (0 == (threadIdx.x | blockIdx.x)) ctl->dbg[30] = wall_clock64();  // (diagnostics)
	const Grid& gr = fg.gr;
	const u32 lds_words = (u32)(gr.bytes >> 2);
	RayConst* rc = reinterpret_cast<RayConst*>(lds + lds_words);
	SegRec* q = reinterpret_cast<SegRec*>(rc + rcap);
	u32* sh = reinterpret_cast<u32*>(q + qcap);  // [0] rays of the round, [1] queue entries asked for, [2] first entry that was refused, [3] rays cast, [4] voxels hit
	// the points of this workgroup: blockIdx.x, blockIdx.x + gridDim.x, ... -- their records are asked for before the grid is cleared
	const u32 pts = (n > blockIdx.x) ? (n - blockIdx.x + gridDim.x - 1) / gridDim.x : 0u;
	{
		uint4* l4 = reinterpret_cast<uint4*>(lds);
		for (u32 j = threadIdx.x; j < (lds_words >> 2); j += blockDim.x) l4[j] = make_uint4(0, 0, 0, 0);
	}
	if (threadIdx.x < 8u) sh[threadIdx.x] = (2u == threadIdx.x) ? 0xFFFFFFFFu : 0u;
	if (err_in) return;  // the scan does not fit the predicted grid (k_fhits): it will be repeated (uniform exit)
	__syncthreads();
	if (0 == (threadIdx.x | blockIdx.x)) ctl->dbg[31] = wall_clock64();  // (diagnostics)
	const u32 rowBits = fg.rowBits, planeBits = fg.planeBits;
	const u32 lim = 1u << g.L;
	const u32 lane = threadIdx.x & 63u;
	unsigned long long steps = 0;
	u32 err = 0, oob = 0, nhit = 0, ncast = 0;
	// A lane takes the workgroup's points threadIdx.x, threadIdx.x + blockDim.x, ... one after the other; a ray that finds the
	// queue full stays with its lane until the next round.
	u32 pcur = threadIdx.x;
	bool pending = false;
	RayState r;
	r.status = 0;
	u32 ax = 0, w = 1, nseg = 0, lin0 = 0, glin = 0;
	i32 dla = 0, dl0 = 0, dl1 = 0;
	for (u32 round = 0;; ++round) {  // (uniform: a round = cuts, barrier, walk, barrier)
		for (;;) {
			if (!pending) {
				if (pcur >= pts) break;
				const u32 i = blockIdx.x + pcur * gridDim.x;
				pcur += blockDim.x;
				const PointRec pr = recs[i];  // (k_fhits ran the head loop on the point)
				const bool odd = 0 != (pr.flags & 4u);
				bool cast = (pr.flags & 1u) && !odd;
				if ((pr.flags & 2u) && !odd) {
					// (the voxel receives a hit, OMB:295, 358-360: its first point's; k_fmerge derives the hit grid from the array)
					const bool winner = first[pr.cell] == i;
					nhit += winner ? 1u : 0u;
					if (DISCRETE && !winner) cast = false;  // OMB:358-360: dropped entirely, no ray
				}
				if (!cast) continue;
				++ncast;
				raySetup(g, sensor, 0u, gr, pr.end, r);
				if (1 == r.status) {
					err |= markBitChecked(gr, lds, rowBits, planeBits, r.start[0], r.start[1], r.start[2], lim, &oob);
					steps += 1;
				} else if (3 == r.status) {
					err |= ERR_GRID_OOB;  // cannot happen: pointRay admits only rays inside the grid's interior
				}
				if (2 != r.status) continue;
				const u32 dxn = (u32)abs((i32)(r.gpk & 1023u) - (i32)(r.pk0 & 1023u));
				const u32 dyn = (u32)abs((i32)((r.gpk >> 10) & 1023u) - (i32)((r.pk0 >> 10) & 1023u));
				const u32 dzn = (u32)abs((i32)(r.gpk >> 20) - (i32)(r.pk0 >> 20));
				ax = (dxn >= dyn && dxn >= dzn) ? 0u : (dyn >= dzn ? 1u : 2u);
				const u32 dmax = ax == 0 ? dxn : (ax == 1 ? dyn : dzn);
				const u32 l1 = dxn + dyn + dzn;
				w = (u32)(((u64)dmax * K) / l1);
				if (w < 1u) w = 1u;
				nseg = (dmax + w - 1u) / w;  // >= 1 (start and goal differ)
				lin0 = pkToLin(r.pk0, rowBits, planeBits);
				glin = pkToLin(r.gpk, rowBits, planeBits);
				const i32 dlx = (i32)r.s[0], dly = (i32)r.s[1] * (i32)rowBits, dlz = (i32)r.s[2] * (i32)planeBits;
				dla = ax == 0 ? dlx : (ax == 1 ? dly : dlz);
				dl0 = ax == 0 ? dly : dlx;
				dl1 = ax == 2 ? dly : dlz;
				if (nseg > qcap) {  // (cannot happen: a ray inside a grid of < 1024 cells per axis has at most ~100 segments)
					err |= ERR_GRID_OOB;
					continue;
				}
				pending = true;
			}
			// room for the ray's constants and its segments, or the next round
			const u32 slot = atomicAdd(&sh[0], 1u);
			u32 off = 0xFFFFFFFFu;
			if (slot < rcap) {
				off = atomicAdd(&sh[1], nseg);
				if (off + nseg > qcap) {
					atomicMin(&sh[2], off);  // (every entry from here on belongs to a ray that was refused)
					off = 0xFFFFFFFFu;
				}
			}
			if (0xFFFFFFFFu == off) break;
			pending = false;
			RayConst c;
			c.td[0] = r.td[0];
			c.td[1] = r.td[1];
			c.td[2] = r.td[2];
			c.dist = r.dist;
			c.dl[0] = (i32)r.s[0];
			c.dl[1] = (i32)r.s[1] * (i32)rowBits;
			c.dl[2] = (i32)r.s[2] * (i32)planeBits;
			c.glin = glin;
			rc[slot] = c;
			// the three chains (vol_kernels.h: k_vcut). a* = the dominant axis, b0 < b1 the two others; after k0 pops of a* element
			// A[k0 - 1] (= v) was popped and t_max_a* = A[k0]; of axis b the elements before v were popped -- strictly smaller, or
			// equal when b wins the tie (b < a*, vector3.h:244-251)
			double ta = ax == 0 ? r.tm[0] : (ax == 1 ? r.tm[1] : r.tm[2]), v = ta;
			const double tda = ax == 0 ? r.td[0] : (ax == 1 ? r.td[1] : r.td[2]);
			double t0 = ax == 0 ? r.tm[1] : r.tm[0], t1 = ax == 2 ? r.tm[1] : r.tm[2];
			const double d0 = ax == 0 ? r.td[1] : r.td[0], d1 = ax == 2 ? r.td[1] : r.td[2];
			const bool pri0 = ax != 0u, pri1 = ax == 2u;
			u32 n0 = 0, n1 = 0, guard = 0;
			auto advance = [&](double& tb, const double dbt, const bool pri, u32& cb) {
				for (;;) {  // four candidates per iteration (the same sequence of additions)
					const double q1 = tb + dbt, q2 = q1 + dbt, q3 = q2 + dbt;
					const bool e0 = pri ? (tb <= v) : (tb < v);
					const bool e1 = e0 & (pri ? (q1 <= v) : (q1 < v)), e2 = e1 & (pri ? (q2 <= v) : (q2 < v)), e3 = e2 & (pri ? (q3 <= v) : (q3 < v));
					if (e3) {
						tb = q3 + dbt;
						cb += 4u;
						if (++guard > 1024u) {
							err |= ERR_RUNAWAY;  // (cannot trip inside a grid of < 1024 cells per axis)
							break;
						}
						continue;
					}
					tb = e2 ? q3 : (e1 ? q2 : (e0 ? q1 : tb));
					cb += (e0 ? 1u : 0u) + (e1 ? 1u : 0u) + (e2 ? 1u : 0u);
					break;
				}
			};
			u32 lin = lin0;
			for (u32 j = 0; j < nseg; ++j) {
				if (j > 0u) {
					u32 np = w;
					for (; np >= 4u; np -= 4u) {  // (the same sequence of additions, four at a time)
						const double a1 = ta + tda, a2 = a1 + tda, a3 = a2 + tda;
						v = a3;
						ta = a3 + tda;
					}
					for (; np > 0u; --np) {
						v = ta;
						ta = ta + tda;
					}
					advance(t0, d0, pri0, n0);
					advance(t1, d1, pri1, n1);
					lin = lin0 + (u32)((i32)(j * w) * dla + (i32)n0 * dl0 + (i32)n1 * dl1);
					q[off + j - 1u].end = lin;  // the segment before ends where this one starts
				}
				SegRec rec;
				rec.tm[0] = ax == 0 ? ta : t0;
				rec.tm[1] = ax == 0 ? t0 : (ax == 1 ? ta : t1);
				rec.tm[2] = ax == 2 ? ta : t1;
				rec.lin = lin;
				rec.end = glin;
				rec.ray = slot | (0u == j ? 0x80000000u : 0u);
				rec.pad = 0;
				q[off + j] = rec;
			}
		}
		{
			__syncthreads();
			if (0 == (threadIdx.x | blockIdx.x) && 0 == round) ctl->dbg[35] = wall_clock64();  // (diagnostics)
			const u32 nsegs = min(min(sh[1], sh[2]), qcap);
			// ---- every lane walks segments ----
			for (u32 si = threadIdx.x; si < nsegs; si += blockDim.x) {
				const SegRec rec = q[si];
				const RayConst c = rc[rec.ray & 0x7FFFFFFFu];
				double tmx = rec.tm[0], tmy = rec.tm[1], tmz = rec.tm[2];
				const double tdx = c.td[0], tdy = c.td[1], tdz = c.td[2];
				const long long idist = __double_as_longlong(c.dist);
				const i32 dlx = c.dl[0], dly = c.dl[1], dlz = c.dl[2];
				const u32 end = rec.end;
				u32 lin = rec.lin;
				bool go = (0 != (rec.ray & 0x80000000u)) ||
				          ((lin != c.glin) && ((__double_as_longlong(tmx) <= idist) | (__double_as_longlong(tmy) <= idist) | (__double_as_longlong(tmz) <= idist)));
				u32 cnt = 0;
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
					const bool more = (__double_as_longlong(tmx) <= idist) | (__double_as_longlong(tmy) <= idist) | (__double_as_longlong(tmz) <= idist);
					go = (lin != end) & more & (cnt < 4096u);
				}
				if (cnt >= 4096u) err |= ERR_RUNAWAY;  // (a segment is ~K steps by construction)
				steps += cnt;  // (one cell marked per step taken. k_fcast derives the count from the cells' coordinates: six integer
				               // divisions per segment, which is why shorter segments made its WALK slower -- 14 us at K = 16 against 7)
			}
			const int more_rounds = __syncthreads_or((pending || pcur < pts) ? 1 : 0);  // (the queue and the counters are no longer read)
			if (!more_rounds) break;
			if (threadIdx.x < 3u) sh[threadIdx.x] = (2u == threadIdx.x) ? 0xFFFFFFFFu : 0u;
			__syncthreads();
		}
	}
	for (int o = 32; o > 0; o >>= 1) {
		nhit += __shfl_xor(nhit, o);
		ncast += __shfl_xor(ncast, o);
	}
	if (0 == lane && nhit) atomicAdd(&sh[4], nhit);
	if (0 == lane && ncast) atomicAdd(&sh[3], ncast);
	__syncthreads();
	if (0 == (threadIdx.x | blockIdx.x)) ctl->dbg[36] = wall_clock64();  // (diagnostics)
	if (0 == threadIdx.x) {
		// per-workgroup partials, folded by k_fmerge (256 workgroups adding to one word serialise at ~12 ns each)
		steps_part[gridDim.x + blockIdx.x] = sh[3];
		steps_part[2u * gridDim.x + blockIdx.x] = sh[4];
	}
	{
		const uint4* l4 = reinterpret_cast<const uint4*>(lds);
		uint4* out4 = reinterpret_cast<uint4*>(slabs) + (size_t)blockIdx.x * (lds_words >> 2);
		const u32 n4 = lds_words >> 2;
		for (u32 j = threadIdx.x; j < n4; j += blockDim.x) out4[j] = l4[j];
	}
	if (0 == (threadIdx.x | blockIdx.x)) ctl->dbg[37] = wall_clock64();  // (diagnostics)
	blockStoreSteps(steps, steps_part);
	if (oob) atomicAdd(&ctl->n_oob, oob);
	if (err) atomicOr(&ctl->err, err);
	if (0 == (threadIdx.x | blockIdx.x)) ctl->dbg[38] = wall_clock64();  // (diagnostics)
}

// ------------------------------------------------------------------------------------------------
// F2'' (round 6): the ray kernel of the steady state, rebuilt around what a DEPENDENT instruction costs a lone wave on this part
// (scripts/micro/chain_latency.hip: a dependent FP64 addition 11 clocks, a loop iteration 30, one DDA step of the walk 244-284 --
// whatever the occupancy: the step is one chain of compares, mask arithmetic and selects). k_fcast2's 25 us were three such chains,
// one behind the other, with the chip idle beside them:
//   * in discrete mode three points in four lose their voxel to an earlier point (OMB:358-360): every wave ran the ~1 000
//     instructions of set-up and cuts with a quarter of its lanes, twice (two points per lane). Here a first pass only LOOKS at the
//     points (record, first-point array) and appends the indices of the survivors to a list in LDS; after one barrier lane t takes
//     survivor t -- the rays of a workgroup (~190 of ~680 points) fill three waves;
//   * the cuts tested the minor axes' elements four at a time in a loop: ~350 clocks of dependent compares, selects and mask
//     arithmetic per iteration, and a wave runs as many iterations as its slowest lane -- 8 us. Now: a LOWER estimate of how many
//     elements precede the cut from one multiplication, that many plain additions (the same sequence of sums, 11 clocks each), a
//     check that the last element skipped really precedes the cut, the rest (one or two) one by one without a branch;
//   * queue room is reserved once per wave (returning LDS atomics of 64 lanes on one word run one lane after the other).
// Same cells, same step count: the cut states are the same sums (the 14 scheduling tests, the bench's self-check).
// ------------------------------------------------------------------------------------------------
template <bool DISCRETE>
__global__ __launch_bounds__(1024) void k_fcast3(MapGeom g, FastGeo fg, D3 sensor, u32 n, const u32* __restrict__ first, u32* __restrict__ slabs, u32 K,
                                                const ScanCtl* ctl_in, ScanCtl* ctl, unsigned long long* __restrict__ steps_part, const PointRec* __restrict__ recs,
                                                u32 rcap, u32 qcap, u32 lcap, u32 prio, Pipe* solo, ScanDesc solo_desc)
{
	if (solo && 0 == (threadIdx.x | blockIdx.x)) {
		solo->ring[0] = solo_desc;
		solo->slot[0].first = 0;
		solo->slot[0].B = 1;
	}
	if (prio >= 3u) __builtin_amdgcn_s_setprio(3);
	else if (2u == prio) __builtin_amdgcn_s_setprio(2);
	else if (1u == prio) __builtin_amdgcn_s_setprio(1);
	extern __shared__ __attribute__((aligned(16))) u32 lds[];
	const u32 err_in = ctl_in->err;  // (looked at once the LDS grid has been cleared: the load is in flight meanwhile)
	if (0 == (threadIdx.x | blockIdx.x)) ctl->dbg[30] = wall_clock64();  // (diagnostics)
	const Grid& gr = fg.gr;
	const u32 lds_words = (u32)(gr.bytes >> 2);
	RayConst* rc = reinterpret_cast<RayConst*>(lds + lds_words);
	SegRec* q = reinterpret_cast<SegRec*>(rc + rcap);
	u32* wl = reinterpret_cast<u32*>(q + qcap);  // the pass's survivors: point indices
	u32* sh = wl + lcap;  // [0] rays of the round, [1] queue entries asked for, [2] first entry that was refused, [3] rays cast, [4] voxels hit, [5] survivors
	                      // (lcap: points per pass -- the list holds their survivors; a workgroup with more points takes several passes)
	const u32 pts = (n > blockIdx.x) ? (n - blockIdx.x + gridDim.x - 1) / gridDim.x : 0u;
	const u32 lane = threadIdx.x & 63u;
	u32 nhit = 0;
	// the first pass's records are asked for before the grid is cleared
	PointRec pr0;
	pr0.flags = 0;
	pr0.cell = 0;
	if (threadIdx.x < pts && !err_in) pr0 = recs[blockIdx.x + threadIdx.x * gridDim.x];
	{
		uint4* l4 = reinterpret_cast<uint4*>(lds);
		for (u32 j = threadIdx.x; j < (lds_words >> 2); j += blockDim.x) l4[j] = make_uint4(0, 0, 0, 0);
	}
	if (threadIdx.x < 16u) sh[threadIdx.x] = (2u == threadIdx.x) ? 0xFFFFFFFFu : 0u;
	if (err_in) return;  // the scan does not fit the predicted grid (k_fhits): it will be repeated (uniform exit)
	__syncthreads();
	if (0 == (threadIdx.x | blockIdx.x)) ctl->dbg[31] = wall_clock64();  // (diagnostics)
	const u32 rowBits = fg.rowBits, planeBits = fg.planeBits;
	const u32 lim = 1u << g.L;
	unsigned long long steps = 0;
	u32 err = 0, oob = 0, ncast = 0;
	for (u32 p0 = 0; p0 < pts; p0 += lcap) {  // (uniform; the steady state: one pass)
		// ---- who casts a ray: the points p0 .. p0 + lcap - 1 of this workgroup (blockIdx.x + p * gridDim.x) ----
		const u32 pend = min(pts, p0 + lcap);
		for (u32 pb = p0; pb < pend; pb += blockDim.x) {  // (uniform)
			const u32 p = pb + threadIdx.x;
			bool cast = false;
			u32 i = 0;
			if (p < pend) {
				i = blockIdx.x + p * gridDim.x;
				const PointRec pr = (0 == pb) ? pr0 : recs[i];  // (k_fhits ran the head loop on the point)
				const bool odd = 0 != (pr.flags & 4u);
				cast = (pr.flags & 1u) && !odd;
				if ((pr.flags & 2u) && !odd) {
					// (the voxel receives a hit, OMB:295, 358-360: its first point's; k_fmerge derives the hit grid from the array)
					const bool winner = first[pr.cell] == i;
					nhit += winner ? 1u : 0u;
					if (DISCRETE && !winner) cast = false;  // OMB:358-360: dropped entirely, no ray
				}
			}
			const u64 m = __ballot(cast);
			u32 base = 0;
			if (0 == lane && m) base = atomicAdd(&sh[5], (u32)__popcll(m));
			base = __shfl(base, 0);
			if (cast) wl[base + (u32)__popcll(m & ((1ULL << lane) - 1ULL))] = i;
		}
		__syncthreads();
		const u32 nw = sh[5];
		if (0 == (threadIdx.x | blockIdx.x) && 0 == p0) ctl->dbg[32] = wall_clock64();  // (diagnostics)
		// ---- lane t: survivors t, t + blockDim, ... -- one per round; a ray that finds the queue full stays with its lane until the next ----
		u32 wcur = threadIdx.x;
		bool pending = false;
		RayState r;
		r.status = 0;
		u32 ax = 0, w = 1, nseg = 0, lin0 = 0, glin = 0;
		i32 dla = 0, dl0 = 0, dl1 = 0;
		for (u32 round = 0;; ++round) {  // (uniform: a round = set-up and cuts, barrier, walk, barrier)
			const unsigned long long tq0 = clock64();
			if (!pending && wcur < nw) {
				const u32 i = wl[wcur];
				wcur += blockDim.x;
				const D3 end = recs[i].end;
				++ncast;
				raySetup(g, sensor, 0u, gr, end, r);
				if (1 == r.status) {
					err |= markBitChecked(gr, lds, rowBits, planeBits, r.start[0], r.start[1], r.start[2], lim, &oob);
					steps += 1;
				} else if (3 == r.status) {
					err |= ERR_GRID_OOB;  // cannot happen: pointRay admits only rays inside the grid's interior
				} else if (2 == r.status) {
					const u32 dxn = (u32)abs((i32)(r.gpk & 1023u) - (i32)(r.pk0 & 1023u));
					const u32 dyn = (u32)abs((i32)((r.gpk >> 10) & 1023u) - (i32)((r.pk0 >> 10) & 1023u));
					const u32 dzn = (u32)abs((i32)(r.gpk >> 20) - (i32)(r.pk0 >> 20));
					ax = (dxn >= dyn && dxn >= dzn) ? 0u : (dyn >= dzn ? 1u : 2u);
					const u32 dmax = ax == 0 ? dxn : (ax == 1 ? dyn : dzn);
					const u32 l1 = dxn + dyn + dzn;
					w = (u32)(((u64)dmax * K) / l1);
					if (w < 1u) w = 1u;
					nseg = (dmax + w - 1u) / w;  // >= 1 (start and goal differ)
					lin0 = pkToLin(r.pk0, rowBits, planeBits);
					glin = pkToLin(r.gpk, rowBits, planeBits);
					const i32 dlx = (i32)r.s[0], dly = (i32)r.s[1] * (i32)rowBits, dlz = (i32)r.s[2] * (i32)planeBits;
					dla = ax == 0 ? dlx : (ax == 1 ? dly : dlz);
					dl0 = ax == 0 ? dly : dlx;
					dl1 = ax == 2 ? dly : dlz;
					if (nseg > qcap) err |= ERR_GRID_OOB;  // (cannot happen: a ray inside a grid of < 1024 cells per axis has at most ~100 segments)
					else pending = true;
				}
			}
			const unsigned long long tq1 = clock64() + (unsigned long long)(77 == r.status ? 1 : 0);
			// room for the ray's constants and its segments, or the next round: ONE reservation per wave -- every lane of the wave is here
			u32 slot = 0xFFFFFFFFu, off = 0xFFFFFFFFu;
			{
				const u64 am = __ballot(pending);
				u32 slot0 = 0;
				if (0 == lane && am) slot0 = atomicAdd(&sh[0], (u32)__popcll(am));
				slot0 = __shfl(slot0, 0);
				if (pending) slot = slot0 + (u32)__popcll(am & ((1ULL << lane) - 1ULL));
				const u32 want = (pending && slot < rcap) ? nseg : 0u;  // (a ray without room for its constants asks for no queue entries)
				u32 scan = want;
				for (int o = 1; o < 64; o <<= 1) {
					const u32 up = __shfl_up(scan, o);
					if ((int)lane >= o) scan += up;
				}
				const u32 total = __shfl(scan, 63);
				u32 off0 = 0;
				if (0 == lane && total) off0 = atomicAdd(&sh[1], total);
				off0 = __shfl(off0, 0);
				if (want) {
					off = off0 + scan - want;
					if (off + nseg > qcap) {
						atomicMin(&sh[2], off);  // (every entry from here on belongs to a ray that was refused)
						off = 0xFFFFFFFFu;
					}
				}
			}
			if (0xFFFFFFFFu != off) {
				pending = false;
				RayConst c;
				c.td[0] = r.td[0];
				c.td[1] = r.td[1];
				c.td[2] = r.td[2];
				c.dist = r.dist;
				c.dl[0] = (i32)r.s[0];
				c.dl[1] = (i32)r.s[1] * (i32)rowBits;
				c.dl[2] = (i32)r.s[2] * (i32)planeBits;
				c.glin = glin;
				rc[slot] = c;
				// the three chains (k_fcast2 / vol_kernels.h: k_vcut). a* = the dominant axis, b0 < b1 the two others; after k0 pops of a* element
				// A[k0 - 1] (= v) was popped and t_max_a* = A[k0]; of axis b the elements before v were popped -- strictly smaller, or
				// equal when b wins the tie (b < a*, vector3.h:244-251)
				double ta = ax == 0 ? r.tm[0] : (ax == 1 ? r.tm[1] : r.tm[2]), v = ta;
				const double tda = ax == 0 ? r.td[0] : (ax == 1 ? r.td[1] : r.td[2]);
				double t0 = ax == 0 ? r.tm[1] : r.tm[0], t1 = ax == 2 ? r.tm[1] : r.tm[2];
				const double d0 = ax == 0 ? r.td[1] : r.td[0], d1 = ax == 2 ? r.td[1] : r.td[2];
				const bool pri0 = ax != 0u, pri1 = ax == 2u;
				u32 n0 = 0, n1 = 0, guard = 0;
				auto advance = [&](double& tb, const double dbt, const double inv, const bool pri, u32& cb) {
					const double est = (v - tb) * inv;  // (negative, NaN or huge for an axis the ray does not move along: no skip)
					u32 kk = (est > 2.0 && est < 4096.0) ? (u32)est - 1u : 0u;  // an element of margin: repeated rounding moves a sum by a few ulp, not by an element -- and the check below decides
					if (kk) {
						const u32 k0 = kk;
						double sv = tb, prev = tb;
						for (; kk >= 4u; kk -= 4u) {
							const double s1 = sv + dbt, s2 = s1 + dbt, s3 = s2 + dbt;
							prev = s3;
							sv = s3 + dbt;
						}
						for (; kk > 0u; --kk) {
							prev = sv;
							sv = sv + dbt;
						}
						if (pri ? (prev <= v) : (prev < v)) {  // (else: the estimate was too high -- everything one by one, below)
							tb = sv;
							cb += k0;
						}
					}
					bool e = true;
#pragma unroll
					for (int u = 0; u < 3; ++u) {
						e = e & (pri ? (tb <= v) : (tb < v));
						const double nt = tb + dbt;
						tb = e ? nt : tb;
						cb += e ? 1u : 0u;
					}
					while (e) {
						e = pri ? (tb <= v) : (tb < v);
						if (e) {
							tb = tb + dbt;
							cb += 1u;
							if (++guard > 4096u) {
								err |= ERR_RUNAWAY;  // (cannot trip inside a grid of < 1024 cells per axis)
								break;
							}
						}
					}
				};
				const double i0 = 1.0 / d0, i1 = 1.0 / d1;
				u32 lin = lin0;
				for (u32 j = 0; j < nseg; ++j) {
					if (j > 0u) {
						u32 np = w;
						for (; np >= 4u; np -= 4u) {  // (the same sequence of additions, four at a time)
							const double a1 = ta + tda, a2 = a1 + tda, a3 = a2 + tda;
							v = a3;
							ta = a3 + tda;
						}
						for (; np > 0u; --np) {
							v = ta;
							ta = ta + tda;
						}
						advance(t0, d0, i0, pri0, n0);
						advance(t1, d1, i1, pri1, n1);
						lin = lin0 + (u32)((i32)(j * w) * dla + (i32)n0 * dl0 + (i32)n1 * dl1);
						q[off + j - 1u].end = lin;  // the segment before ends where this one starts
					}
					SegRec rec;
					rec.tm[0] = ax == 0 ? ta : t0;
					rec.tm[1] = ax == 0 ? t0 : (ax == 1 ? ta : t1);
					rec.tm[2] = ax == 2 ? ta : t1;
					rec.lin = lin;
					rec.end = glin;
					rec.ray = slot | (0u == j ? 0x80000000u : 0u);
					rec.pad = 0;
					q[off + j] = rec;
				}
			}
			if (0 == round && 0 == p0 && 0 == blockIdx.x) {  // (diagnostics: the slowest lane's clocks in set-up and cuts, workgroup 0)
				u32 d0c = (u32)(tq1 - tq0), d1c = (u32)(clock64() - tq1);
				for (int o = 32; o > 0; o >>= 1) {
					d0c = max(d0c, (u32)__shfl_xor((int)d0c, o));
					d1c = max(d1c, (u32)__shfl_xor((int)d1c, o));
				}
				if (0 == lane) {
					atomicMax(&sh[8], d0c);
					atomicMax(&sh[9], d1c);
				}
			}
			__syncthreads();
			if (0 == (threadIdx.x | blockIdx.x) && 0 == round && 0 == p0) {  // (diagnostics)
				ctl->dbg[35] = wall_clock64();
				ctl->dbg[40] = sh[8];
				ctl->dbg[41] = sh[9];
			}
			const u32 nsegs = min(min(sh[1], sh[2]), qcap);
			// ---- every lane walks segments ----
			const unsigned long long tw0 = clock64();
			u32 wmaxc = 0;
			for (u32 si = threadIdx.x; si < nsegs; si += blockDim.x) {
				const SegRec rec = q[si];
				const RayConst c = rc[rec.ray & 0x7FFFFFFFu];
				double tmx = rec.tm[0], tmy = rec.tm[1], tmz = rec.tm[2];
				const double tdx = c.td[0], tdy = c.td[1], tdz = c.td[2];
				const long long idist = __double_as_longlong(c.dist);
				const i32 dlx = c.dl[0], dly = c.dl[1], dlz = c.dl[2];
				const u32 end = rec.end;
				u32 lin = rec.lin;
				bool go = (0 != (rec.ray & 0x80000000u)) ||
				          ((lin != c.glin) && ((__double_as_longlong(tmx) <= idist) | (__double_as_longlong(tmy) <= idist) | (__double_as_longlong(tmz) <= idist)));
				u32 cnt = 0;
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
					const bool more = (__double_as_longlong(tmx) <= idist) | (__double_as_longlong(tmy) <= idist) | (__double_as_longlong(tmz) <= idist);
					go = (lin != end) & more & (cnt < 4096u);
				}
				if (cnt >= 4096u) err |= ERR_RUNAWAY;  // (a segment is ~K steps by construction)
				steps += cnt;
				wmaxc = max(wmaxc, cnt);
			}
			if (0 == round && 0 == p0 && 0 == blockIdx.x) {  // (diagnostics)
				u32 dwc = (u32)(clock64() - tw0);
				for (int o = 32; o > 0; o >>= 1) {
					dwc = max(dwc, (u32)__shfl_xor((int)dwc, o));
					wmaxc = max(wmaxc, (u32)__shfl_xor((int)wmaxc, o));
				}
				if (0 == lane) {
					atomicMax(&sh[10], dwc);
					atomicMax(&sh[11], wmaxc);
					sh[12] = nsegs;
				}
			}
			const int more_rounds = __syncthreads_or((pending || wcur < nw) ? 1 : 0);  // (the queue and the counters are no longer read)
			if (threadIdx.x < 3u) sh[threadIdx.x] = (2u == threadIdx.x) ? 0xFFFFFFFFu : 0u;
			if (!more_rounds) break;
			__syncthreads();
		}
		if (0 == threadIdx.x) sh[5] = 0;
		__syncthreads();  // (the list and the counters are free for the next pass)
	}
	for (int o = 32; o > 0; o >>= 1) {
		nhit += __shfl_xor(nhit, o);
		ncast += __shfl_xor(ncast, o);
	}
	if (0 == lane && nhit) atomicAdd(&sh[4], nhit);
	if (0 == lane && ncast) atomicAdd(&sh[3], ncast);
	__syncthreads();
	if (0 == (threadIdx.x | blockIdx.x)) {  // (diagnostics)
		ctl->dbg[36] = wall_clock64();
		ctl->dbg[42] = sh[10];
		ctl->dbg[43] = sh[11];
		ctl->dbg[39] = sh[12];
	}
	if (0 == threadIdx.x) {
		// per-workgroup partials, folded by k_fmerge (256 workgroups adding to one word serialise at ~12 ns each)
		steps_part[gridDim.x + blockIdx.x] = sh[3];
		steps_part[2u * gridDim.x + blockIdx.x] = sh[4];
	}
	{
		const uint4* l4 = reinterpret_cast<const uint4*>(lds);
		uint4* out4 = reinterpret_cast<uint4*>(slabs) + (size_t)blockIdx.x * (lds_words >> 2);
		const u32 n4 = lds_words >> 2;
		for (u32 j = threadIdx.x; j < n4; j += blockDim.x) out4[j] = l4[j];
	}
	if (0 == (threadIdx.x | blockIdx.x)) ctl->dbg[37] = wall_clock64();  // (diagnostics)
	blockStoreSteps(steps, steps_part);
	if (oob) atomicAdd(&ctl->n_oob, oob);
	if (err) atomicOr(&ctl->err, err);
	if (0 == (threadIdx.x | blockIdx.x)) ctl->dbg[38] = wall_clock64();  // (diagnostics)
}

// ------------------------------------------------------------------------------------------------
// F2''' (round 6): OCTANT SUB-BOXES. Every ray of a scan starts at the sensor, so the cells of a ray lie in the box spanned by the
// sensor's cell and the ray's end cell -- inside ONE of the eight sub-boxes the sensor's cell cuts the ray grid into (the sensor's
// row / plane belongs to both sides; x in whole 32-bit words). k_fcast2/3 keep the WHOLE grid in a workgroup's LDS (82 KB of 148:
// one workgroup per CU) and hand 192 copies of it to the tree update: 20 MB written and read back for 0.1 MB of grid -- the largest
// waste of the scan. Here k_fhits leaves the points' records sorted by octant (per 256-point stretch, with the
// stretch's eight counts), a workgroup of the ray kernel takes rays of one octant and keeps only that octant's sub-box in LDS
// (10-20 KB: two or three workgroups per CU, each with its own segment queue), and what it hands over is the sub-box: a few MB per scan.
//   * which workgroups take which octant is decided IN the kernel, from the scan's own counts (every workgroup adds up the stretches'
//     counts -- 8 KB -- and splits the launch in proportion, at least one workgroup per octant that has points): a LiDAR looks down,
//     its upper four octants hold 7 % of the rays; workgroup 0 leaves the table (first workgroup and slab offset per octant) for k_fmerge;
//   * workgroup j of an octant's n takes the stretches j, j + n, ...: their records of the octant are contiguous, the first pass looks
//     at them and lists the survivors (k_fcast3), then set-up, cuts, queue, walk as there -- with the sub-box's strides.
// Same cells, same step count.
// ------------------------------------------------------------------------------------------------
// The launch's G workgroups split over the octants in proportion to their work (tot: k_fhits' sums of squared ray lengths), at least one
// per octant that has any; the same arithmetic in every workgroup.
__device__ inline void octAllocate(const unsigned long long tot[8], u32 G, const OctGeo& og, u32 wg_start[9], u32 slab_off4[9])
{
	unsigned long long T = 0;
	u32 nzo = 0, big = 0;
	for (u32 o = 0; o < 8u; ++o) {
		T += tot[o];
		nzo += tot[o] ? 1u : 0u;
		if (tot[o] > tot[big]) big = o;
	}
	u32 nw[8], sum = 0;
	for (u32 o = 0; o < 8u; ++o) {
		nw[o] = tot[o] ? 1u + (u32)((double)(G - nzo) * ((double)tot[o] / (double)T)) : 0u;  // (rounded down: what is left goes to the largest)
		sum += nw[o];
	}
	if (T) nw[big] += G - sum;  // (what the rounding left)
	u32 ws = 0, so = 0;
	for (u32 o = 0; o < 8u; ++o) {
		wg_start[o] = ws;
		slab_off4[o] = so;
		ws += nw[o];
		so += nw[o] * octWords4(og, o);
	}
	wg_start[8] = ws;
	slab_off4[8] = so;
}
template <bool DISCRETE>
__global__ __launch_bounds__(512) void k_fcast4(MapGeom g, FastGeo fg, D3 sensor, OctGeo og, u32 nsrc, const uint4* __restrict__ bcnt, const u32* __restrict__ bwgt,
                                               const PointRec* __restrict__ brecs, const u32* __restrict__ first, uint4* __restrict__ slabs, OctTab* __restrict__ tab, u32 K,
                                               const ScanCtl* ctl_in, ScanCtl* ctl, unsigned long long* __restrict__ steps_part, u32 rcap, u32 qcap, u32 lcap, u32 prio,
                                               Pipe* solo, ScanDesc solo_desc)
{
	if (solo && 0 == (threadIdx.x | blockIdx.x)) {
		solo->ring[0] = solo_desc;
		solo->slot[0].first = 0;
		solo->slot[0].B = 1;
	}
	if (prio >= 3u) __builtin_amdgcn_s_setprio(3);
	else if (2u == prio) __builtin_amdgcn_s_setprio(2);
	else if (1u == prio) __builtin_amdgcn_s_setprio(1);
	extern __shared__ __attribute__((aligned(16))) u32 lds[];
	const u32 err_in = ctl_in->err;  // (looked at once the LDS grid has been cleared: the load is in flight meanwhile)
	if (0 == (threadIdx.x | blockIdx.x)) ctl->dbg[30] = wall_clock64();  // (diagnostics)
	const Grid& gr = fg.gr;
	const u32 lds_words = og.wmax4 * 4u;
	RayConst* rc = reinterpret_cast<RayConst*>(lds + lds_words);
	SegRec* q = reinterpret_cast<SegRec*>(rc + rcap);
	u32* wl = reinterpret_cast<u32*>(q + qcap);  // the window's survivors: indices of their records
	u32* srcb = wl + lcap;                        // per stretch of the cloud: where the octant's records start ...
	u32* srcp = srcb + ((nsrc + 2u) & ~1u);       // ... and how many records of the octant lie before the stretch ([nsrc]: all of them)
	unsigned long long* wtot = reinterpret_cast<unsigned long long*>(srcp + ((nsrc + 2u) & ~1u));  // [8] the octants' work
	u32* sh = reinterpret_cast<u32*>(wtot + 8);  // [0] rays of the round, [1] queue entries asked for, [2] first entry refused, [3] rays cast, [4] voxels hit,
	                                             // [5] survivors, [6] running prefix, [8..12] diagnostics, [24..39] wave totals of the prefix
	const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
	{
		uint4* l4 = reinterpret_cast<uint4*>(lds);
		for (u32 j4 = threadIdx.x; j4 < og.wmax4; j4 += blockDim.x) l4[j4] = make_uint4(0, 0, 0, 0);
	}
	if (threadIdx.x < 40u) sh[threadIdx.x] = (2u == threadIdx.x) ? 0xFFFFFFFFu : 0u;
	if (threadIdx.x < 8u) wtot[threadIdx.x] = 0ull;
	if (err_in) return;  // the scan does not fit the predicted grid (k_fhits): it will be repeated (uniform exit)
	__syncthreads();
	// ---- the octants' work (k_fhits: sums of squared ray lengths per stretch and octant) ----
	for (u32 w0 = 0; w0 < nsrc; w0 += blockDim.x) {  // (uniform)
		const u32 wv = w0 + threadIdx.x;
		if (wv < nsrc) {
			const uint4 wa = reinterpret_cast<const uint4*>(bwgt)[2u * wv], wb = reinterpret_cast<const uint4*>(bwgt)[2u * wv + 1u];
			const u32 v8[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
			for (u32 o = 0; o < 8u; ++o)
				if (v8[o]) atomicAdd(&wtot[o], (unsigned long long)v8[o]);
		}
	}
	__syncthreads();
	if (0 == (threadIdx.x | blockIdx.x)) ctl->dbg[31] = wall_clock64();  // (diagnostics)
	u32 wg_start[9], slab_off4[9];
	{
		unsigned long long tot[8];
		for (u32 o = 0; o < 8u; ++o) tot[o] = wtot[o];
		octAllocate(tot, gridDim.x, og, wg_start, slab_off4);
	}
	if (0 == blockIdx.x && threadIdx.x < 9u) {
		if (0 == threadIdx.x) tab->og = og;
		tab->wg_start[threadIdx.x] = wg_start[threadIdx.x];
		tab->slab_off4[threadIdx.x] = slab_off4[threadIdx.x];
	}
	u32 oct = 0;
	for (u32 o = 1; o < 8u; ++o)
		if (blockIdx.x >= wg_start[o]) oct = o;  // (the last octant whose first workgroup is at or below this one: empty octants share their successor's start)
	const u32 jwg = blockIdx.x - wg_start[oct], nwg = wg_start[oct + 1u] - wg_start[oct];
	// the sub-box: a cell's bit is (x - 32 xw0) + rowBits ((y - y0) + ny (z - z0))
	const u32 ox = oct & 1u, oy = (oct >> 1) & 1u, oz = oct >> 2;
	const u32 bx0 = 32u * og.xw0[ox], rowBits = 32u * og.nxw[ox], by0 = og.y0[oy], bny = og.ny[oy], bz0 = og.z0[oz], bnz = og.nz[oz];
	const u32 planeBits = rowBits * bny;
	auto subLin = [&](u32 x, u32 y, u32 z) -> u32 {  // (0xFFFFFFFF: outside the sub-box)
		const u32 dx = x - bx0, dy = y - by0, dz = z - bz0;
		return (dx < rowBits && dy < bny && dz < bnz) ? dx + rowBits * (dy + bny * dz) : 0xFFFFFFFFu;
	};
	// ---- the octant's records across the cloud: stretch wv holds cnt of them from record 256 wv + st; srcp = how many lie before ----
	for (u32 w0 = 0; w0 < nsrc; w0 += blockDim.x) {  // (uniform)
		const u32 wv = w0 + threadIdx.x;
		u32 cnt_e = 0, st = 0;
		if (wv < nsrc) {
			const uint4 c = bcnt[wv];
			const u32 v8[8] = {c.x & 0xFFFFu, c.x >> 16, c.y & 0xFFFFu, c.y >> 16, c.z & 0xFFFFu, c.z >> 16, c.w & 0xFFFFu, c.w >> 16};
#pragma unroll
			for (u32 o = 0; o < 8u; ++o) {
				if (o < oct) st += v8[o];
				if (o == oct) cnt_e = v8[o];
			}
			srcb[wv] = wv * 256u + st;
		}
		u32 scan = cnt_e;
		for (int o2 = 1; o2 < 64; o2 <<= 1) {
			const u32 up = __shfl_up(scan, o2);
			if ((int)lane >= o2) scan += up;
		}
		if (63u == lane) sh[24u + wave] = scan;
		__syncthreads();
		u32 before = sh[6], total = 0;
		for (u32 wv2 = 0; wv2 < nwaves; ++wv2) {
			const u32 v = sh[24u + wv2];
			if (wv2 < wave) before += v;
			total += v;
		}
		if (wv < nsrc) srcp[wv] = before + scan - cnt_e;
		__syncthreads();
		if (0 == threadIdx.x) sh[6] += total;
	}
	__syncthreads();
	const u32 npts_oct = sh[6];
	if (0 == threadIdx.x) srcp[nsrc] = npts_oct;
	unsigned long long steps = 0;
	u32 err = 0, oob = 0, ncast = 0, nhit = 0;
	bool first_pass = true;
	// this workgroup's records: jwg, jwg + nwg, ... of the octant's -- neighbouring rays go to different workgroups, every workgroup
	// sees the octant's mix of lengths (as k_fcast2 / 3 take every gridDim-th point of the cloud)
	const u32 nmine = (nwg && npts_oct > jwg) ? (npts_oct - jwg + nwg - 1u) / nwg : 0u;
	{
		for (u32 f0 = 0; f0 < nmine; f0 += lcap) {  // (uniform; the steady state: one window)
			// ---- who casts a ray: this workgroup's records f0 .. f0 + lcap - 1 ----
			const u32 fend = min(nmine, f0 + lcap);
			for (u32 fb = f0; fb < fend; fb += blockDim.x) {  // (uniform)
				const u32 mf = fb + threadIdx.x;
				bool cast = false;
				u32 ri = 0;
				if (mf < fend) {
					const u32 kf = jwg + mf * nwg;
					// the stretch that holds record kf of the octant: the last one with srcp <= kf
					u32 lo = 0, hi = nsrc;
					while (hi - lo > 1u) {
						const u32 mid = (lo + hi) >> 1;
						if (srcp[mid] <= kf) lo = mid;
						else hi = mid;
					}
					ri = srcb[lo] + (kf - srcp[lo]);
					const PointRec pr = brecs[ri];  // (k_fhits ran the head loop on the point: a record is a point that may cast a ray)
					const u32 i = pr.flags >> 3;
					cast = true;
					if (pr.flags & 2u) {
						// (the voxel receives a hit, OMB:295, 358-360: its first point's; k_fmerge derives the hit grid from the array)
						const bool winner = first[pr.cell] == i;
						nhit += winner ? 1u : 0u;
						if (DISCRETE && !winner) cast = false;  // OMB:358-360: dropped entirely, no ray
					}
				}
				const u64 mb = __ballot(cast);
				u32 base = 0;
				if (0 == lane && mb) base = atomicAdd(&sh[5], (u32)__popcll(mb));
				base = __shfl(base, 0);
				if (cast) wl[base + (u32)__popcll(mb & ((1ULL << lane) - 1ULL))] = ri;
			}
			__syncthreads();
			const u32 nw = sh[5];
			if (0 == (threadIdx.x | blockIdx.x) && first_pass) ctl->dbg[32] = wall_clock64();  // (diagnostics)
			// ---- lane t: survivors t, t + blockDim, ... -- one per round; a ray that finds the queue full stays with its lane until the next ----
			u32 wcur = threadIdx.x;
			bool pending = false;
			RayState r;
			r.status = 0;
			u32 ax = 0, w = 1, nseg = 0, lin0 = 0, glin = 0;
			i32 dla = 0, dl0 = 0, dl1 = 0;
			for (u32 round = 0;; ++round) {  // (uniform: a round = set-up and cuts, barrier, walk, barrier)
				const unsigned long long tq0 = clock64();
				if (!pending && wcur < nw) {
					const u32 ri = wl[wcur];
					wcur += blockDim.x;
					const D3 end = brecs[ri].end;
					++ncast;
					raySetup(g, sensor, 0u, gr, end, r);
					if (1 == r.status) {
						// (start == goal: the sensor's own cell, which every octant's sub-box holds)
						const u32 sl = subLin((u32)(r.start[0] - gr.base[0]), (u32)(r.start[1] - gr.base[1]), (u32)(r.start[2] - gr.base[2]));
						if (0xFFFFFFFFu == sl) err |= ERR_GRID_OOB;
						else atomicOr(&lds[sl >> 5], 1u << (sl & 31u));
						steps += 1;
					} else if (3 == r.status) {
						err |= ERR_GRID_OOB;  // cannot happen: pointRay admits only rays inside the grid's interior
					} else if (2 == r.status) {
						const u32 dxn = (u32)abs((i32)(r.gpk & 1023u) - (i32)(r.pk0 & 1023u));
						const u32 dyn = (u32)abs((i32)((r.gpk >> 10) & 1023u) - (i32)((r.pk0 >> 10) & 1023u));
						const u32 dzn = (u32)abs((i32)(r.gpk >> 20) - (i32)(r.pk0 >> 20));
						ax = (dxn >= dyn && dxn >= dzn) ? 0u : (dyn >= dzn ? 1u : 2u);
						const u32 dmax = ax == 0 ? dxn : (ax == 1 ? dyn : dzn);
						const u32 l1 = dxn + dyn + dzn;
						w = (u32)(((u64)dmax * K) / l1);
						if (w < 1u) w = 1u;
						nseg = (dmax + w - 1u) / w;  // >= 1 (start and goal differ)
						lin0 = subLin(r.pk0 & 1023u, (r.pk0 >> 10) & 1023u, r.pk0 >> 20);
						glin = subLin(r.gpk & 1023u, (r.gpk >> 10) & 1023u, r.gpk >> 20);
						if (0xFFFFFFFFu == lin0 || 0xFFFFFFFFu == glin) {  // (cannot happen: k_fhits sorted the ray into the octant of its end cell)
							err |= ERR_GRID_OOB;
							nseg = qcap + 1u;
						}
						const i32 dlx = (i32)r.s[0], dly = (i32)r.s[1] * (i32)rowBits, dlz = (i32)r.s[2] * (i32)planeBits;
						dla = ax == 0 ? dlx : (ax == 1 ? dly : dlz);
						dl0 = ax == 0 ? dly : dlx;
						dl1 = ax == 2 ? dly : dlz;
						if (nseg > qcap) err |= ERR_GRID_OOB;  // (cannot happen: a ray inside a grid of < 1024 cells per axis has at most ~100 segments)
						else pending = true;
					}
				}
				const unsigned long long tq1 = clock64() + (unsigned long long)(77 == r.status ? 1 : 0);
				// room for the ray's constants and its segments, or the next round: ONE reservation per wave -- every lane of the wave is here
				u32 slot = 0xFFFFFFFFu, off = 0xFFFFFFFFu;
				{
					const u64 am = __ballot(pending);
					u32 slot0 = 0;
					if (0 == lane && am) slot0 = atomicAdd(&sh[0], (u32)__popcll(am));
					slot0 = __shfl(slot0, 0);
					if (pending) slot = slot0 + (u32)__popcll(am & ((1ULL << lane) - 1ULL));
					const u32 want = (pending && slot < rcap) ? nseg : 0u;  // (a ray without room for its constants asks for no queue entries)
					u32 scan = want;
					for (int o = 1; o < 64; o <<= 1) {
						const u32 up = __shfl_up(scan, o);
						if ((int)lane >= o) scan += up;
					}
					const u32 total = __shfl(scan, 63);
					u32 off0 = 0;
					if (0 == lane && total) off0 = atomicAdd(&sh[1], total);
					off0 = __shfl(off0, 0);
					if (want) {
						off = off0 + scan - want;
						if (off + nseg > qcap) {
							atomicMin(&sh[2], off);  // (every entry from here on belongs to a ray that was refused)
							off = 0xFFFFFFFFu;
						}
					}
				}
				if (0xFFFFFFFFu != off) {
					pending = false;
					RayConst c;
					c.td[0] = r.td[0];
					c.td[1] = r.td[1];
					c.td[2] = r.td[2];
					c.dist = r.dist;
					c.dl[0] = (i32)r.s[0];
					c.dl[1] = (i32)r.s[1] * (i32)rowBits;
					c.dl[2] = (i32)r.s[2] * (i32)planeBits;
					c.glin = glin;
					rc[slot] = c;
					// the three chains (k_fcast2 / vol_kernels.h: k_vcut). a* = the dominant axis, b0 < b1 the two others; after k0 pops of a* element
					// A[k0 - 1] (= v) was popped and t_max_a* = A[k0]; of axis b the elements before v were popped -- strictly smaller, or
					// equal when b wins the tie (b < a*, vector3.h:244-251)
					double ta = ax == 0 ? r.tm[0] : (ax == 1 ? r.tm[1] : r.tm[2]), v = ta;
					const double tda = ax == 0 ? r.td[0] : (ax == 1 ? r.td[1] : r.td[2]);
					double t0 = ax == 0 ? r.tm[1] : r.tm[0], t1 = ax == 2 ? r.tm[1] : r.tm[2];
					const double d0 = ax == 0 ? r.td[1] : r.td[0], d1 = ax == 2 ? r.td[1] : r.td[2];
					const bool pri0 = ax != 0u, pri1 = ax == 2u;
					u32 n0 = 0, n1 = 0, guard = 0;
					auto advance = [&](double& tb, const double dbt, const double inv, const bool pri, u32& cb) {
						const double est = (v - tb) * inv;  // (negative, NaN or huge for an axis the ray does not move along: no skip)
						u32 kk = (est > 2.0 && est < 4096.0) ? (u32)est - 1u : 0u;  // an element of margin: repeated rounding moves a sum by a few ulp, not by an element -- and the check below decides
						if (kk) {
							const u32 k0 = kk;
							double sv = tb, prev = tb;
							for (; kk >= 4u; kk -= 4u) {
								const double s1 = sv + dbt, s2 = s1 + dbt, s3 = s2 + dbt;
								prev = s3;
								sv = s3 + dbt;
							}
							for (; kk > 0u; --kk) {
								prev = sv;
								sv = sv + dbt;
							}
							if (pri ? (prev <= v) : (prev < v)) {  // (else: the estimate was too high -- everything one by one, below)
								tb = sv;
								cb += k0;
							}
						}
						bool e = true;
	#pragma unroll
						for (int u = 0; u < 3; ++u) {
							e = e & (pri ? (tb <= v) : (tb < v));
							const double nt = tb + dbt;
							tb = e ? nt : tb;
							cb += e ? 1u : 0u;
						}
						while (e) {
							e = pri ? (tb <= v) : (tb < v);
							if (e) {
								tb = tb + dbt;
								cb += 1u;
								if (++guard > 4096u) {
									err |= ERR_RUNAWAY;  // (cannot trip inside a grid of < 1024 cells per axis)
									break;
								}
							}
						}
					};
					const double i0 = 1.0 / d0, i1 = 1.0 / d1;
					u32 lin = lin0;
					for (u32 j = 0; j < nseg; ++j) {
						if (j > 0u) {
							u32 np = w;
							for (; np >= 4u; np -= 4u) {  // (the same sequence of additions, four at a time)
								const double a1 = ta + tda, a2 = a1 + tda, a3 = a2 + tda;
								v = a3;
								ta = a3 + tda;
							}
							for (; np > 0u; --np) {
								v = ta;
								ta = ta + tda;
							}
							advance(t0, d0, i0, pri0, n0);
							advance(t1, d1, i1, pri1, n1);
							lin = lin0 + (u32)((i32)(j * w) * dla + (i32)n0 * dl0 + (i32)n1 * dl1);
							q[off + j - 1u].end = lin;  // the segment before ends where this one starts
						}
						SegRec rec;
						rec.tm[0] = ax == 0 ? ta : t0;
						rec.tm[1] = ax == 0 ? t0 : (ax == 1 ? ta : t1);
						rec.tm[2] = ax == 2 ? ta : t1;
						rec.lin = lin;
						rec.end = glin;
						rec.ray = slot | (0u == j ? 0x80000000u : 0u);
						rec.pad = 0;
						q[off + j] = rec;
					}
				}
				if (0 == round && first_pass && 0 == blockIdx.x) {  // (diagnostics: the slowest lane's clocks in set-up and cuts, workgroup 0)
					u32 d0c = (u32)(tq1 - tq0), d1c = (u32)(clock64() - tq1);
					for (int o = 32; o > 0; o >>= 1) {
						d0c = max(d0c, (u32)__shfl_xor((int)d0c, o));
						d1c = max(d1c, (u32)__shfl_xor((int)d1c, o));
					}
					if (0 == lane) {
						atomicMax(&sh[8], d0c);
						atomicMax(&sh[9], d1c);
					}
				}
				__syncthreads();
				if (0 == (threadIdx.x | blockIdx.x) && 0 == round && first_pass) {  // (diagnostics)
					ctl->dbg[35] = wall_clock64();
					ctl->dbg[40] = sh[8];
					ctl->dbg[41] = sh[9];
				}
				const u32 nsegs = min(min(sh[1], sh[2]), qcap);
				// ---- every lane walks segments ----
				const unsigned long long tw0 = clock64();
				u32 wmaxc = 0;
				for (u32 si = threadIdx.x; si < nsegs; si += blockDim.x) {
					const SegRec rec = q[si];
					const RayConst c = rc[rec.ray & 0x7FFFFFFFu];
					double tmx = rec.tm[0], tmy = rec.tm[1], tmz = rec.tm[2];
					const double tdx = c.td[0], tdy = c.td[1], tdz = c.td[2];
					const long long idist = __double_as_longlong(c.dist);
					const i32 dlx = c.dl[0], dly = c.dl[1], dlz = c.dl[2];
					const u32 end = rec.end;
					u32 lin = rec.lin;
					bool go = (0 != (rec.ray & 0x80000000u)) ||
					          ((lin != c.glin) && ((__double_as_longlong(tmx) <= idist) | (__double_as_longlong(tmy) <= idist) | (__double_as_longlong(tmz) <= idist)));
					u32 cnt = 0;
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
						const bool more = (__double_as_longlong(tmx) <= idist) | (__double_as_longlong(tmy) <= idist) | (__double_as_longlong(tmz) <= idist);
						go = (lin != end) & more & (cnt < 4096u);
					}
					if (cnt >= 4096u) err |= ERR_RUNAWAY;  // (a segment is ~K steps by construction)
					steps += cnt;
					wmaxc = max(wmaxc, cnt);
				}
				if (0 == round && first_pass && 0 == blockIdx.x) {  // (diagnostics)
					u32 dwc = (u32)(clock64() - tw0);
					for (int o = 32; o > 0; o >>= 1) {
						dwc = max(dwc, (u32)__shfl_xor((int)dwc, o));
						wmaxc = max(wmaxc, (u32)__shfl_xor((int)wmaxc, o));
					}
					if (0 == lane) {
						atomicMax(&sh[10], dwc);
						atomicMax(&sh[11], wmaxc);
						sh[12] = nsegs;
					}
				}
				const int more_rounds = __syncthreads_or((pending || wcur < nw) ? 1 : 0);  // (the queue and the counters are no longer read)
				if (threadIdx.x < 3u) sh[threadIdx.x] = (2u == threadIdx.x) ? 0xFFFFFFFFu : 0u;
				if (!more_rounds) break;
				__syncthreads();
			}
			if (0 == threadIdx.x) sh[5] = 0;
			first_pass = false;
			__syncthreads();  // (the list and the counters are free for the next window)
		}
	}
	for (int o2 = 32; o2 > 0; o2 >>= 1) {
		nhit += __shfl_xor(nhit, o2);
		ncast += __shfl_xor(ncast, o2);
	}
	if (0 == lane && nhit) atomicAdd(&sh[4], nhit);
	if (0 == lane && ncast) atomicAdd(&sh[3], ncast);
	__syncthreads();
	if (0 == (threadIdx.x | blockIdx.x)) {  // (diagnostics)
		ctl->dbg[36] = wall_clock64();
		ctl->dbg[42] = sh[10];
		ctl->dbg[43] = sh[11];
		ctl->dbg[39] = sh[12];
	}
	if (0 == threadIdx.x) {
		// per-workgroup partials, folded by k_fmerge (hundreds of workgroups adding to one word serialise at ~12 ns each)
		steps_part[gridDim.x + blockIdx.x] = sh[3];
		steps_part[2u * gridDim.x + blockIdx.x] = sh[4];
	}
	if (nwg) {
		// the sub-box to its place among the octant's slabs
		const u32 w4 = octWords4(og, oct);
		const uint4* l4 = reinterpret_cast<const uint4*>(lds);
		uint4* out4 = slabs + (size_t)slab_off4[oct] + (size_t)jwg * w4;
		for (u32 j4 = threadIdx.x; j4 < w4; j4 += blockDim.x) out4[j4] = l4[j4];
	}
	if (0 == (threadIdx.x | blockIdx.x)) ctl->dbg[37] = wall_clock64();  // (diagnostics)
	blockStoreSteps(steps, steps_part);
	if (oob) atomicAdd(&ctl->n_oob, oob);
	if (err) atomicOr(&ctl->err, err);
	if (0 == (threadIdx.x | blockIdx.x)) ctl->dbg[38] = wall_clock64();  // (diagnostics)
}

// ------------------------------------------------------------------------------------------------
// F2s: the ray kernel of the fast path for SIMPLE ray casting (freeSpaceSimple, occupancy_map_base.h:1303-1339; the server's
// `simple_ray_casting` switch): n = int(distance / size) fixed steps of dir * size from the ray's end towards the sensor,
// the cell of every point on the way -- three independent chains of repeated additions per ray, no DDA state, at most a
// few hundred steps: one lane per ray, marks into the workgroup's LDS copy of the grid, the slab handed over like
// k_fcast's. Same prologue (the points blockIdx.x, blockIdx.x + gridDim.x, ...; losers of their voxel dropped).
// ------------------------------------------------------------------------------------------------
template <bool DISCRETE>
__global__ __launch_bounds__(1024) void k_fcast_simple(MapGeom g, FastGeo fg, D3 sensor, u32 n, const u32* __restrict__ first, u32* __restrict__ slabs,
                                                      const ScanCtl* ctl_in, ScanCtl* ctl, unsigned long long* __restrict__ steps_part,
                                                      const PointRec* __restrict__ recs, Pipe* solo, ScanDesc solo_desc)
{
	if (solo && 0 == (threadIdx.x | blockIdx.x)) {
		solo->ring[0] = solo_desc;
		solo->slot[0].first = 0;
		solo->slot[0].B = 1;
	}
	extern __shared__ __attribute__((aligned(16))) u32 lds[];
	__shared__ u32 sh_cnt[2];
	const u32 err_in = ctl_in->err;
	const Grid& gr = fg.gr;
	const u32 lds_words = (u32)(gr.bytes >> 2);
	{
		uint4* l4 = reinterpret_cast<uint4*>(lds);
		for (u32 j = threadIdx.x; j < (lds_words >> 2); j += blockDim.x) l4[j] = make_uint4(0, 0, 0, 0);
	}
	if (threadIdx.x < 2u) sh_cnt[threadIdx.x] = 0;
	if (err_in) return;  // the scan does not fit the predicted grid (k_fhits): it will be repeated (uniform exit)
	__syncthreads();
	const u32 rowBits = fg.rowBits, planeBits = fg.planeBits;
	const u32 lim = 1u << g.L;
	const u32 lane = threadIdx.x & 63u;
	unsigned long long steps = 0;
	u32 err = 0, oob = 0, nray = 0, nhit = 0;
	const u32 pts = (n > blockIdx.x) ? (n - blockIdx.x + gridDim.x - 1) / gridDim.x : 0u;
	const double ns = nodeSize(g, 0u);
	const u64 budget = 3ull * (1ull << g.L) + 8;
	for (u32 p = threadIdx.x; p < pts; p += blockDim.x) {
		const u32 i = blockIdx.x + p * gridDim.x;
		const PointRec r = recs[i];
		const bool odd = 0 != (r.flags & 4u);
		bool cast = (r.flags & 1u) && !odd;
		if ((r.flags & 2u) && !odd) {
			const bool winner = first[r.cell] == i;
			nhit += winner ? 1u : 0u;
			if (DISCRETE && !winner) cast = false;  // OMB:358-360: dropped entirely, no ray
		}
		if (!cast) continue;
		++nray;
		// freeSpace (OMB:1229-1259) -> freeSpaceSimple: backwards from the ray's end (the fast path admits only segments inside the
		// map cube: moveLineInside leaves them alone)
		D3 cur = r.end;
		D3 dir = sensor - cur;
		const double dist = norm(dir);
		dir = dir / dist;
		const int num_steps = (int)(dist / ns);
		if (num_steps < 0 || (u64)num_steps > budget) {
			err |= ERR_RUNAWAY;
			continue;
		}
		const D3 stepv = dir * ns;
		for (int k = 0; k <= num_steps; ++k) {
			err |= markBitChecked(gr, lds, rowBits, planeBits, (i32)toKey1(g, cur.x, 0), (i32)toKey1(g, cur.y, 0), (i32)toKey1(g, cur.z, 0), lim, &oob);
			cur = cur + stepv;
		}
		steps += (unsigned long long)num_steps + 1ull;
	}
	for (int o = 32; o > 0; o >>= 1) {
		nray += __shfl_xor(nray, o);
		nhit += __shfl_xor(nhit, o);
	}
	if (0 == lane) {
		if (nray) atomicAdd(&sh_cnt[0], nray);
		if (nhit) atomicAdd(&sh_cnt[1], nhit);
	}
	__syncthreads();
	if (0 == threadIdx.x) {
		steps_part[gridDim.x + blockIdx.x] = sh_cnt[0];
		steps_part[2u * gridDim.x + blockIdx.x] = sh_cnt[1];
	}
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
// Who applies which scan. The tree update of the fast path runs in SLOTS on the map stream -- k_claim, k_fmerge, k_tile,
// k_ftail, enqueued by the call that brought scan f. When the slot gets its turn (the walk before it has finished, scan
// f's scan half has finished), k_claim CLAIMS a run of scans for it: every scan up to f that no walk has taken yet (the
// host enqueues no slot of its own for a scan while two slots are still waiting on the map stream -- the next slot takes
// it along), and beyond f every scan whose scan half has finished meanwhile (same ray grid, no other update of the map in
// between). The slot's three kernels apply the whole run of B scans in order, in ONE walk of the tree -- one merge launch,
// each block record read and written once (k_tile), one pass over the levels above (k_ftail); a slot whose scan has gone
// with an earlier walk does nothing. So the batch forms on the device, out of whatever has queued up behind the map
// stream at that moment: a host that feeds scans slowly gets B = 1 and the latency of one scan, a host that runs ahead
// gets walks as large as the map stream needs to keep up with the scan stream -- and no scan waits for company.
// ------------------------------------------------------------------------------------------------
// end of a scan half (scan stream, one thread): the scan's descriptor becomes visible, then its number
__global__ void k_scan_done(Pipe* p, ScanDesc d)
{
	tsMark(p->ts, d.fseq, 3, wall_clock64());
	p->ring[d.fseq & (UFO_RING - 1u)] = d;
	__hip_atomic_store(&p->scan_done, d.fseq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// A walk whose scans the HOST names (several GPUs: the scans of all ranks, gathered; ufomap_map_insert_batch): slot 0 of a
// Pipe of its own applies scans 0 .. B-1, the descriptors arrive as kernel arguments.
struct DescPack {
	ScanDesc d[UFO_BATCH_MAX];
};
__global__ void k_batch_descs(Pipe* p, DescPack pack, u32 B)
{
	if (threadIdx.x < B) p->ring[threadIdx.x] = pack.d[threadIdx.x];
	if (0 == threadIdx.x) {
		p->slot[0].first = 0;
		p->slot[0].B = B;
	}
}
// What a rank contributes to the exchange of a batch step, in one piece: [control block (1 KiB) | tile bitmap (1 KiB) |
// ray cells | hit voxels] -- 2 KiB + two bit grids, ~200 KB for a 16 cm / 20 m scan (the update list of the same scan: 0.8 MB).
static_assert(sizeof(ScanCtl) <= UFO_XSLOT_CTL && UFO_FAST_MAX_TILES / 8u <= UFO_XSLOT_HDR - UFO_XSLOT_CTL, "exchange slot layout");
__global__ __launch_bounds__(256) void k_pack_slot(uint4* __restrict__ dst, const uint4* __restrict__ ctl, const uint4* __restrict__ tile_bits,
                                                   const uint4* __restrict__ gridM, const uint4* __restrict__ gridH, u32 n4)
{
	const u32 nctl = (u32)((sizeof(ScanCtl) + 15u) / 16u), ntb = UFO_FAST_MAX_TILES / 8u / 16u;
	const u32 total = UFO_XSLOT_HDR / 16u + 2u * n4;
	for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
		uint4 v = make_uint4(0, 0, 0, 0);
		if (j < UFO_XSLOT_CTL / 16u) {
			if (j < nctl) v = ctl[j];
		} else if (j < UFO_XSLOT_HDR / 16u) {
			if (j - UFO_XSLOT_CTL / 16u < ntb) v = tile_bits[j - UFO_XSLOT_CTL / 16u];
		} else if (j < UFO_XSLOT_HDR / 16u + n4) {
			v = gridM[j - UFO_XSLOT_HDR / 16u];
		} else {
			v = gridH[j - UFO_XSLOT_HDR / 16u - n4];
		}
		dst[j] = v;
	}
}
// Head of slot f (map stream, one wave): wait for scan f's scan half, then claim the run of scans this slot's walk applies.
// (The wait is bounded: a tool that serialises kernels across streams -- rocprofv3 --pmc does -- would keep the producer
// from ever running while this wave spins. The host uses events when it sees such a tool, ufomap_hip.hip: useGates; should
// one slip through, the gate gives up after max_ticks and flags the scan, which then leaves the map alone, is repeated,
// and the handle hands over with events from then on.)
__global__ void k_claim(Pipe* p, unsigned long long f, u32 bmax, ScanCtl* ctl, unsigned long long max_ticks, ScanCtl* host_result,
                        unsigned long long done_value)
{
	if (0 != threadIdx.x) return;
	const unsigned long long t0 = wall_clock64();
	tsMark(p->ts, f, 4, t0);
	unsigned long long done;
	Pipe::Slot& sl = p->slot[f & (UFO_RING - 1u)];
	while ((done = __hip_atomic_load(&p->scan_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < f) {
		__builtin_amdgcn_s_sleep(1);
		if (wall_clock64() - t0 > max_ticks) {  // 100 MHz clock
			// The scan half never finished (its descriptor is not in the ring): the slot applies nothing, its scan is reported
			// as flagged -- the host repeats it -- and the walks behind this one stand back.
			atomicOr(&ctl->err, ERR_GATE);
			sl.first = p->claimed + 1ull;
			sl.B = 0;
			p->wstat[f & (UFO_RING - 1u)] = 1u;
			host_result->err = ERR_GATE;
			__threadfence_system();
			__hip_atomic_store(reinterpret_cast<unsigned long long*>(host_result + 1), done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
			return;
		}
	}
	const unsigned long long first = p->claimed + 1ull;  // scans up to f that have no slot of their own come along (the host
	sl.first = first;                                     // sees to it that they share f's ray grid and are fewer than UFO_BATCH_MAX)
	if (first > f) {
		sl.B = 0;
		tsMark(p->ts, f, 5, wall_clock64());
		return;
	}
	u32 B = (u32)(f - first) + 1u;
	if (done >= f) {
		const u32 geo = p->ring[f & (UFO_RING - 1u)].geo;
		while (B < bmax && first + B <= done) {
			const ScanDesc& d = p->ring[(first + B) & (UFO_RING - 1u)];
			if (d.fseq != first + B || d.geo != geo) break;
			++B;
		}
	}
	p->claimed = first + B - 1u;
	sl.B = B;
	tsMark(p->ts, f, 5, wall_clock64());
	tsMark(p->ts, f, 7, B);
}

// ------------------------------------------------------------------------------------------------
// F3: k_merge_slabs (scan_kernels.h) + which depth-3 tiles of the grid hold a marked cell (one bit per tile): the
// tree-update kernels start from that bitmap instead of searching the grid. First kernel of a walk: for every scan the
// walk has claimed.
// ------------------------------------------------------------------------------------------------
// One scan of a walk merged: the slabs ORed into the scan's ray grid, its hit grid from the first-point array, its tile bitmap; the
// launch's last workgroup folds the per-workgroup counts and boxes instead (k_fmerge, k_fmerge_batch)
__device__ __forceinline__ void fmergeScan(const FastGeo& fg, const ScanDesc& d, u32 n4, uint4 (*part)[64], uint8_t (*hb)[64], u32* tb)
{
	const u32 tb_words = (fg.ntiles + 31u) / 32u;
	ScanCtl* ctl = d.ctl;
	// (n_slabs == 0: a ray grid beyond LDS -- the ray kernel has marked the scan's grid in HBM itself, k_cast<2>; what is left
	// to do here is the hit grid and the tile bitmap)
	const bool noslab = 0 == d.n_slabs;
	if (blockIdx.x + 1u == gridDim.x) {
		// The LAST workgroup of the launch does not merge: it folds the scan's per-workgroup results
		foldBoxes(d.boxes, d.nboxes, ctl);
		if (threadIdx.x < 64u) {
			unsigned long long v = 0, r = 0, h = 0;
			for (u32 s = threadIdx.x; s < d.n_slabs; s += 64u) {
				v += d.parts[s];
				r += d.parts[d.n_slabs + s];
				h += d.parts[2u * d.n_slabs + s];
			}
			for (int o = 32; o > 0; o >>= 1) {
				v += __shfl_xor(v, o);
				r += __shfl_xor(r, o);
				h += __shfl_xor(h, o);
			}
			if (noslab) {
				// (a grid beyond LDS: rays cast / voxels hit per 256-point stretch of the cloud, k_fselect)
				r = h = 0;
				for (u32 s = threadIdx.x; s < d.nboxes; s += 64u) {
					const unsigned long long q = d.parts[s];
					r += q & 0xFFFFFFFFull;
					h += q >> 32;
				}
				for (int o = 32; o > 0; o >>= 1) {
					r += __shfl_xor(r, o);
					h += __shfl_xor(h, o);
				}
				v = 0;  // (k_cast<2> has added its steps itself)
			}
			if (0 == threadIdx.x && 0 == ctl->err) {
				if (v) atomicAdd(&ctl->n_steps, v);
				ctl->n_rays = (u32)r;
				ctl->n_hits = (u32)h;
			}
		}
		__syncthreads();  // (foldBoxes' shared arrays are reused for the next scan)
		return;
	}
	const u32 nmerge = gridDim.x - 1u;  // workgroups that merge
	if (ctl->err) return;               // (uniform: the scan was flagged by its scan half; the walk will stand back)
	const uint4* __restrict__ slabs = d.slabs;
	uint4* __restrict__ grid = reinterpret_cast<uint4*>(d.gridM);
	const u32 n_slabs = d.n_slabs;
	const u32 col = threadIdx.x & 63u, sl16 = threadIdx.x >> 6;
	for (u32 j = threadIdx.x; j < tb_words; j += blockDim.x) tb[j] = 0;
	__syncthreads();
	const u32 rowW = fg.rowBits >> 5, ny = 2u * (u32)fg.gr.nb[1];
	for (u32 j0 = blockIdx.x * 64u; j0 < n4; j0 += nmerge * 64u) {
		const u32 j = j0 + col;
		uint4 acc = make_uint4(0, 0, 0, 0);
		if (j < n4 && noslab) {
			if (0 == sl16) acc = grid[j];
		} else if (j < n4) {
			if (d.oct) {
				// (k_fcast4: a word of the grid lies in the sub-box of one octant -- of up to eight on the sensor's row / plane / word --
				// and only that octant's workgroups have a copy of it: sixteen slab lanes share them)
				const OctTab* ot = d.oct;
				const OctGeo og = ot->og;
				const u32* sw = reinterpret_cast<const u32*>(slabs);
				const u32 sxw = og.s[0] >> 5, nzf = 2u * (u32)fg.gr.nb[2];
				u32 accw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
				for (u32 k = 0; k < 4u; ++k) {
					const u32 widx = 4u * j + k;
					const u32 row = widx / rowW, wx = widx - row * rowW;
					const u32 z = row / ny, y = row - z * ny;
					if (z >= nzf) continue;  // (padding behind the last row)
					// which sides of the sensor's word / row / plane the word lies on (the sensor's own belong to both)
					const u32 ax = (wx <= sxw ? 1u : 0u) | (wx >= sxw ? 2u : 0u), ay = (y <= og.s[1] ? 1u : 0u) | (y >= og.s[1] ? 2u : 0u),
					          az = (z <= og.s[2] ? 1u : 0u) | (z >= og.s[2] ? 2u : 0u);
					for (u32 o = 0; o < 8u; ++o) {
						const u32 ox = o & 1u, oy = (o >> 1) & 1u, oz = o >> 2;
						if (!((ax >> ox) & (ay >> oy) & (az >> oz) & 1u)) continue;
						const u32 nw = ot->wg_start[o + 1u] - ot->wg_start[o];
						if (0 == nw) continue;
						const u32 sub = (wx - og.xw0[ox]) + og.nxw[ox] * ((y - og.y0[oy]) + og.ny[oy] * (z - og.z0[oz]));
						const size_t w4x4 = 4u * (size_t)octWords4(og, o);
						const u32* base = sw + 4u * (size_t)ot->slab_off4[o] + sub;
						for (u32 jj = sl16; jj < nw; jj += 16u) accw[k] |= base[(size_t)jj * w4x4];
					}
				}
				acc = make_uint4(accw[0], accw[1], accw[2], accw[3]);
			} else
			// (asking for four slabs' words at a time was measured and lost: 13.3 -> 16.3 us by events, 15.3 -> 21.6 under rocprofv3)
			for (u32 s = sl16; s < n_slabs; s += 16u) {
				const uint4 a = slabs[(size_t)s * n4 + j];
				acc.x |= a.x;
				acc.y |= a.y;
				acc.z |= a.z;
				acc.w |= a.w;
			}
		}
		// the scan's hit grid: one bit per cell that holds a first point (the voxel receives a hit, OMB:295, 358-360), from the
		// dense first-point array -- 128 entries per column, eight per slab lane -- which is left clean for the set's next scan
		u32 hbits = 0;
		if (j < n4 && d.first && !noslab) {  // (noslab: k_fselect has built the hit grid)
			uint4* f4 = reinterpret_cast<uint4*>(d.first + (size_t)128u * j + 8u * sl16);
			const uint4 fa = f4[0], fb = f4[1];
			hbits = (fa.x != 0xFFFFFFFFu ? 1u : 0u) | (fa.y != 0xFFFFFFFFu ? 2u : 0u) | (fa.z != 0xFFFFFFFFu ? 4u : 0u) | (fa.w != 0xFFFFFFFFu ? 8u : 0u) |
			        (fb.x != 0xFFFFFFFFu ? 16u : 0u) | (fb.y != 0xFFFFFFFFu ? 32u : 0u) | (fb.z != 0xFFFFFFFFu ? 64u : 0u) | (fb.w != 0xFFFFFFFFu ? 128u : 0u);
			if (hbits && !d.rgb) {  // (a coloured scan: k_tile reads the first points -- whose colour the voxel gets -- and cleans up)
				f4[0] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
				f4[1] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
			}
		}
		hb[sl16][col] = (uint8_t)hbits;
		part[sl16][col] = acc;
		__syncthreads();
		if (0 == sl16 && j < n4) {
			if (d.first && !noslab) {
				uint4 hv;
				hv.x = (u32)hb[0][col] | ((u32)hb[1][col] << 8) | ((u32)hb[2][col] << 16) | ((u32)hb[3][col] << 24);
				hv.y = (u32)hb[4][col] | ((u32)hb[5][col] << 8) | ((u32)hb[6][col] << 16) | ((u32)hb[7][col] << 24);
				hv.z = (u32)hb[8][col] | ((u32)hb[9][col] << 8) | ((u32)hb[10][col] << 16) | ((u32)hb[11][col] << 24);
				hv.w = (u32)hb[12][col] | ((u32)hb[13][col] << 8) | ((u32)hb[14][col] << 16) | ((u32)hb[15][col] << 24);
				reinterpret_cast<uint4*>(d.gridH)[j] = hv;
			}
			if (!noslab) {
				for (u32 k = 1; k < 16u; ++k) {
					const uint4 a = part[k][col];
					acc.x |= a.x;
					acc.y |= a.y;
					acc.z |= a.z;
					acc.w |= a.w;
				}
				grid[j] = acc;
			}
			const u32 wv[4] = {acc.x, acc.y, acc.z, acc.w};
			for (u32 k = 0; k < 4u; ++k) {
				u32 m = wv[k];
				if (0 == m) continue;
				const u32 widx = 4u * j + k;
				const u32 row = widx / rowW, wx = widx % rowW;
				const u32 ly = row % ny, lz = row / ny;
				if (lz >= 2u * (u32)fg.gr.nb[2]) continue;  // (padding behind the last row)
				const i32 ty = ((fg.gr.base[1] + (i32)ly) >> 3) - fg.tbase[1], tz = ((fg.gr.base[2] + (i32)lz) >> 3) - fg.tbase[2];
				while (m) {
					const u32 bit = (u32)__ffs(m) - 1u;
					const i32 ax = fg.gr.base[0] + (i32)(32u * wx + bit);
					const i32 tx = (ax >> 3) - fg.tbase[0];
					// all cells of this word that fall into the same tile
					const i32 first_in_tile = ((ax >> 3) << 3) - fg.gr.base[0] - (i32)(32u * wx);  // bit index of the tile's first cell (may be < 0)
					const u32 lo = (u32)max(first_in_tile, 0), hi = (u32)min(first_in_tile + 8, 32);
					const u32 span = (hi >= 32u ? 0xFFFFFFFFu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
					m &= ~span;
					const u32 tile = (u32)tx + fg.nt[0] * ((u32)ty + fg.nt[1] * (u32)tz);
					if (tile < fg.ntiles) atomicOr(&tb[tile >> 5], 1u << (tile & 31u));
				}
			}
		}
		__syncthreads();
	}
	for (u32 j = threadIdx.x; j < tb_words; j += blockDim.x)
		if (tb[j]) atomicOr(&d.tile_bits[j], tb[j]);
	__syncthreads();  // (tb is cleared for the next scan)
}
__global__ __launch_bounds__(1024) void k_fmerge(FastGeo fg, const Pipe* __restrict__ p, unsigned long long f, u32 n4)
{
	__shared__ uint4 part[16][64];
	__shared__ uint8_t hb[16][64];
	__shared__ u32 tb[UFO_BIG_MAX_TILES / 32];
	const Pipe::Slot sl = p->slot[f & (UFO_RING - 1u)];
	// (the scans of the walk: one after the other, or -- gridDim.y > 1 -- every gridDim.y-th by this row of workgroups: while the ray
	// kernel holds three CUs in four the rows queue on the same CUs either way, but the LAST walk of a run of scans has the chip to
	// itself, and what a timed region waits for at its end is exactly that walk)
	for (u32 b = blockIdx.y; b < sl.B; b += gridDim.y) fmergeScan(fg, p->ring[(sl.first + b) & (UFO_RING - 1u)], n4, part, hb, tb);
}
// A batch step's own scan (several GPUs, ufomap_map_insert_batch; round 6): merged STRAIGHT INTO the rank's exchange slot -- `own.gridM` /
// `own.gridH` point into it -- and the workgroup that finishes last adds the slot's header (the finished control block and tile bitmap)
// and the descriptors of the step's walk (the scans of all ranks, in the receive buffer: known to the host before the step starts).
// What k_batch_descs + k_fmerge + k_pack_slot on the scan stream and k_batch_descs on the map stream did in four launches.
// Hand-over inside the launch: every wave's stores drained, the workgroup's release, a ticket; the last ticket acquires and reads
// what the others' atomics and stores left with loads that bypass its L1 (the guide's last-arriver recipe).
static_assert(0 == sizeof(ScanCtl) % 8u, "the control block is copied in 8-byte words");
__global__ __launch_bounds__(1024) void k_fmerge_batch(FastGeo fg, Pipe* __restrict__ p, ScanDesc own, u32 n4, unsigned long long* __restrict__ slot_hdr, DescPack walk, u32 B)
{
	__shared__ uint4 part[16][64];
	__shared__ uint8_t hb[16][64];
	__shared__ u32 tb[UFO_BIG_MAX_TILES / 32];
	fmergeScan(fg, own, n4, part, hb, tb);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (0 == threadIdx.x) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		const u32 t = __hip_atomic_fetch_add(&p->merge_arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const u32 last = (t + 1u == gridDim.x) ? 1u : 0u;
		if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		tb[0] = last;
	}
	__syncthreads();
	if (!tb[0]) return;
	const unsigned long long* cw = reinterpret_cast<const unsigned long long*>(own.ctl);
	const unsigned long long* tw = reinterpret_cast<const unsigned long long*>(own.tile_bits);
	for (u32 j = threadIdx.x; j < UFO_XSLOT_HDR / 8u; j += blockDim.x) {
		unsigned long long v = 0ull;
		if (j < UFO_XSLOT_CTL / 8u) {
			if (j < (u32)(sizeof(ScanCtl) / 8u)) v = __hip_atomic_load(cw + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		} else if (j - UFO_XSLOT_CTL / 8u < UFO_FAST_MAX_TILES / 64u) {
			v = __hip_atomic_load(tw + (j - UFO_XSLOT_CTL / 8u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		slot_hdr[j] = v;
	}
	if (threadIdx.x < B) p->ring[threadIdx.x] = walk.d[threadIdx.x];
	if (0 == threadIdx.x) {
		p->slot[0].first = 0;
		p->slot[0].B = B;
		p->merge_arrivals = 0u;  // (for the set's next step)
	}
}

// The node blocks above the tiles, found without searching: the tiles form a regular grid, so do their ancestors --
// level l is the tile grid coarsened by 2^(l-3). One dense grid of cells per level 4 .. L (a few hundred cells in all);
// off[l] = first cell of level l in the concatenation, which is therefore in level order. Filled by the host.
#define UFO_UPPER_MAX 1024u  // cells (hence node blocks) above the tiles that one scan may touch; the host keeps other scans off this path
struct UpperGeo {
	i32 lo[24][3];
	u32 n[24][3];
	u32 off[25];
};
__host__ __device__ inline u32 upperCell(const UpperGeo& ug, u32 l, const i32 c[3])
{
	// (no short-circuit evaluation: with a uniform level the seven words are scalar loads, and a chain of branches would
	// wait for them one at a time -- measured: 650 cycles per call in k_ftail's marking loop)
	const u32 n0 = ug.n[l][0], n1 = ug.n[l][1], n2 = ug.n[l][2], off = ug.off[l];
	const u32 x = (u32)(c[0] - ug.lo[l][0]), y = (u32)(c[1] - ug.lo[l][1]), z = (u32)(c[2] - ug.lo[l][2]);  // (negative: huge)
	const bool inside = (x < n0) & (y < n1) & (z < n2);
	return inside ? off + x + n0 * (y + n1 * z) : 0xFFFFFFFFu;
}
// ------------------------------------------------------------------------------------------------
// Tree update, part 1 (k_tile): one wavefront per active depth-3 tile, ONE WALK FOR A BATCH OF SCANS. Lane l owns the
// level-1 node block whose 6-bit position inside the tile is l (three Morton digits: child index inside the level-2
// block = l & 7, level-2 block = l >> 3) AND the slice of every block above that lies on its path: "its" depth-1 node's
// value in the level-2 block, "its" depth-2 node's value in the level-3 block (replicated over the 8 lanes of a group).
// A level of updateNode is then a reduction over 8 lanes (xor shuffles 1, 2, 4 for level 2; 8, 16, 32 for level 3) --
// every lane ends up with the same summary, nothing is broadcast, nothing goes through memory.
//
// The batch: scans 0 .. B-1 (one GPU: the scans that queued up behind the previous walk; several GPUs: the scans of
// all ranks, in rank order) are applied IN ORDER, exactly as if the reference had integrated them one after the other
// (occupancy_map_base.h:340-417 called B times) -- but the tile's 73 block records are read once, live in registers
// while the B scans go over them (createNode with inheritance, updateOccupancy, updateNode, pruning and re-expansion
// of what an earlier scan of the batch collapsed: all of it on the register copy), and are written once. Per scan the
// wave reads 8 words per lane: its cells' bits in the scan's miss grid (ray cells) and hit grid.
// Reads: those words, per lane one block's values, key and flags (44 B from three arrays: table.h), the parents' slices
// (coalesced 4-byte loads). Writes: each touched block's values and flags once, and only if they changed. Nothing above level 3 is written here: a tile whose level-3 block is new
// looks its inherited value up (read-only: nobody changes the blocks above during this launch) and leaves the rest --
// creating the blocks above, linking, their summaries -- to k_ftail.
//
// Semantics per scan and level are exactly k_apply_leaf + propagateCore (map_kernels.h): hits (clamp) then misses
// (clamp) on the voxels; a block's summary goes to its parent's slot; a parent is re-evaluated only if a child's stored
// summary changed or the child's last update alone changed it; a node collapses only if the last update beneath it
// reached it. At insert depth 0 every touched block has a miss (the end cell of every ray is a miss cell, OMB:1286), and
// misses are applied after all hits in ascending code order: "the last update beneath a node" is always the miss in
// the highest touched child of the LAST scan that touched it, at every level.
// ------------------------------------------------------------------------------------------------
// What a new node block inherits (createChildren, octree.h:1044-1054): the value (and colour) of the nearest node above that has a
// live block -- the root node's own when there is none. The NL lanes of a group (lane index `sub` inside it) look the ancestors up
// side by side: a lone lane walking up the empty path of a fresh map is a dozen DEPENDENT round trips, and a fresh 2 mm frame
// creates 9e5 tiles -- 2.6 of its tree update's 4.5 ms (round 5). Nothing above the caller's level is written during its launch.
template <bool COLOR, int NL>
__device__ inline void inheritLookup(const Table& t, u64 lk, u32 sub, float* vin, u32* cin)
{
	const u32 na = (63u - (u32)__clzll((long long)lk)) / 3u;  // ancestors: lk >> 3, lk >> 6, ..., 1
	u32 best = 0xFFFFFFFFu, bc = 0;
	float bv = 0.f;
	for (u32 a = sub; a < na; a += (u32)NL) {
		const u32 sa = tableFind(t, lk >> (3u * (a + 1u)));
		if (sa != NONE && !(t.flags(sa) & F_DEAD)) {
			const u32 child = (u32)((lk >> (3u * a)) & 7);
			best = a;
			bv = t.occ(sa)[child];
			if (COLOR) bc = t.rgb[8 * (size_t)sa + child];
			break;  // (ascending: the lane's nearest)
		}
	}
	u32 m = best;
	for (int o = 1; o < NL; o <<= 1) m = min(m, (u32)__shfl_xor((int)m, o));
	if (0xFFFFFFFFu == m) {
		*vin = t.root->occ;
		if (COLOR) *cin = t.root->rgb;
	} else {
		const int src = (int)((__lane_id() & ~(u32)(NL - 1)) + (m % (u32)NL));
		*vin = __shfl(bv, src);
		if (COLOR) *cin = (u32)__shfl((int)bc, src);
	}
}
struct TileRec {
	float occ, pre_occ;  // summary of the tile's level-3 block after the batch / just before its last update
	u32 slot;            // table slot of the level-3 block
	u32 bits;            // 0-1 fl, 2-3 pre fl, 4 evaluated (summary handed to the parent), 5 last update reached and changed it,
	                     // 6 the level-3 block is new (to be linked to its parent), 7 it collapsed, 8-10 child index in the parent
	u32 seq;             // walk that wrote the record
	// bookkeeping that must not become 1 400 atomics on one word (each ~12 ns, serialised): summed up by k_ftail.
	// bits 0-10 level-1 blocks updated (<= 64 per scan), 11-24 voxels that received a hit (<= 512 per scan), 25-31 node blocks created (<= 73)
	u32 counts;
	u32 last;            // index (in the batch) of the last scan that touched the tile: the "time" of its last update
	u32 rgb;             // colour maps: colour summary of the level-3 block as last evaluated (updateNode, OMC.cpp:115-140)
};
__device__ inline u32 flagsOf(const MapGeom& g, float v) { return (isFreeV(g, v) ? 1u : 0u) | (isUnknownV(g, v) ? 2u : 0u); }
// reductions over the 8 lanes that differ in the three lane-index bits starting at bit `sh` (0: a level-2 group, 3: across groups)
// Round 6: a cross-lane move by __shfl_xor is a ds_bpermute -- an LDS-unit operation, ~120 clocks of latency each, and a reduction is
// three of them one behind the other: the level loops of k_ftail spent their time there (a level through REGISTERS measured no faster
// than a level through LDS: 1 900 clocks). Partners inside a row of 16 lanes are reached by DPP instead -- a VALU operand modifier, a
// few clocks: xor 1 / xor 2 = quad_perm, "the other quad of my eight" = row_half_mirror (lane i <-> 7 - i: any pairing that joins the
// two quads serves a reduction whose quads are already reduced), xor 8 = row_ror:8. Only the steps across rows (xor 16, xor 32) stay
// shuffles. Every lane of the wave must be active (all callers: uniform control flow).
__device__ __forceinline__ float grpMax(float v, int sh)
{
	if (0 == sh) {
		v = fmaxf(v, dppF<UFO_DPP_X1>(v));
		v = fmaxf(v, dppF<UFO_DPP_X2>(v));
		return fmaxf(v, dppF<UFO_DPP_HM>(v));
	}
	v = fmaxf(v, dppF<UFO_DPP_R8>(v));
	v = fmaxf(v, __shfl_xor(v, 16));
	return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float grpMin(float v, int sh)
{
	if (0 == sh) {
		v = fminf(v, dppF<UFO_DPP_X1>(v));
		v = fminf(v, dppF<UFO_DPP_X2>(v));
		return fminf(v, dppF<UFO_DPP_HM>(v));
	}
	v = fminf(v, dppF<UFO_DPP_R8>(v));
	v = fminf(v, __shfl_xor(v, 16));
	return fminf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ u32 grpOr(u32 v, int sh)
{
	if (0 == sh) {
		v |= dppU<UFO_DPP_X1>(v);
		v |= dppU<UFO_DPP_X2>(v);
		return v | dppU<UFO_DPP_HM>(v);
	}
	v |= dppU<UFO_DPP_R8>(v);
	v |= (u32)__shfl_xor((int)v, 16);
	return v | (u32)__shfl_xor((int)v, 32);
}
__device__ __forceinline__ u32 grpMaxU(u32 v, int sh)
{
	if (0 == sh) {
		v = max(v, dppU<UFO_DPP_X1>(v));
		v = max(v, dppU<UFO_DPP_X2>(v));
		return max(v, dppU<UFO_DPP_HM>(v));
	}
	v = max(v, dppU<UFO_DPP_R8>(v));
	v = max(v, (u32)__shfl_xor((int)v, 16));
	return max(v, (u32)__shfl_xor((int)v, 32));
}
__device__ __forceinline__ u32 grpMinU(u32 v, int sh)
{
	if (0 == sh) {
		v = min(v, dppU<UFO_DPP_X1>(v));
		v = min(v, dppU<UFO_DPP_X2>(v));
		return min(v, dppU<UFO_DPP_HM>(v));
	}
	v = min(v, dppU<UFO_DPP_R8>(v));
	v = min(v, (u32)__shfl_xor((int)v, 16));
	return min(v, (u32)__shfl_xor((int)v, 32));
}
// all eight equal: largest == smallest (log-odds are never NaN; -0 == +0 either way)
__device__ __forceinline__ bool grpAllEq(float v, int sh, u32 lane)
{
	(void)lane;
	return grpMax(v, sh) == grpMin(v, sh);
}
__device__ __forceinline__ bool grpAllEqU(u32 v, int sh, u32 lane)
{
	(void)lane;
	return grpMaxU(v, sh) == grpMinU(v, sh);
}
// Colour summary of a node (updateNode of a colour map, OMC.cpp:115-140 = blockSummary, map_kernels.h): per channel the
// root mean square of the children that have a colour. Sums of at most eight squares of bytes: exact in any order.
__device__ inline void rgbSq(u32 c, double& rr, double& gg, double& bb, u32& cnt)
{
	if (c) {
		const double cr = (double)(c & 0xFFu), cg = (double)((c >> 8) & 0xFFu), cb = (double)((c >> 16) & 0xFFu);
		rr += cr * cr;
		gg += cg * cg;
		bb += cb * cb;
		++cnt;
	}
}
__device__ inline u32 rgbRms(double rr, double gg, double bb, u32 cnt)
{
	if (0 == cnt) return 0u;
	const double num = (double)cnt;
	const u32 R = (u32)(uint8_t)sqrt(rr / num), G = (u32)(uint8_t)sqrt(gg / num), B = (u32)(uint8_t)sqrt(bb / num);
	return R | (G << 8) | (B << 16);
}
// ... over the 8 lanes of a group (lane = child): every lane gets the summary
__device__ inline u32 grpRgb(u32 c, int sh)
{
	double rr = 0, gg = 0, bb = 0;
	u32 cnt = 0;
	rgbSq(c, rr, gg, bb, cnt);
	if (0 == sh) {
		// (sums of at most eight squares of bytes: exact in any order)
		rr += dppD<UFO_DPP_X1>(rr); gg += dppD<UFO_DPP_X1>(gg); bb += dppD<UFO_DPP_X1>(bb); cnt += dppU<UFO_DPP_X1>(cnt);
		rr += dppD<UFO_DPP_X2>(rr); gg += dppD<UFO_DPP_X2>(gg); bb += dppD<UFO_DPP_X2>(bb); cnt += dppU<UFO_DPP_X2>(cnt);
		rr += dppD<UFO_DPP_HM>(rr); gg += dppD<UFO_DPP_HM>(gg); bb += dppD<UFO_DPP_HM>(bb); cnt += dppU<UFO_DPP_HM>(cnt);
		return rgbRms(rr, gg, bb, cnt);
	}
	for (int o = 1; o < 8; o <<= 1) {
		rr += __shfl_xor(rr, o << sh);
		gg += __shfl_xor(gg, o << sh);
		bb += __shfl_xor(bb, o << sh);
		cnt += (u32)__shfl_xor((int)cnt, o << sh);
	}
	return rgbRms(rr, gg, bb, cnt);
}
// level-3 block key of a tile; false if the tile lies outside the key range
__device__ inline bool tileKey(const MapGeom& g, const FastGeo& fg, u32 tile, u64* lk3, u32* tcoord)
{
	const u32 ttx = tile % fg.nt[0], r = tile / fg.nt[0];
	const u32 tty = r % fg.nt[1], ttz = r / fg.nt[1];
	const i32 T[3] = {fg.tbase[0] + (i32)ttx, fg.tbase[1] + (i32)tty, fg.tbase[2] + (i32)ttz};
	const i32 lim = (i32)(1u << (g.L - 3u));
	if (T[0] < 0 || T[1] < 0 || T[2] < 0 || T[0] >= lim || T[1] >= lim || T[2] >= lim) return false;
	*lk3 = (1ULL << (3 * (g.L - 3))) | morton3((u32)T[0], (u32)T[1], (u32)T[2]);
	if (tcoord) {
		tcoord[0] = ttx;
		tcoord[1] = tty;
		tcoord[2] = ttz;
	}
	return true;
}

// COLOR (OccupancyMapColor): every node carries a colour beside its value. A voxel that receives a hit takes the colour
// of its first point, blended with the colour it has (updateNodeColor, OMC.cpp:142-171: with the occupancy BEFORE the
// hit); misses leave colours alone -- so the last update beneath a node (a miss) never changes a colour and the "reached"
// chain is the one of plain maps; what colours add: a summary per node (root mean square per channel), "all children
// equal" includes their colours, and a changed colour summary re-evaluates the parent like a changed value does.
// VOL (the volume path, vol_kernels.h: ray grids of millions of tiles, e.g. a 2 mm RGB-D frame): the wave's tile comes from the
// list of active tiles, the scan's ray cells and hit voxels from TILE-MAJOR brick grids -- per tile eight 64-bit words, one
// per 4x4x4 cells, bit = x | y << 2 | z << 4 inside the brick -- i.e. one 64-byte line per tile and grid; one scan per walk.
// Blocks are created against a reserve (sharded counters): a tile that finds it used up has not written anything yet,
// flags ERR_GROW and stands back -- the host grows the table and runs the tiles that are left (their records do not carry
// the walk's number yet).
// 64-bit words of one XCD's copy of the brick grid: whole 256-byte pieces, so that no cache line holds words of two copies
__host__ __device__ inline size_t volCopyWords(u32 ntiles) { return ((size_t)ntiles * 8u + 31u) & ~(size_t)31u; }
#define UFO_UPCNT_STRIDE 32u  // 32-bit words between two of k_up's pairs of counters (volume path; 64 pairs)
#define UFO_UPCNT_WORDS (64u * UFO_UPCNT_STRIDE)
#define UFO_RESV_STRIDE 32u
struct TileVol {
	u64* Mx;               // ray cells: eight copies, one per XCD (k_vdda); read, ORed and left zeroed here
	u64* Mm;               // ... the tile's merged words, for whoever asks for the scan's ray cells afterwards (may be null)
	u64* H;                // hit voxels; left zeroed
	const u32* list;       // active tiles
	const uint8_t* copies; // ... and the copies each was marked in
	HitHash hh;            // colour maps: the scan's hit hash (k_classify: voxel code -> index of the voxel's FIRST point) ...
	const uint8_t* rgb;    // ... and the cloud's colours, 3 bytes per point (null: a cloud without colours)
	const u32* slots;      // ... and a guess of the slot of each tile's level-3 block (k_vlist: from the record of an earlier walk)
	u32 retry;             // the walk is run again after a table growth: tiles whose records carry its number are done
	u32 count;
	u32* resv;             // 64 counters of tile groups claimed by this walk, UFO_RESV_STRIDE words apart (a cache line each)
	u32 resv_lim;          // ... and what each may reach
};
template <bool COLOR, bool VOL = false>
__global__ __launch_bounds__(256) void k_tile(Table t, MapGeom g, FastGeo fg, const Pipe* __restrict__ p, unsigned long long f, TileRec* __restrict__ recs,
                                              float upd_hit, float upd_miss, u32 scan_id, const u32* __restrict__ prev_stat, ChangeLog cl, TileVol va)
{
	const u32 lane = threadIdx.x & 63u;
	u32 tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	u32 vcopies = 0, vslot = NONE;
	if (VOL) {
		vcopies = tile < va.count ? va.copies[tile] : 0u;
		vslot = tile < va.count ? va.slots[tile] : NONE;
		tile = tile < va.count ? va.list[tile] : 0xFFFFFFFFu;
	}
	if (tile >= fg.ntiles) return;
	if (VOL && va.retry && recs[tile].seq == scan_id) return;  // (a repeat after the table has grown: the tile is done)
	const u32 spec_slot = VOL ? vslot : recs[tile].slot;  // where the tile's level-3 block was when a walk last left a record for this tile (a guess, checked below)
	__builtin_amdgcn_s_setprio(2);  // (the map stream is the pipeline's critical path: ahead of the ray kernel's waves)
	const Pipe::Slot sl = p->slot[f & (UFO_RING - 1u)];
	const u32 B = VOL ? 1u : sl.B;  // (0: the slot's scan went with an earlier walk; the volume path: one scan per walk, known at compile time)
#define UFO_DESC(b) (p->ring[(sl.first + (b)) & (UFO_RING - 1u)])
	// words that can end the wave here are asked for together (a wave's time is its chain of dependent round trips):
	// the walk enqueued just before this one flagged itself and left the map alone (this one stands back too, k_ftail
	// tells the host); a scan of the batch was flagged by its scan half (k_fhits / k_fcast), i.e. before anything touched
	// the map; the tile holds no ray cell of any scan
	u32 errs = prev_stat ? *prev_stat : 0u;
	u32 tmask = 0;  // bit b: the tile holds a ray cell of scan b
	for (u32 b = 0; b < B; ++b) {
		errs |= UFO_DESC(b).ctl->err;
		if (VOL) tmask |= 1u << b;  // (the list holds active tiles only)
		else tmask |= ((UFO_DESC(b).tile_bits[tile >> 5] >> (tile & 31u)) & 1u) << b;
	}
	if (errs) return;  // (uniform)
	if (!tmask) return;
	u64 lk3;
	u32 tt[3];
	if (!tileKey(g, fg, tile, &lk3, tt)) return;
	const u32 c2 = lane >> 3, c1 = lane & 7u;
	const u32 bx = (c1 & 1u) | ((c2 & 1u) << 1), by = ((c1 >> 1) & 1u) | (((c2 >> 1) & 1u) << 1), bz = ((c1 >> 2) & 1u) | (((c2 >> 2) & 1u) << 1);
	const u64 lk2 = (lk3 << 3) | (u64)c2, lk1 = (lk2 << 3) | (u64)c1;
	// The wave's time is its chain of dependent memory round trips, so everything that can be asked for at once is:
	// round 1: the grid words and a SPECULATIVE lookup of every block the tile could touch (64 level-1 keys by the
	// 64 lanes, the 8 level-2 keys and the level-3 key by lanes 0..8); round 2: the records and slices behind those
	// slots; round 3 (rare): creations.
	// ---- round 1 ----
	const i32 ox = (fg.tbase[0] + (i32)tt[0]) * 8 - fg.gr.base[0] + 2 * (i32)bx;
	const i32 oy = (fg.tbase[1] + (i32)tt[1]) * 8 - fg.gr.base[1] + 2 * (i32)by;
	const i32 oz = (fg.tbase[2] + (i32)tt[2]) * 8 - fg.gr.base[2] + 2 * (i32)bz;
	const i32 nx = 2 * fg.gr.nb[0], ny = 2 * fg.gr.nb[1], nz = 2 * fg.gr.nb[2];
	const u32 rowW = fg.rowBits >> 5;
	// the lane's 8 cells in every scan's grids: 8 bits of ray cells + 8 bits of hits per scan, packed (scans 0-7 / 8-15)
	u64 mmA = 0, mmB = 0, hmA = 0, hmB = 0;
	const u32 vbrick = (bx >> 1) | ((by >> 1) << 1) | ((bz >> 1) << 2), vsh = 2u * (bx & 1u) + 8u * (by & 1u) + 32u * (bz & 1u);
	if (VOL) {
		// the lane's 2x2x2 cells inside their brick: bits vsh + cx + 4 cy + 16 cz
		u64 mword = 0;
		{
			const size_t cstride = volCopyWords(fg.ntiles);
			u64 cw[8];
#pragma unroll
			for (int k = 0; k < 8; ++k) cw[k] = ((vcopies >> k) & 1u) ? va.Mx[(size_t)k * cstride + (size_t)tile * 8u + vbrick] : 0ull;  // (uniform: in flight together)
#pragma unroll
			for (int k = 0; k < 8; ++k) mword |= cw[k];
		}
		const u64 mw = mword >> vsh, hw = va.H[(size_t)tile * 8u + vbrick] >> vsh;
		const u32 ml = (u32)mw, hl = (u32)hw;
		mmA = (ml & 3u) | (((ml >> 4) & 3u) << 2) | (((ml >> 16) & 3u) << 4) | (((ml >> 20) & 3u) << 6);
		hmA = (hl & 3u) | (((hl >> 4) & 3u) << 2) | (((hl >> 16) & 3u) << 4) | (((hl >> 20) & 3u) << 6);
		hmA &= mmA;  // (a hit voxel is a ray cell: the end cell of its point's ray)
	} else {
		u32 widx[4];
		bool wok[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const i32 ly = oy + (k & 1), lz = oz + (k >> 1);
			wok[k] = ox >= 0 && ox + 1 < nx && ly >= 0 && ly < ny && lz >= 0 && lz < nz;
			widx[k] = wok[k] ? ((u32)lz * (u32)ny + (u32)ly) * rowW + ((u32)ox >> 5) : 0u;
		}
		const u32 sh = (u32)ox & 31u;
		// four scans' words at a time (32 loads in flight), scans that do not touch the tile are skipped (uniform)
		u32 rem = tmask;
		while (rem) {
			u32 bs[4], nb = 0;
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				bs[q] = rem ? (u32)__ffs(rem) - 1u : bs[0];
				nb += rem ? 1u : 0u;
				rem &= rem - 1u;  // (0 stays 0)
			}
			u32 wm[4][4], wh[4][4];
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const u32* gM = UFO_DESC(bs[q]).gridM;
				const u32* gH = UFO_DESC(bs[q]).gridH;
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					wm[q][k] = wok[k] ? gM[widx[k]] : 0u;
					wh[q][k] = wok[k] ? gH[widx[k]] : 0u;
				}
			}
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				u32 mm = 0, hm = 0;
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					mm |= ((wm[q][k] >> sh) & 3u) << (2 * (k & 1) + 4 * (k >> 1));
					hm |= ((wh[q][k] >> sh) & 3u) << (2 * (k & 1) + 4 * (k >> 1));
				}
				hm &= mm;  // (a hit voxel is a ray cell: the end cell of its point's ray)
				if ((u32)q < nb) {
					const u32 s8 = 8u * (bs[q] & 7u);
					if (bs[q] < 8u) {
						mmA |= (u64)mm << s8;
						hmA |= (u64)hm << s8;
					} else {
						mmB |= (u64)mm << s8;
						hmB |= (u64)hm << s8;
					}
				}
			}
		}
	}
	// Tile-major table (table.h): ONE probe of the tile directory; the 73 slots follow from the group, and whether a block is
	// there is read off the key that arrives with its record -- the records are asked for right after the probe, no round trip
	// for the keys of their own. (Maps of fewer than four levels have no groups: the blocks one by one.)
	const bool grouped = t.L >= 4u;
	u32 s1 = NONE, s2 = NONE, s3 = NONE;
	const bool uactive = 0 != (mmA | mmB);                 // the lane's block is touched by some scan of the batch
	const u32 uact2 = grpOr(uactive ? 1u : 0u, 0);         // ... its level-2 group is
	u32 fl3r = F_DEAD, fl2r = F_DEAD, fl1r = F_DEAD;
	float v2l = 0.f, v1l = 0.f;
	float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	u32 r2l = 0, r1l = 0;  // COLOR: the colours beside v2l, v1l, v[]
	u32 col[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	// The record a walk left for this tile names the slot of its level-3 block -- same tile grid, same table: the group is
	// known without the directory (2.2 dependent probes on average), and the key that arrives with the record says whether the
	// guess was right (another grid, a table exchanged since: the directory after all). One round trip of a wave's four.
	bool spec = false;
	u32 grp = NONE;
	if (grouped) {
		if (spec_slot >= t.capU && spec_slot - t.capU < UFO_GROUP * t.nG && (spec_slot - t.capU) % UFO_GROUP == 72u) {
			grp = (spec_slot - t.capU) / UFO_GROUP;
			spec = true;
		} else grp = groupFind(t, lk3);
	}
	for (;;) {  // (at most twice; uniform)
		s1 = s2 = s3 = NONE;
		if (grouped) {
			if (grp != NONE) {
				s1 = groupSlot(t, grp, lane);
				s2 = groupSlot(t, grp, 64u + c2);
				s3 = groupSlot(t, grp, 72u);
			}
		} else {
			s1 = tableFind(t, lk1);
			u32 sx = NONE;  // lanes 0..7: the level-2 block of group `lane`; lane 8: the level-3 block
			if (lane < 8u) sx = tableFind(t, (lk3 << 3) | (u64)lane);
			else if (8u == lane) sx = tableFind(t, lk3);
			s3 = __shfl(sx, 8);
			s2 = __shfl(sx, (int)c2);
		}
		// ---- round 2: the records as they are stored (a block found DEAD was collapsed: the node is a leaf, octree.h:1060-1066;
		// a block that is not there counts as DEAD) ----
		fl3r = fl2r = fl1r = F_DEAD;
		v2l = v1l = 0.f;
		r2l = r1l = 0;
		u64 k3 = lk3, k2 = lk2, k1 = lk1;  // the keys the slots hold (grouped tables: a slot of the group may be empty)
		if (s3 != NONE) {
			if (grouped) k3 = t.key(s3);
			fl3r = t.flags(s3);
			v2l = t.occ(s3)[c2];
			if (COLOR) r2l = t.rgb[8 * (size_t)s3 + c2];
		}
		if (s2 != NONE && uact2) {
			if (grouped) k2 = t.key(s2);
			fl2r = t.flags(s2);
			v1l = t.occ(s2)[c1];
			if (COLOR) r1l = t.rgb[8 * (size_t)s2 + c1];
		}
		if (s1 != NONE && uactive) {
			if (grouped) k1 = t.key(s1);
			fl1r = t.flags(s1);
			const float4* po = reinterpret_cast<const float4*>(t.occ(s1));
			const float4 ra = po[0], rb = po[1];
			v[0] = ra.x; v[1] = ra.y; v[2] = ra.z; v[3] = ra.w;
			v[4] = rb.x; v[5] = rb.y; v[6] = rb.z; v[7] = rb.w;
			if (COLOR) {
				const uint4* pc = reinterpret_cast<const uint4*>(t.rgb + 8 * (size_t)s1);
				const uint4 ca = pc[0], cb = pc[1];
				col[0] = ca.x; col[1] = ca.y; col[2] = ca.z; col[3] = ca.w;
				col[4] = cb.x; col[5] = cb.y; col[6] = cb.z; col[7] = cb.w;
			}
		}
		if (spec && k3 != lk3) {  // the guess was wrong (uniform): the directory
			spec = false;
			grp = groupFind(t, lk3);
			continue;
		}
		if (grouped) {
			// a slot whose key is not the block's: the block is not there (what was read from the slot is not its record)
			if (s3 != NONE && k3 != lk3) {
				s3 = NONE;
				fl3r = F_DEAD;
				v2l = 0.f;
				r2l = 0;
			}
			if (s2 != NONE && uact2 && k2 != lk2) {
				s2 = NONE;
				fl2r = F_DEAD;
				v1l = 0.f;
				r1l = 0;
			}
			if (s1 != NONE && uactive && k1 != lk1) {
				s1 = NONE;
				fl1r = F_DEAD;
			}
		}
		if (!(s1 != NONE && uactive)) {
#pragma unroll
			for (int c = 0; c < 8; ++c) {
				v[c] = 0.f;
				col[c] = 0;
			}
		}
		break;
	}
	// ---- round 3: createNode for what is missing from the table (octree.h:997-1016); a level-3 block that is not live
	// inherits the value of the nearest node above that has one (the blocks above are not written during this launch) ----
	u32 n_created = 0;
	bool new1 = false, new2 = false;  // the lane's level-1 / level-2 block got its key in this walk: the parent link is written
	const u32 fl1_read = fl1r;        // (the flags word as it is stored: written back only when it changes)
	float v3s = 0.f;
	u32 r3s = 0;  // COLOR: the colour that goes with v3s
	{
		const bool need3 = 0 != (fl3r & F_DEAD);
		const bool mk3 = s3 == NONE, mk2 = uact2 && s2 == NONE && 0 == c1, mk1 = uactive && s1 == NONE;
		if (__ballot(need3 || mk2 || mk1)) {
			const u32 max_probe = (t.mask >> 1) + 1;
			bool dummy;
			if (VOL) {
				// nothing has been written yet: a tile that needs a group of its own (its level-3 block is not there: table.h) takes it
				// out of the walk's reserve, or stands back
				const u32 need = mk3 ? 1u : 0u;
				u32 over = 0;
				if (need && 0 == lane) over = (atomicAdd(&va.resv[((tile * 0x9E3779B1u) >> 26) * UFO_RESV_STRIDE], need) + need > va.resv_lim) ? 1u : 0u;
				if (__shfl((int)over, 0)) {
					if (0 == lane) atomicOr(&UFO_DESC(B - 1u).ctl->err, ERR_GROW);
					return;
				}
			}
			// VOL on a grouped table: the listed tile belongs to THIS wave for the whole launch (the list holds every tile once,
			// nothing else updates the map meanwhile), so once its group is there -- lane 0's claim in the directory -- the keys of
			// the blocks it creates are plain stores into the group's slots: no probe and no compare-and-swap per block (a fresh
			// 2 mm frame creates 4.9e7 blocks; device-scope atomics run at ~1e7 per ms on this part)
			const bool own = VOL && grouped;
			u32 grp_own = NONE;
			if (own) {
				if (0 == lane) grp_own = (s3 != NONE) ? (s3 - t.capU) / UFO_GROUP : groupEnsure(t, lk3);
				grp_own = (u32)__shfl((int)grp_own, 0);
			}
			if (need3) inheritLookup<COLOR, 64>(t, lk3, lane, &v3s, &r3s);  // (uniform: every lane gets the values)
			if (need3 && 0 == lane) {
				if (mk3) {
					if (own) {
						if (grp_own != NONE) {
							s3 = groupSlot(t, grp_own, 72u);
							t.key(s3) = lk3;
							t.stamp(s3) = scan_id;
							++n_created;
						}
					} else s3 = tableEnsure(t, lk3, scan_id, max_probe, &dummy, &n_created);
				}
			}
			if (own) {
				if (grp_own != NONE) {
					if (mk2) {
						s2 = groupSlot(t, grp_own, 64u + c2);
						t.key(s2) = lk2;
						t.stamp(s2) = scan_id;
						++n_created;
					}
					if (mk1) {
						s1 = groupSlot(t, grp_own, lane);
						t.key(s1) = lk1;
						t.stamp(s1) = scan_id;
						++n_created;
					}
				}
			} else {
				if (mk2) s2 = tableEnsure(t, lk2, scan_id, max_probe, &dummy, &n_created);
				if (mk1) s1 = tableEnsure(t, lk1, scan_id, max_probe, &dummy, &n_created);
			}
			new1 = mk1;
			new2 = mk2;
			s3 = __shfl(s3, 0);
			s2 = __shfl(s2, (int)(lane & ~7u));
			if (__ballot((s3 == NONE) || (uact2 && s2 == NONE) || (uactive && s1 == NONE))) {
				if (0 == lane) atomicOr(&UFO_DESC(B - 1u).ctl->err, ERR_TABLE_FULL);  // (the host sizes the table for the worst case before launching)
				return;
			}
		}
	}
	// ---- the scans of the batch, in order, on the register copy of the tile's records ----
	bool w2 = false, w3 = false;      // the lane's slot of the level-2 / level-3 record has to be written
	bool any_eval3 = false, created3 = false, reach_last = false;
	float m3c = 0.f, pm_last = 0.f;   // summary of the level-3 block as last evaluated / before its last update
	u32 fl3c = 0, pfl_last = 0, last_b = 0, rgb3c = 0;
	u32 n_touched = 0, nhit = 0;
	bool wr1 = false;  // the values (or colours) of the lane's level-1 block changed: the record is written
	for (u32 b = 0; b < B; ++b) {
		if (8u == b) {
			mmA = mmB;
			hmA = hmB;
		}
		const u32 mmask = (u32)mmA & 255u, hmask = (u32)hmA & 255u;
		mmA >>= 8;
		hmA >>= 8;
		if (!((tmask >> b) & 1u)) continue;  // (uniform)
		const bool active = 0 != mmask;
		const u32 act2 = grpOr(active ? 1u : 0u, 0);  // the lane's level-2 group is touched by this scan
		// which blocks are there for this scan
		const bool cr3 = 0 != (fl3r & F_DEAD);
		const bool cr2 = act2 && (cr3 || 0 != (fl2r & F_DEAD));
		const bool cr1 = active && (cr2 || 0 != (fl1r & F_DEAD));
		created3 = created3 || cr3;
		// ---- stored state of the lane's slices; new blocks inherit (octree.h:1044-1054) ----
		// depth-2 node c2: slot c2 of the level-3 block
		float v2s;
		u32 f2s, in2s;  // stored value, flags, "has a live block" of the lane's depth-2 node
		u32 r2s = 0, r1s = 0;  // COLOR: stored colours of the lane's depth-2 / depth-1 node
		if (cr3) {
			v2s = v3s;
			f2s = flagsOf(g, v3s);
			in2s = 0;
			r2s = r3s;
		} else {
			r2s = r2l;
			v2s = v2l;
			f2s = ((fl3r >> c2) & 1u) | (((fl3r >> (8 + c2)) & 1u) << 1);
			in2s = (fl3r >> (16 + c2)) & 1u;
		}
		// depth-1 node c1 of group c2: slot c1 of the level-2 block (if the group is touched)
		float v1s = v2s;
		u32 f1s = flagsOf(g, v2s), in1s = 0;
		r1s = r2s;
		if (act2 && !cr2) {
			r1s = r1l;
			v1s = v1l;
			f1s = ((fl2r >> c1) & 1u) | (((fl2r >> (8 + c1)) & 1u) << 1);
			in1s = (fl2r >> (16 + c1)) & 1u;
		}
		const u32 f3s = 0;  // (only the defaults of a summary that is not handed over)
		// ---- level 1: updateOccupancy on the voxels (hits, then misses), the block's own updateNode ----
		float cur1 = v1s;  // current value / flags of the lane's depth-1 node
		u32 curf1 = f1s, in1 = in1s;
		bool want1 = false, reach1 = false;
		float pre1 = v1s;
		u32 pref1 = f1s;
		u32 curc1 = r1s;  // COLOR: current colour of the lane's depth-1 node
		if (active) {
			if (cr1) {
#pragma unroll
				for (int c = 0; c < 8; ++c) {
					v[c] = v1s;
					if (COLOR) col[c] = r1s;
				}
			}
			if (COLOR) {
				// updateValue(code, update, color) (OMC.h:275-277): the colour first, with the occupancy the voxel has before the
				// hit; the colour is the one of the voxel's FIRST point (OMC.h:195-233), whose index the dense first-point array
				// holds -- read here (k_fmerge left the array alone for a coloured scan) and cleaned for the set's next scan
				const uint8_t* rgbp = VOL ? va.rgb : UFO_DESC(b).rgb;
				if (VOL && rgbp && hmask) {
					// (the volume path: no dense first-point array over a grid of 4e9 cells -- the voxel's first point is looked up in
					// the scan's hit hash, a few hundred thousand entries for the hits of a frame)
					const u64 vbase = (lk1 ^ (1ULL << (3 * (g.L - 1)))) << 3;
#pragma unroll
					for (int c = 0; c < 8; ++c) {
						if (!((hmask >> c) & 1u)) continue;
						const u32 hs = hitHashFind(va.hh, vbase | (u64)c);
						if (NONE == hs) continue;  // (cannot happen: the hit grid was derived from the hash's entries)
						const u32 pt = va.hh.minidx[hs];
						const u32 u = (u32)rgbp[3 * (size_t)pt] | ((u32)rgbp[3 * (size_t)pt + 1] << 8) | ((u32)rgbp[3 * (size_t)pt + 2] << 16);
						col[c] = blendColor(g, col[c], u, v[c]);
					}
				} else if (rgbp && hmask) {
					u32* fp = UFO_DESC(b).first;
#pragma unroll
					for (int c = 0; c < 8; ++c) {
						if (!((hmask >> c) & 1u)) continue;
						const u32 cell = (u32)(ox + (c & 1)) + (u32)(oy + ((c >> 1) & 1)) * fg.rowBits + (u32)(oz + (c >> 2)) * fg.planeBits;
						const u32 pt = fp[cell];
						fp[cell] = 0xFFFFFFFFu;
						if (0xFFFFFFFFu == pt) continue;  // (cannot happen: the hit grid was derived from this array)
						const u32 u = (u32)rgbp[3 * (size_t)pt] | ((u32)rgbp[3 * (size_t)pt + 1] << 8) | ((u32)rgbp[3 * (size_t)pt + 2] << 16);
						col[c] = blendColor(g, col[c], u, v[c]);
					}
				}
			}
			const int c_last = 31 - __clz((int)mmask);  // ascending code order: the highest touched voxel is updated last
			float v_old_last = 0.f;
			u32 chg = 0;  // voxels whose value a hit or a miss changed: updateOccupancy returned true (OMB:1069-1072)
#pragma unroll
			for (int c = 0; c < 8; ++c) {
				float x = v[c];
				if ((hmask >> c) & 1u) {
					const float y = clampAdd(x, upd_hit, g.cmin, g.cmax);
					chg |= (y != x) ? (1u << c) : 0u;
					x = y;
				}
				if ((mmask >> c) & 1u) {
					if (c == c_last) v_old_last = x;
					const float y = clampAdd(x, upd_miss, g.cmin, g.cmax);
					chg |= (y != x) ? (1u << c) : 0u;
					x = y;
				}
				v[c] = x;
			}
			// (a saturated voxel's update changes nothing: a record none of whose values changed is not stored)
			wr1 = wr1 || cr1 || 0 != chg || (COLOR && 0 != hmask);
			// change detection (OMB:783, 1069-1072): the voxels' codes go to the log, as k_apply_leaf's do
			if (cl.buf) logChanges(t, cl, (lk1 ^ (1ULL << (3 * (g.L - 1)))) << 3, 0u, chg);
			// updateNode of a depth-1 node (OMB:1195-1224): max, flags from the 8 voxels, collapsible if all equal
			float m = v[0], pm = (0 == c_last) ? v_old_last : v[0];
			u32 fl = 0, pfl = 0;
			bool eq = true;
#pragma unroll
			for (int c = 0; c < 8; ++c) {
				const float pv = (c == c_last) ? v_old_last : v[c];
				m = fmaxf(m, v[c]);
				pm = fmaxf(pm, pv);
				fl |= flagsOf(g, v[c]);
				pfl |= flagsOf(g, pv);
				eq = eq && (v[c] == v[0]);
			}
			if (COLOR) {
				double rr = 0, gg = 0, bb = 0;
				u32 cnt = 0;
#pragma unroll
				for (int c = 0; c < 8; ++c) {
					rgbSq(col[c], rr, gg, bb, cnt);
					eq = eq && (col[c] == col[0]);
				}
				curc1 = rgbRms(rr, gg, bb, cnt);
			}
			reach1 = !(pm == m && pfl == fl);  // level 1 is always reached (OMB:1128 starts at depth 1)
			pre1 = pm;
			pref1 = pfl;
			const bool dead1 = eq;             // collapsed: the node is a leaf again (octree.h:1060-1066)
			in1 = dead1 ? 0u : 1u;
			want1 = (m != v1s) || (fl != f1s) || reach1 || (COLOR && curc1 != r1s);
			cur1 = m;
			curf1 = fl;
			// the record's flags (low bits as k_init_new leaves them for a new block)
			u32 fw = cr1 ? ((isFreeV(g, v1s) ? F_CFREE : 0u) | (isUnknownV(g, v1s) ? F_CUNK : 0u)) : (fl1r & ~(F_DEAD | F_DIRTY));
			if (dead1) fw |= F_DEAD;
			fl1r = fw;
			nhit += (u32)__popc(hmask);
		}
		// ---- level 2 (reductions over the 8 lanes of a group; every lane of the group computes the same) ----
		const u32 top1 = grpMaxU(active ? (c1 + 1u) : 0u, 0);     // 1 + the highest touched child of the group (0: none)
		const bool eval2 = 0 != grpOr(want1 ? 1u : 0u, 0);
		float cur2 = v2s;
		u32 curf2 = f2s, in2 = in2s;
		bool want2 = false, reach2 = false, dead2 = false;
		float pre2 = v2s;
		u32 pref2 = f2s;
		u32 curc2 = r2s;
		{
			const float m = grpMax(cur1, 0);
			const u32 fl = grpOr(curf1, 0);
			const bool eq = grpAllEq(cur1, 0, lane) && (!COLOR || grpAllEqU(curc1, 0, lane));
			const u32 rgb2 = COLOR ? grpRgb(curc1, 0) : 0u;
			const u32 inner_any = grpOr(in1, 0);
			const bool is_top = active && (c1 + 1u == top1);
			const bool reached = 0 != grpOr((is_top && reach1) ? 1u : 0u, 0);  // the last update beneath reached the child and changed it
			const float pm = grpMax(is_top ? pre1 : cur1, 0);
			const u32 pfl = grpOr(is_top ? pref1 : curf1, 0);
			if (eval2) {
				dead2 = reached && eq && 0 == inner_any;
				reach2 = reached && !(pm == m && pfl == fl);
				want2 = (m != v2s) || (fl != f2s) || reach2 || (COLOR && rgb2 != r2s);
				pre2 = pm;
				pref2 = pfl;
				cur2 = m;
				curf2 = fl;
				curc2 = rgb2;
			}
			if (act2) in2 = dead2 ? 0u : 1u;
		}
		if (act2) {
			// the level-2 record: the lane's child slot; flags of the whole block
			if (active || cr2) {
				v1l = cur1;
				r1l = curc1;
				w2 = true;
			}
			const u32 fbits = grpOr(((curf1 & 1u) << c1) | (((curf1 >> 1) & 1u) << (8 + c1)) | (in1 << (16 + c1)), 0);
			fl2r = fbits | (dead2 ? F_DEAD : 0u);
		}
		// ---- level 3 (reductions across the 8 groups; every lane computes the same) ----
		const u32 top2 = grpMaxU(act2 ? (c2 + 1u) : 0u, 3);
		const bool eval3 = 0 != grpOr(want2 ? 1u : 0u, 3);
		bool reach3 = false, dead3 = false;
		float m3 = v3s, pm3 = v3s;
		u32 fl3n = f3s, pfl3 = f3s, rgb3n = r3s;
		{
			const float m = grpMax(cur2, 3);
			const u32 fl = grpOr(curf2, 3);
			const bool eq = grpAllEq(cur2, 3, lane) && (!COLOR || grpAllEqU(curc2, 3, lane));
			const u32 rgb3 = COLOR ? grpRgb(curc2, 3) : 0u;
			const u32 inner_any = grpOr(in2, 3);
			const bool is_top = act2 && (c2 + 1u == top2);
			const bool reached = 0 != grpOr((is_top && reach2) ? 1u : 0u, 3);
			const float pm = grpMax(is_top ? pre2 : cur2, 3);
			const u32 pfl = grpOr(is_top ? pref2 : curf2, 3);
			if (eval3) {
				dead3 = reached && eq && 0 == inner_any;
				reach3 = reached && !(pm == m && pfl == fl);
				m3 = m;
				fl3n = fl;
				pm3 = pm;
				pfl3 = pfl;
				rgb3n = rgb3;
			}
		}
		{
			// the level-3 record: the group's slot (all eight when the block is new); flags of the whole block
			if (act2 || cr3) {
				v2l = cur2;
				r2l = curc2;
				w3 = true;
			}
			const u32 fbits = grpOr((0 == c1) ? (((curf2 & 1u) << c2) | (((curf2 >> 1) & 1u) << (8 + c2)) | (in2 << (16 + c2))) : 0u, 3);
			fl3r = fbits | (dead3 ? F_DEAD : 0u);
		}
		// what k_ftail needs: the summary as last evaluated, and how the tile's LAST update (this scan's, so far) went
		if (eval3) {
			any_eval3 = true;
			m3c = m3;
			fl3c = fl3n;
			rgb3c = rgb3n;
		}
		reach_last = reach3;
		pm_last = pm3;
		pfl_last = pfl3;
		last_b = b;
		if (dead3) {  // a later scan of the batch re-expands the node from its own value (all children were equal to it)
			v3s = m3;
			r3s = rgb3n;
		}
		n_touched += (u32)__popcll(__ballot(active));
	}
	// ---- every touched record back to the table, once ----
	if (uactive) {
		if (wr1) {
			float4* po = reinterpret_cast<float4*>(t.occ(s1));
			po[0] = make_float4(v[0], v[1], v[2], v[3]);
			po[1] = make_float4(v[4], v[5], v[6], v[7]);
		}
		if (fl1r != fl1_read || new1) t.flags(s1) = fl1r;
		if (new1) t.parent(s1) = s2;  // (a block's parent never changes: linked when the block gets its key)
		if (COLOR && wr1) {
			uint4* pc = reinterpret_cast<uint4*>(t.rgb + 8 * (size_t)s1);
			pc[0] = make_uint4(col[0], col[1], col[2], col[3]);
			pc[1] = make_uint4(col[4], col[5], col[6], col[7]);
		}
	}
	if (uact2) {
		if (w2) {
			t.occ(s2)[c1] = v1l;
			if (COLOR) t.rgb[8 * (size_t)s2 + c1] = r1l;
		}
		if (0 == c1) {
			t.flags(s2) = fl2r;
			if (new2) t.parent(s2) = s3;
		}
	}
	if (0 == c1 && w3) {
		t.occ(s3)[c2] = v2l;
		if (COLOR) t.rgb[8 * (size_t)s3 + c2] = r2l;
	}
	for (int o = 32; o > 0; o >>= 1) {
		n_created += __shfl_xor(n_created, o);
		nhit += __shfl_xor(nhit, o);
	}
	if (VOL && 0 == vsh) {
		// (the brick's first lane: the scan's words are consumed -- the copies and H are clean for the next scan)
		const size_t cstride = volCopyWords(fg.ntiles);
		u64 mword = 0;
#pragma unroll
		for (int k = 0; k < 8; ++k)
			if ((vcopies >> k) & 1u) {
				u64* q = &va.Mx[(size_t)k * cstride + (size_t)tile * 8u + vbrick];
				const u64 v = *q;
				if (v) *q = 0ull;
				mword |= v;
			}
		if (va.Mm) va.Mm[(size_t)tile * 8u + vbrick] = mword;
		u64* qh = &va.H[(size_t)tile * 8u + vbrick];
		if (*qh) *qh = 0ull;
	}
	if (0 == lane) {
		t.flags(s3) = fl3r;  // (the parent link of a new block: k_ftail)
		TileRec r;
		r.occ = m3c;
		r.pre_occ = pm_last;
		r.slot = s3;
		r.bits = (fl3c & 3u) | ((pfl_last & 3u) << 2) | (any_eval3 ? 16u : 0u) | (reach_last ? 32u : 0u) | (created3 ? 64u : 0u) | ((fl3r & F_DEAD) ? 128u : 0u);
		r.bits |= (u32)(lk3 & 7) << 8;
		r.seq = scan_id;
		r.counts = n_touched | (nhit << 11) | (n_created << 25);
		r.last = last_b;
		r.rgb = rgb3c;
		recs[tile] = r;
	}
}
// ------------------------------------------------------------------------------------------------
// Tree update, part 2 (k_ftail): everything above the tiles, by ONE workgroup with the blocks in LDS.
//   1. Which blocks: the tiles form a regular grid, so do their ancestors -- level l is the tile grid coarsened by
//      2^(l-3). Every active tile marks its ancestors in a bitmap over those dense grids (a few hundred cells in all, in
//      LDS; a tile stops at the first ancestor somebody else marked); a prefix popcount turns the bitmap into the node
//      list, in cell order = level order (no hashing, no sorting).
//   2. createNode (octree.h:997-1016) for every node: find or create its block, load it, and let a new block inherit
//      the value of the nearest node above that had one (createChildren, octree.h:1044-1054).
//   3. The tiles' hand-over records: summary into the level-4 slot (writeToParent), the new tiles are linked.
//   4. updateParents (occupancy_map_base.h:1126-1133) level 4 .. root, eight lanes per block (lane = child; the block's
//      summary is three xor shuffles): a block is re-evaluated only if a child asked for it, collapses only if the last
//      update beneath it (the highest touched child, see k_tile) reached it, and leaves what its parent needs to know
//      about that last update in its own out_* entry; a level that re-evaluates nothing ends the walk (nothing above can
//      change). One barrier per level; once a level has at most 8 blocks, wave 0 alone carries on without barriers -- LDS
//      operations of one wave execute in order.
//   5. Every block back to the table once; the root summary to MapRoot; bookkeeping for the host.
// The kernel's duration is its longest chain of dependent instructions (a lone workgroup hides nothing), so the code
// below counts instructions, not bytes.
// ------------------------------------------------------------------------------------------------
// one level of the dense grids above the tiles, computed from the tile grid (the same numbers makeUpperGeo tabulates on
// the host: no table look-ups in the kernel -- scalar loads of the kernel arguments cost ~200 cycles each there)
struct UpperLevel {
	i32 lo[3];
	u32 n[3];
};
__device__ inline UpperLevel upperLevel(const FastGeo& fg, u32 l)
{
	UpperLevel u;
	const u32 sh = l - fg.tl;
	for (int a = 0; a < 3; ++a) {
		u.lo[a] = fg.tbase[a] >> sh;
		u.n[a] = (u32)(((fg.tbase[a] + (i32)fg.nt[a] - 1) >> sh) - u.lo[a] + 1);
	}
	return u;
}
__device__ inline u32 upperCellAt(const UpperLevel& u, u32 off, const i32 c[3])
{
	const u32 x = (u32)(c[0] - u.lo[0]), y = (u32)(c[1] - u.lo[1]), z = (u32)(c[2] - u.lo[2]);  // (negative: huge)
	const bool inside = (x < u.n[0]) & (y < u.n[1]) & (z < u.n[2]);
	return inside ? off + x + u.n[0] * (y + u.n[1] * z) : 0xFFFFFFFFu;
}
// ------------------------------------------------------------------------------------------------
// Tree update, part 1b (k_up; ray grids beyond LDS only): LEVEL 4 IN PARALLEL. k_ftail holds every block above the tiles
// in the LDS of one workgroup, at most UFO_UPPER_MAX of them -- a grid of tens of thousands of tiles has more level-4
// blocks than that. Here eight lanes take one level-4 block (lane = child = one tile's hand-over record): createNode
// with inheritance, the children's summaries into its slots, updateNode -- one step of k_ftail's level loop, the same
// rules -- and the block leaves a hand-over record of its own, in the tiles' format: k_ftail then starts one level
// higher, with the level-4 blocks as its "tiles" (FastGeo::tl = 4).
// ------------------------------------------------------------------------------------------------
template <bool COLOR>
__global__ __launch_bounds__(256) void k_up(Table t, MapGeom g, FastGeo fg, const Pipe* __restrict__ p, unsigned long long f, const TileRec* __restrict__ recs,
                                            TileRec* __restrict__ recs_up, u32* __restrict__ up_bits, u32 scan_id, const u32* __restrict__ prev_stat,
                                            u32* __restrict__ up_cnt = nullptr)
{
	const Pipe::Slot sl = p->slot[f & (UFO_RING - 1u)];
	const u32 B = sl.B;
	if (0 == B) return;
	u32 errs = prev_stat ? *prev_stat : 0u;
	for (u32 b = 0; b < B; ++b) errs |= UFO_DESC(b).ctl->err;  // (UFO_DESC: k_tile's)
	if (errs) return;  // (uniform: the walk stands back, k_tile has left the map alone)
	ScanCtl* const ctl = UFO_DESC(B - 1u).ctl;
	const u32 L = g.L;
	const u32 lane = threadIdx.x & 63u, sub = lane & 7u;
	// (fg.tl = 3: the children are k_tile's tiles, the blocks level 4. The volume path runs this kernel level after level,
	// fg = the grid of the level below, until k_ftail's LDS holds what is left above)
	const u32 lv = fg.tl + 1u;
	const UpperLevel u4 = upperLevel(fg, lv);
	const u32 n4 = u4.n[0] * u4.n[1] * u4.n[2];
	const u32 cell = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
	// the lane's child: tile (2 * cell + d) of the tile grid
	i32 ac[3] = {0, 0, 0};
	TileRec r;
	r.bits = 0;
	bool have = false;
	if (cell < n4) {
		const u32 x = cell % u4.n[0], rr = cell / u4.n[0];
		ac[0] = u4.lo[0] + (i32)x;
		ac[1] = u4.lo[1] + (i32)(rr % u4.n[1]);
		ac[2] = u4.lo[2] + (i32)(rr / u4.n[1]);
		const i32 tx = 2 * ac[0] + (i32)(sub & 1u) - fg.tbase[0], ty = 2 * ac[1] + (i32)((sub >> 1) & 1u) - fg.tbase[1],
		          tz = 2 * ac[2] + (i32)(sub >> 2) - fg.tbase[2];
		if (tx >= 0 && ty >= 0 && tz >= 0 && (u32)tx < fg.nt[0] && (u32)ty < fg.nt[1] && (u32)tz < fg.nt[2]) {
			r = recs[(u32)tx + fg.nt[0] * ((u32)ty + fg.nt[1] * (u32)tz)];
			have = r.seq == scan_id;  // (written by this walk's k_tile: the tile holds a ray cell of one of its scans)
		}
	}
	const bool act = 0 != grpOr(have ? 1u : 0u, 0);  // (whole groups of 8 lanes)
	// ---- createNode (octree.h:997-1016): lane 0 of the group finds or creates the block; a new (or revived) block inherits
	// the value of the nearest node above that has a live block (nothing above level 4 is written during this launch) ----
	const u64 lk = (1ULL << (3 * (L - lv))) | morton3((u32)ac[0], (u32)ac[1], (u32)ac[2]);
	u32 s = NONE, n_created = 0, fw = 0;
	bool cr = false;
	float vin = 0.f;
	u32 cin = 0;
	if (act && 0 == sub) {
		s = tableEnsure(t, lk, scan_id, (t.mask >> 1) + 1, &cr, &n_created);
		if (s == NONE) atomicOr(&ctl->err, ERR_TABLE_FULL);
		else if (!cr) fw = t.flags(s) & ~F_DIRTY;
	}
	const int l0 = (int)(lane & ~7u);
	s = (u32)__shfl((int)s, l0);
	cr = 0 != __shfl(cr ? 1 : 0, l0);
	if (cr) {  // (whole groups of 8 lanes: the ancestors looked up side by side)
		inheritLookup<COLOR, 8>(t, lk, sub, &vin, &cin);
		// leaf children carry the flags of a leaf with this value (OMB:1181-1189)
		fw = (isFreeV(g, vin) ? F_CFREE : 0u) | (isUnknownV(g, vin) ? F_CUNK : 0u);
	}
	fw = (u32)__shfl((int)fw, l0);
	const bool ok = act && s != NONE;
	float v = vin;
	u32 c = cin;
	if (ok && !cr) {
		v = t.occ(s)[sub];
		if (COLOR) c = t.rgb[8 * (size_t)s + sub];
	}
	// ---- the tiles' hand-over records (what k_ftail does for level 4 when it starts at the tiles): summary into the
	// child's slot, link, "is inner" bit, who carries the last update ----
	u32 setb = 0, clrb = 0, key = 0, lub = 0;
	float luo = 0.f;
	bool dirty = false;
	u32 touched = 0, created = n_created;
	if (ok && have) {
		const u32 bits = r.bits;
		touched = r.counts & 2047u;
		created += r.counts >> 25;
		if ((bits & 64u) && r.slot != NONE) t.parent(r.slot) = s;  // a new level-3 block: its parent link
		if (bits & 128u) clrb |= 1u << (16 + sub);
		else if (bits & 64u) setb |= 1u << (16 + sub);
		key = ((r.last + 1u) << 4) | (sub + 1u);
		if (bits & 16u) {
			const u32 old_fl = ((fw >> sub) & 1u) | (((fw >> (8 + sub)) & 1u) << 1), fl = bits & 3u;
			const bool changed = v != r.occ || old_fl != fl || (COLOR && c != r.rgb);
			v = r.occ;
			if (COLOR) c = r.rgb;
			const u32 sm = ((fl & 1u) << sub) | (((fl >> 1) & 1u) << (8 + sub));
			setb |= sm;
			clrb |= ((1u << sub) | (1u << (8 + sub))) & ~sm;
			dirty = changed || 0 != (bits & 32u);
		}
		lub = (((bits & 16u) && (bits & 32u)) ? 4u : 0u) | ((bits >> 2) & 3u);
		luo = r.pre_occ;
	}
	fw = (fw & ~grpOr(clrb, 0)) | grpOr(setb, 0);
	const u32 topkey = grpMaxU(key, 0);
	const u32 tc = (topkey & 15u) - 1u;  // the child that carries the last update beneath the block
	const bool evaluated = 0 != grpOr(dirty ? 1u : 0u, 0);
	const u32 lubt = (u32)__shfl((int)lub, l0 + (int)(tc & 7u));
	const float luot = __shfl(luo, l0 + (int)(tc & 7u));
	const bool reached = evaluated && 0 != (lubt & 4u);
	// ---- updateNode (OMB:1195-1224), as k_ftail's step: the summary, the summary just before the last update beneath ----
	const float m = grpMax(v, 0);
	const bool eq = grpAllEq(v, 0, lane) && (!COLOR || grpAllEqU(c, 0, lane));
	const u32 rgb = COLOR ? grpRgb(c, 0) : 0u;
	const float pm = grpMax((reached && sub == tc) ? luot : v, 0);
	u32 fl = 0, pfl = 0;
	bool dead = false, reach_out = false;
	if (evaluated) {
		fl = ((fw & F_CFREE) ? 1u : 0u) | ((fw & F_CUNK) ? 2u : 0u);
		pfl = fl;
		if (reached) {
			const u32 fsub = (fw & ~((1u << tc) | (1u << (8 + tc)))) | ((lubt & 1u) << tc) | (((lubt >> 1) & 1u) << (8 + tc));
			pfl = ((fsub & F_CFREE) ? 1u : 0u) | ((fsub & F_CUNK) ? 2u : 0u);
		}
		dead = reached && eq && 0 == (fw & F_INNER);
		reach_out = reached && !(pm == m && pfl == fl);
	}
	// ---- the block back to the table, its own hand-over record ----
	if (ok) {
		t.occ(s)[sub] = v;
		if (COLOR) t.rgb[8 * (size_t)s + sub] = c;
		if (0 == sub) {
			t.flags(s) = fw | (dead ? F_DEAD : 0u);
			TileRec o;
			o.occ = m;
			o.pre_occ = reached ? pm : m;
			o.slot = s;
			o.bits = (fl & 3u) | ((pfl & 3u) << 2) | (evaluated ? 16u : 0u) | (reach_out ? 32u : 0u) | (cr ? 64u : 0u) | (dead ? 128u : 0u) | ((u32)(lk & 7) << 8);
			o.seq = scan_id;
			o.counts = 0;  // (the bookkeeping does not travel in the record: counters, below)
			o.last = (topkey >> 4) - 1u;
			o.rgb = rgb;
			recs_up[cell] = o;
			if (up_bits) atomicOr(&up_bits[cell >> 5], 1u << (cell & 31u));
		}
	}
	for (int o = 32; o > 0; o >>= 1) {
		touched += __shfl_xor(touched, o);
		created += __shfl_xor(created, o);
	}
	if (0 == lane) {
		if (up_cnt) {
			// (the volume path: tens of thousands of waves per launch have something to report, and one word takes an atomic every
			// ~12 ns however many are in flight -- 64 pairs of counters, 128 bytes apart, folded into the control block by k_ftail)
			u32* q = up_cnt + UFO_UPCNT_STRIDE * (blockIdx.x & 63u);
			if (touched) atomicAdd(q, touched);
			if (created) atomicAdd(q + 1, created);
		} else {
			if (touched) atomicAdd(&ctl->n_entries[0], touched);
			if (created) {
				atomicAdd(&t.root->used, created);
				atomicAdd(&ctl->ph[0].n_new, created);
			}
		}
	}
}

#define UFO_FTAIL_THREADS 1024
static_assert(UFO_FTAIL_THREADS == UFO_UPPER_MAX, "k_ftail: one thread per cell of the dense grids above the tiles");
template <bool COLOR>
__global__ __launch_bounds__(UFO_FTAIL_THREADS) void k_ftail(Table t, MapGeom g, FastGeo fg, Pipe* __restrict__ p, unsigned long long f,
                                                             const TileRec* __restrict__ recs, u32 scan_id, const u32* __restrict__ prev_stat,
                                                             const ScanCtl* ctl_init, u32* __restrict__ up_bits, u32 nwords3, u32* __restrict__ up_cnt = nullptr,
                                                             u32* __restrict__ up_guess = nullptr, u32 report_dbg = 1u)
{
	// (fg.tl = 3: the tiles are k_tile's, their activity bitmap the union of the scans' tile bitmaps. fg.tl = 4, ray grids
	// beyond LDS: the "tiles" are the level-4 blocks k_up has evaluated, fg describes THEIR grid, up_bits is their activity
	// bitmap; nwords3 = words of a scan's tile bitmap, which this kernel leaves clean either way)
	const u32 tl = fg.tl;
	// the host waits for the word behind a scan's pinned result block, not for an event: an event record is one more packet
	// the map stream's command processor has to get through between two scans (~5 us, scripts/micro/stream_wait.hip)
	const Pipe::Slot sl = p->slot[f & (UFO_RING - 1u)];
	const u32 B = sl.B;
	if (0 == B) return;  // the slot's scan went with an earlier walk (which reports for it)
	ScanCtl* const ctl = UFO_DESC(B - 1u).ctl;  // what the walk as a whole reports (blocks touched / created, table fill, clocks) goes with its last scan
	// This lone workgroup shares its CU with waves of the next scan's ray kernel and of k_tile; its time is its chain of
	// dependent instructions, so its waves take issue priority over theirs (measured: 35 -> 27 us when overlapped).
	__builtin_amdgcn_s_setprio(3);
	__shared__ u32 tbits[UFO_FAST_MAX_TILES / 32], ubits[UFO_UPPER_MAX / 32], uprefix[UFO_UPPER_MAX / 32 + 1];
	__shared__ u64 nk[UFO_UPPER_MAX];
	__shared__ unsigned long long top64[UFO_UPPER_MAX];  // (1 + last scan of the batch that touched the subtree) << 40 | (1 + child index of the highest child it touched) << 32 | who that is (tile or node)
	__shared__ u32 nslot[UFO_UPPER_MAX], npar[UFO_UPPER_MAX], nflags[UFO_UPPER_MAX], out_bits[UFO_UPPER_MAX], chld[UFO_UPPER_MAX];
	__shared__ float nocc[UFO_UPPER_MAX][8], out_pre[UFO_UPPER_MAX];
	__shared__ u32 nrgb[COLOR ? UFO_UPPER_MAX : 1u][8];  // colour maps: the children's colours beside nocc (see k_tile)
	__shared__ uint8_t dirty[UFO_UPPER_MAX], ncreated[UFO_UPPER_MAX];
	__shared__ u32 lstart[26], lvl_dirty[26], created_total, sh_used, sh_ng, sh_nu;
	const u32 nwords = (fg.ntiles + 31u) / 32u;
	{
		// (one round of loads: the error words and the scans' tile bitmaps, whose union is the walk's)
		const u32 perr = prev_stat ? *prev_stat : 0u;
		u32 cerr = 0;
		for (u32 b = 0; b < B; ++b) cerr |= UFO_DESC(b).ctl->err;
		for (u32 j = threadIdx.x; j < nwords; j += blockDim.x) {
			u32 w = 0;
			if (up_bits) w = up_bits[j];
			else
				for (u32 b = 0; b < B; ++b) w |= UFO_DESC(b).tile_bits[j];
			tbits[j] = w;
		}
		if (perr | cerr) {
			// the walk enqueued just before this one flagged itself and left the map alone, or the scan half of one of this
			// walk's scans flagged it (ERR_SPEC / a bound): the whole walk stood back (k_tile), the map is as it was. Every
			// scan of the walk is reported flagged -- with its own error, the others with ERR_PREV -- and the host repeats
			// them in order; the walk enqueued behind this one finds their status words set and stands back as well.
			if (threadIdx.x < B) {
				const u32 b = threadIdx.x;
				const ScanDesc& d = UFO_DESC(b);
				const u32 own = d.ctl->err;
				// (ERR_TABLE_FULL from this walk's own k_tile: the map IS inconsistent, and every scan of the walk says so)
				const u32 e = ((perr || 0 == own) ? (atomicOr(&d.ctl->err, ERR_PREV) | ERR_PREV) : own) | (cerr & ERR_TABLE_FULL);
				p->wstat[(sl.first + b) & (UFO_RING - 1u)] = 1u;
				d.host_result->err = e;
				__threadfence_system();
				__hip_atomic_store(reinterpret_cast<unsigned long long*>(d.host_result + 1), d.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
			}
			return;
		}
	}
	if (0 == threadIdx.x) ctl->dbg[10] = wall_clock64();  // (diagnostics: ufomap_map_debug)
	const u32 L = g.L;
	const u32 lane = threadIdx.x & 63u;
	if (threadIdx.x < UFO_UPPER_MAX / 32) ubits[threadIdx.x] = 0;
	if (threadIdx.x < 26u) lvl_dirty[threadIdx.x] = 0;
	if (0 == threadIdx.x) created_total = 0;
	__syncthreads();
	if (0 == threadIdx.x) ctl->dbg[11] = wall_clock64();  // (diagnostics: ufomap_map_debug)
	// ---- 1. the active cells become the node list ----
	const UpperLevel u4 = upperLevel(fg, tl + 1u);
	const u32 n4all = u4.n[0] * u4.n[1] * u4.n[2];  // cells of level 4 = first cell of level 5
	constexpr u32 MAXT = UFO_FAST_MAX_TILES / UFO_FTAIL_THREADS;  // tiles per thread: tile = k * blockDim + thread
	u32 cell4[MAXT];  // the level-4 parent's cell of the thread's tiles (NONE: tile not active)
#pragma unroll
	for (u32 k = 0; k < MAXT; ++k) {
		const u32 tile = k * blockDim.x + threadIdx.x;
		cell4[k] = NONE;
		if (tile >= fg.ntiles || !((tbits[tile >> 5] >> (tile & 31u)) & 1u)) continue;
		const u32 ttx = tile % fg.nt[0], rr = tile / fg.nt[0];
		const i32 c[3] = {fg.tbase[0] + (i32)ttx, fg.tbase[1] + (i32)(rr % fg.nt[1]), fg.tbase[2] + (i32)(rr / fg.nt[1])};
		const i32 lim = (i32)(1u << (L - tl));
		if (c[0] < 0 || c[1] < 0 || c[2] < 0 || c[0] >= lim || c[1] >= lim || c[2] >= lim) continue;  // (outside the key range: k_tile skipped it, too)
		const i32 pc[3] = {c[0] >> 1, c[1] >> 1, c[2] >> 1};
		const u32 cell = upperCellAt(u4, 0u, pc);
		if (cell >= UFO_UPPER_MAX) continue;
		cell4[k] = cell;
	}
	// first cell of every level (off[l]; off[L+1] = number of cells), by every thread -- a uniform loop of scalar arithmetic; the level
	// whose range holds this thread's cell is the last one that starts at or below it
	u32 my_l = tl + 1u, my_off = 0, ncells = 0;
	{
		u32 off = 0;
		for (u32 k = tl + 1u; k <= L; ++k) {
			const UpperLevel uk = upperLevel(fg, k);
			if (threadIdx.x >= off) {
				my_l = k;
				my_off = off;
			}
			off += uk.n[0] * uk.n[1] * uk.n[2];
		}
		ncells = min(off, UFO_UPPER_MAX);
	}
	// Round 6: no walk "up from every tile" any more (a chain of LDS atomics per level, on sixteen copies of the bitmap so that the waves
	// would not queue on its words: 4.5 us). A cell's activity is read off the level below directly: a level-4 cell looks at its eight
	// tiles' bits, one thread each; then every cell of the levels above -- all levels at once -- looks at ITS BOX of the level-4 bitmap,
	// row by row (a cell of level l covers 2^(l-4) level-4 cells per axis; the grids are small: the box is a few rows of a few bits).
	// One atomic per 32 lanes (the lanes of a word, by ballot), three barriers.
	{
		const u32 cell = threadIdx.x;
		bool act = false;
		if (cell < min(n4all, UFO_UPPER_MAX)) {
			const u32 x = cell % u4.n[0], r = cell / u4.n[0];
			const i32 X = u4.lo[0] + (i32)x, Y = u4.lo[1] + (i32)(r % u4.n[1]), Z = u4.lo[2] + (i32)(r / u4.n[1]);
			const i32 lim = (i32)(1u << (L - tl));
#pragma unroll
			for (int d = 0; d < 8; ++d) {
				const i32 cx = 2 * X + (d & 1), cy = 2 * Y + ((d >> 1) & 1), cz = 2 * Z + (d >> 2);
				const i32 tx = cx - fg.tbase[0], ty = cy - fg.tbase[1], tz = cz - fg.tbase[2];
				const bool in = tx >= 0 && ty >= 0 && tz >= 0 && (u32)tx < fg.nt[0] && (u32)ty < fg.nt[1] && (u32)tz < fg.nt[2] && cx >= 0 && cy >= 0 && cz >= 0 &&
				                cx < lim && cy < lim && cz < lim;
				const u32 tile = in ? (u32)tx + fg.nt[0] * ((u32)ty + fg.nt[1] * (u32)tz) : 0u;
				act = act || (in && ((tbits[tile >> 5] >> (tile & 31u)) & 1u));
			}
		}
		const u64 bm = __ballot(act);
		if (0 == lane && (u32)bm) atomicOr(&ubits[threadIdx.x >> 5], (u32)bm);
		if (32u == lane && (u32)(bm >> 32)) atomicOr(&ubits[threadIdx.x >> 5], (u32)(bm >> 32));
	}
	__syncthreads();
	if (0 == threadIdx.x) ctl->dbg[21] = wall_clock64();
	{
		const u32 cell = threadIdx.x;
		bool act = false;
		if (cell >= n4all && cell < ncells) {
			const u32 lv = my_l;
			const UpperLevel ul = upperLevel(fg, lv);
			const u32 c = cell - my_off;
			const u32 x = c % ul.n[0], r = c / ul.n[0];
			const i32 X = ul.lo[0] + (i32)x, Y = ul.lo[1] + (i32)(r % ul.n[1]), Z = ul.lo[2] + (i32)(r / ul.n[1]);
			const u32 sft = min(lv - (tl + 1u), 24u);
			// the cell's box of level-4 cells, clipped to the level-4 grid
			const long long bx0 = (long long)X << sft, by0 = (long long)Y << sft, bz0 = (long long)Z << sft, ext = (1ll << sft) - 1;
			const i32 x0 = (i32)max(bx0, (long long)u4.lo[0]), x1 = (i32)min(bx0 + ext, (long long)u4.lo[0] + (long long)u4.n[0] - 1);
			const i32 y0 = (i32)max(by0, (long long)u4.lo[1]), y1 = (i32)min(by0 + ext, (long long)u4.lo[1] + (long long)u4.n[1] - 1);
			const i32 z0 = (i32)max(bz0, (long long)u4.lo[2]), z1 = (i32)min(bz0 + ext, (long long)u4.lo[2] + (long long)u4.n[2] - 1);
			if (x0 <= x1) {
				const u32 len = (u32)(x1 - x0 + 1);
				for (i32 z = z0; z <= z1 && !act; ++z)
					for (i32 y = y0; y <= y1 && !act; ++y) {
						u32 start = (u32)(x0 - u4.lo[0]) + u4.n[0] * ((u32)(y - u4.lo[1]) + u4.n[1] * (u32)(z - u4.lo[2]));
						u32 left = len;
						while (left && start < UFO_UPPER_MAX) {  // (the row's bits, a word at a time)
							const u32 b0 = start & 31u, take = min(left, 32u - b0);
							const u32 mask = (take >= 32u ? 0xFFFFFFFFu : ((1u << take) - 1u)) << b0;
							if (ubits[start >> 5] & mask) {
								act = true;
								break;
							}
							start += take;
							left -= take;
						}
					}
			}
		}
		__syncthreads();  // (the level-4 words have been read: the words of the levels above may share them)
		const u64 bm = __ballot(act);
		if (0 == lane && (u32)bm) atomicOr(&ubits[threadIdx.x >> 5], (u32)bm);
		if (32u == lane && (u32)(bm >> 32)) atomicOr(&ubits[threadIdx.x >> 5], (u32)(bm >> 32));
	}
	__syncthreads();
	if (0 == threadIdx.x) ctl->dbg[22] = wall_clock64();
	if (threadIdx.x < 64u) {
		// prefix popcount over the bitmap's 32 words
		u32 word = 0;
		if (lane < UFO_UPPER_MAX / 32) word = ubits[lane];
		const u32 c = (u32)__popc(word);
		u32 incl = c;
		for (int o = 1; o < 64; o <<= 1) {
			const u32 v = __shfl_up(incl, o);
			if ((int)lane >= o) incl += v;
		}
		if (lane <= UFO_UPPER_MAX / 32) uprefix[lane] = incl - c;
	}
	__syncthreads();
	if (0 == threadIdx.x) ctl->dbg[12] = wall_clock64();  // (diagnostics: ufomap_map_debug)
	auto idOf = [&](u32 cell) -> u32 {  // node of a dense cell (NONE: not active)
		if (cell >= UFO_UPPER_MAX) return NONE;
		const u32 w = ubits[cell >> 5], b = cell & 31u;
		if (!((w >> b) & 1u)) return NONE;
		return uprefix[cell >> 5] + (u32)__popc(w & ((1u << b) - 1u));
	};
	// nodes of level l are [lstart[l], lstart[l+1]): the cells below off[l] that are active
	{
		u32 off = 0;
		for (u32 k = tl + 1u; k <= L; ++k) {
			const UpperLevel uk = upperLevel(fg, k);
			if (threadIdx.x == k) {
				const u32 c0 = min(off, UFO_UPPER_MAX);
				lstart[k] = uprefix[c0 >> 5] + ((c0 & 31u) ? (u32)__popc(ubits[c0 >> 5] & ((1u << (c0 & 31u)) - 1u)) : 0u);
			}
			off += uk.n[0] * uk.n[1] * uk.n[2];
		}
		if (threadIdx.x == L + 1u) lstart[L + 1u] = uprefix[ncells >> 5] + ((ncells & 31u) ? (u32)__popc(ubits[ncells >> 5] & ((1u << (ncells & 31u)) - 1u)) : 0u);
	}
	const u32 max_probe = (t.mask >> 1) + 1;
	u32 n_created = 0;
	// ---- 2. one thread per active cell: the node's key, parent, block (found or created, loaded) ----
	{
		const u32 cell = threadIdx.x;  // (UFO_UPPER_MAX == blockDim.x)
		const u32 id = cell < ncells ? idOf(cell) : NONE;
		if (id != NONE) {
			const u32 l = my_l;
			const UpperLevel ul = upperLevel(fg, l);  // (per-thread level: vector arithmetic)
			const u32 c = cell - my_off;
			const u32 x = c % ul.n[0], r = c / ul.n[0];
			const i32 ac[3] = {ul.lo[0] + (i32)x, ul.lo[1] + (i32)(r % ul.n[1]), ul.lo[2] + (i32)(r / ul.n[1])};
			const u64 lk = (1ULL << (3 * (L - l))) | morton3((u32)ac[0], (u32)ac[1], (u32)ac[2]);
			nk[id] = lk;
			u32 par = NONE;
			if (l < L) {
				const i32 pc[3] = {ac[0] >> 1, ac[1] >> 1, ac[2] >> 1};
				par = idOf(upperCellAt(upperLevel(fg, l + 1u), my_off + ul.n[0] * ul.n[1] * ul.n[2], pc));
			}
			npar[id] = par;
			if (par != NONE) chld[par] = id;  // (a child of the block: THE child where a block has one -- the runs of the level loop)
			top64[id] = 0;
			out_bits[id] = 0;
			out_pre[id] = 0.f;
			dirty[id] = 0;
			// Where the cell's block was when a walk last looked (a guess: the same ray grid, the same table -- the steady state): its
			// key, values and flags are asked for together, and the key that arrives says whether the guess was right; the hashed
			// find-or-create (two to three dependent round trips more) is for the cells whose guess fails, a block that has to be
			// created, or one that was collapsed (DEAD: revived like a new one).
			const u32 gs = up_guess ? up_guess[cell] : NONE;
			bool cr = false, hit = false;
			u32 s = NONE, fl = 0;
			if (gs < t.capU) {
				const u64 kg = t.key(gs);
				const float4* po = reinterpret_cast<const float4*>(t.occ(gs));
				const float4 a = po[0], b = po[1];
				uint4 ca = make_uint4(0, 0, 0, 0), cb = ca;
				if (COLOR) {
					const uint4* pc = reinterpret_cast<const uint4*>(t.rgb + 8 * (size_t)gs);
					ca = pc[0];
					cb = pc[1];
				}
				const u32 fg_ = t.flags(gs);
				if (kg == lk && !(fg_ & F_DEAD)) {
					hit = true;
					s = gs;
					float4* lo4 = reinterpret_cast<float4*>(nocc[id]);
					lo4[0] = a;
					lo4[1] = b;
					if (COLOR) {
						uint4* lc4 = reinterpret_cast<uint4*>(nrgb[id]);
						lc4[0] = ca;
						lc4[1] = cb;
					}
					fl = fg_ & ~F_DIRTY;
				}
			}
			if (!hit) {
				s = tableEnsure(t, lk, scan_id, max_probe, &cr, &n_created);
				if (s == NONE) {
					atomicOr(&ctl->err, ERR_TABLE_FULL);
				} else if (!cr) {
					const float4* po = reinterpret_cast<const float4*>(t.occ(s));
					const float4 a = po[0], b = po[1];
					float4* lo4 = reinterpret_cast<float4*>(nocc[id]);
					lo4[0] = a;
					lo4[1] = b;
					if (COLOR) {
						const uint4* pc = reinterpret_cast<const uint4*>(t.rgb + 8 * (size_t)s);
						const uint4 ca = pc[0], cb = pc[1];
						uint4* lc4 = reinterpret_cast<uint4*>(nrgb[id]);
						lc4[0] = ca;
						lc4[1] = cb;
					}
					fl = t.flags(s) & ~F_DIRTY;
				}
				if (up_guess && s != NONE) up_guess[cell] = s;
			}
			nslot[id] = s;
			ncreated[id] = cr ? 1 : 0;
			nflags[id] = fl;
		}
	}
	if (n_created) atomicAdd(&created_total, n_created);
	__syncthreads();
	if (0 == threadIdx.x) ctl->dbg[13] = wall_clock64();  // (diagnostics: ufomap_map_debug)
	const u32 U = lstart[L + 1];
	// new blocks inherit the value of the nearest node above that had a block (walk up the list through the new ones)
	for (u32 i = threadIdx.x; i < U; i += blockDim.x) {
		if (!ncreated[i] || nslot[i] == NONE) continue;
		u32 cur = i;
		float v;
		u32 vc = 0;
		for (;;) {
			const u32 p = npar[cur];
			if (p == NONE) {
				v = t.root->occ;  // the root block itself is new: the root's value
				if (COLOR) vc = t.root->rgb;
				break;
			}
			if (!ncreated[p]) {
				v = nocc[p][(u32)(nk[cur] & 7)];  // (slots of existing blocks have not been written yet)
				if (COLOR) vc = nrgb[p][(u32)(nk[cur] & 7)];
				break;
			}
			cur = p;
		}
		for (int c = 0; c < 8; ++c) nocc[i][c] = v;
		if (COLOR)
			for (int c = 0; c < 8; ++c) nrgb[i][c] = vc;
		// leaf children carry the flags of a leaf with this value (OMB:1181-1189)
		atomicOr(&nflags[i], (isFreeV(g, v) ? F_CFREE : 0u) | (isUnknownV(g, v) ? F_CUNK : 0u));
		if (npar[i] != NONE) atomicOr(&nflags[npar[i]], 1u << (16 + (u32)(nk[i] & 7)));
	}
	__syncthreads();
	if (0 == threadIdx.x) ctl->dbg[14] = wall_clock64();  // (diagnostics: ufomap_map_debug)
	// ---- 3. the tiles hand their level-3 summaries to their level-4 blocks (writeToParent); new tiles are linked ----
	u32 my_touched = 0, my_created = 0;
	{
		TileRec r[MAXT];
#pragma unroll
		for (u32 k = 0; k < MAXT; ++k)
			if (cell4[k] != NONE) r[k] = recs[k * blockDim.x + threadIdx.x];  // (requested together)
#pragma unroll
		for (u32 k = 0; k < MAXT; ++k) {
			if (cell4[k] == NONE || r[k].seq != scan_id) continue;
			const u32 tile = k * blockDim.x + threadIdx.x;
			const u32 n4 = idOf(cell4[k]);
			if (n4 == NONE) continue;  // (cannot happen: marked above)
			my_touched += r[k].counts & 2047u;
			my_created += r[k].counts >> 25;
			const u32 bits = r[k].bits, ci = (bits >> 8) & 7u;
			if (bits & 64u) t.parent(r[k].slot) = nslot[n4];  // a new level-3 block: its parent link
			if (bits & 128u) atomicAnd(&nflags[n4], ~(1u << (16 + ci)));
			else if (bits & 64u) atomicOr(&nflags[n4], 1u << (16 + ci));
			atomicMax(&top64[n4], ((unsigned long long)(r[k].last + 1u) << 40) | ((unsigned long long)(ci + 1u) << 32) | tile);
			if (bits & 16u) {
				const u32 f = nflags[n4];
				const u32 old_fl = ((f >> ci) & 1u) | (((f >> (8 + ci)) & 1u) << 1), fl = bits & 3u;
				const bool changed = nocc[n4][ci] != r[k].occ || old_fl != fl || (COLOR && nrgb[n4][ci] != r[k].rgb);
				nocc[n4][ci] = r[k].occ;
				if (COLOR) nrgb[n4][ci] = r[k].rgb;
				if (old_fl != fl) {
					const u32 setm = ((fl & 1u) << ci) | (((fl >> 1) & 1u) << (8 + ci));
					const u32 clrm = ((1u << ci) | (1u << (8 + ci))) & ~setm;
					if (setm) atomicOr(&nflags[n4], setm);
					if (clrm) atomicAnd(&nflags[n4], ~clrm);
				}
				if (changed || (bits & 32u)) dirty[n4] = 1;
			}
		}
	}
	{
		// the tiles' bookkeeping: one atomic per wave on words only this workgroup touches
		for (int o = 32; o > 0; o >>= 1) {
			my_touched += __shfl_xor(my_touched, o);
			my_created += __shfl_xor(my_created, o);
		}
		if (0 == lane) {
			if (my_touched) atomicAdd(&ctl->n_entries[0], my_touched);
			if (my_created) atomicAdd(&created_total, my_created);
		}
	}
	if (up_cnt && threadIdx.x < 64u) {
		// what the volume path's k_up launches counted (64 pairs of counters): read, left at zero for the next walk
		u32 a = atomicExch(&up_cnt[UFO_UPCNT_STRIDE * lane], 0u), b = atomicExch(&up_cnt[UFO_UPCNT_STRIDE * lane + 1u], 0u);
		for (int o = 32; o > 0; o >>= 1) {
			a += __shfl_xor(a, o);
			b += __shfl_xor(b, o);
		}
		if (0 == lane) {
			if (a) atomicAdd(&ctl->n_entries[0], a);
			if (b) atomicAdd(&created_total, b);
		}
	}
	__syncthreads();
	if (0 == threadIdx.x) ctl->dbg[15] = wall_clock64();  // (diagnostics: ufomap_map_debug)
	// The table's fill after this walk -- nothing is created from here on -- is asked for NOW, beside the level loop, not behind it
	// (an atomic's round trip and two rows of sharded counters: 2 us of the 7.7 that used to sit behind the kernel's last stamp).
	if (0 == threadIdx.x) {
		u32 used = __hip_atomic_load(&t.root->used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (created_total) {
			used = atomicAdd(&t.root->used, created_total) + created_total;
			atomicAdd(&ctl->ph[0].n_new, created_total);
		}
		sh_used = used;
	}
	if (threadIdx.x >= 64u && threadIdx.x < 128u) {
		u32 ng, nu;
		tableCounts(t, threadIdx.x - 64u, &ng, &nu);
		if (64u == threadIdx.x) {
			sh_ng = ng;
			sh_nu = nu;
		}
	}
	// who carries the last update beneath a level-4 block: its highest touched tile (the record is L2-warm)
	for (u32 i = lstart[tl + 1u] + threadIdx.x; i < lstart[tl + 2u]; i += blockDim.x) {
		const unsigned long long tt = top64[i];
		if (0 == tt) continue;
		const TileRec r = recs[(u32)tt];
		out_bits[i] = ((r.bits & 16u) && (r.bits & 32u) ? 4u : 0u) | ((r.bits >> 2) & 3u);  // (parked in the block's own entry until its step)
		out_pre[i] = r.pre_occ;
	}
	__syncthreads();
	if (0 == threadIdx.x) ctl->dbg[16] = wall_clock64();  // (diagnostics: ufomap_map_debug)
	// ---- 4. level by level to the root: lanes 8k .. 8k+7 of a wave take one block, lane = child ----
	auto step8 = [&](u32 i, bool have, u32 l) -> bool {
		const u32 sub = lane & 7u;
		const float v = have ? nocc[i][sub] : 0.f;
		const bool evaluated = have && 0 != dirty[i];
		// (all lanes run the shuffles; only the first lane of an evaluated block acts on the results)
		const unsigned long long tt = have ? top64[i] : 0ull;
		const u32 tc = ((u32)(tt >> 32) & 255u) - 1u;  // (>= 0 for an evaluated block: it has a touched child)
		u32 lub = 0;
		float luo = 0.f;
		if (evaluated) {
			const u32 who = (tl + 1u == l) ? i : (u32)tt;  // level 4: parked in the block's own entry by the tile pass
			lub = out_bits[who];
			luo = out_pre[who];
		}
		const bool reached = 0 != (lub & 4u);
		const float m = grpMax(v, 0);
		const u32 cv = (COLOR && have) ? nrgb[i][sub] : 0u;
		const bool eq = grpAllEq(v, 0, lane) && (!COLOR || grpAllEqU(cv, 0, lane));
		const u32 rgb = COLOR ? grpRgb(cv, 0) : 0u;
		const float pm = grpMax((reached && sub == tc) ? luo : v, 0);
		if (have && 0 == sub) {
			const u64 lk = nk[i];
			const u32 p = npar[i], ci = (u32)(lk & 7);
			u32 ob = 0;
			float op = 0.f;
			if (evaluated) {
				const u32 f = nflags[i];
				const u32 fl = ((f & F_CFREE) ? 1u : 0u) | ((f & F_CUNK) ? 2u : 0u);
				u32 pfl = fl;
				if (reached) {
					// summary with the top child as it was before its last update
					const u32 fsub = (f & ~((1u << tc) | (1u << (8 + tc)))) | ((lub & 1u) << tc) | (((lub >> 1) & 1u) << (8 + tc));
					pfl = ((fsub & F_CFREE) ? 1u : 0u) | ((fsub & F_CUNK) ? 2u : 0u);
				}
				const bool dead = reached && eq && 0 == (f & F_INNER);
				if (dead) {
					atomicOr(&nflags[i], F_DEAD);
					if (1 != lk) atomicAnd(&nflags[p], ~(1u << (16 + ci)));
				}
				if (1 == lk) {
					t.root->occ = m;
					t.root->flags = fl;
					if (COLOR) t.root->rgb = rgb;
				} else {
					const u32 fp = nflags[p];
					const u32 old_fl = ((fp >> ci) & 1u) | (((fp >> (8 + ci)) & 1u) << 1);
					const bool changed = nocc[p][ci] != m || old_fl != fl || (COLOR && nrgb[p][ci] != rgb);
					nocc[p][ci] = m;
					if (COLOR) nrgb[p][ci] = rgb;
					if (old_fl != fl) {
						const u32 setm = ((fl & 1u) << ci) | (((fl >> 1) & 1u) << (8 + ci));
						const u32 clrm = ((1u << ci) | (1u << (8 + ci))) & ~setm;
						if (setm) atomicOr(&nflags[p], setm);
						if (clrm) atomicAnd(&nflags[p], ~clrm);
					}
					const bool reach_out = reached && !((reached ? pm : m) == m && pfl == fl);
					ob = (reach_out ? 4u : 0u) | (pfl & 3u);
					op = reached ? pm : m;
					if (changed || reach_out) dirty[p] = 1;
				}
			}
			// what the parent needs if this block turns out to be its highest touched child
			out_bits[i] = ob;
			out_pre[i] = op;
			// the time of the last update beneath (last scan of the batch that touched the subtree, then the child it lies
			// under) travels up whether or not the block was evaluated
			if (p != NONE) atomicMax(&top64[p], (tt & 0xFFFFFF0000000000ull) | ((unsigned long long)(ci + 1u) << 32) | i);
		}
		return evaluated;
	};
	u32 l = tl + 1u;
	for (; l <= L; ++l) {
		const u32 lo = lstart[l], hi = lstart[l + 1];
		if (hi - lo <= 8u) break;  // (levels only get narrower towards the root)
		bool ev = false;
		for (u32 i0 = lo; i0 < hi; i0 += blockDim.x >> 3) {
			const u32 i = i0 + (threadIdx.x >> 3);
			ev |= step8(i, i < hi, l);
		}
		if (ev) lvl_dirty[l] = 1;
		__syncthreads();
		if (0 == lvl_dirty[l]) {
			l = L + 1;  // nothing was re-evaluated on this level: nothing above can change
			break;
		}
	}
	if (0 == threadIdx.x) ctl->dbg[17] = wall_clock64();  // (diagnostics: ufomap_map_debug)
	if (threadIdx.x < 64u) {
		bool ended = false;
		u32 dg_run = 0, dg_step = 0;
		unsigned long long dg_runc = 0, dg_stepc = 0;
		while (l <= L && !ended) {
			{
				const unsigned long long dg0 = clock64();
				++dg_step;
				const u32 lo = lstart[l], hi = lstart[l + 1];
				const u32 i = lo + (lane >> 3);
				const bool ev = step8(i, i < hi, l);
				if (0 == __ballot(ev)) {  // nothing was re-evaluated on this level: nothing above can change
					ended = true;
					break;
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				++l;
				dg_stepc += clock64() - dg0;
			}
			// ---- 4b. A RUN of levels with as many blocks as the level below: every block has ONE touched child, the block beneath it.
			// A map is centred on the origin and a sensor near it looks into several octants: from the scan's extent up to the root's
			// children every level holds the same two or four blocks -- about ten levels of a 16-level map, each a chain of dependent LDS
			// round trips in step8 (the block's state, its child's record, the parent's slot: ~0.7 us, whoever does them). Here the
			// group of eight lanes that takes a block keeps going upwards: what the block hands to its parent (summary, flags, "reached",
			// the summary before its last update) stays in REGISTERS -- moved to the parent's group by shuffles -- and the parent's own
			// state was asked for from LDS a level ahead: a level costs its arithmetic. Same rules as step8, restated for one child: the
			// child's hand-over is applied to the parent's copy (value, flags, "is inner" bit when the child collapsed; the parent is
			// re-evaluated iff that changed something or the child's last update reached it and changed it), then the block's own updateNode.
			if (l > L) break;
			const u32 nrun = lstart[l + 1] - lstart[l];
			if (nrun != lstart[l] - lstart[l - 1u] || nrun > 8u) continue;
			const u32 q = lane >> 3, sub = lane & 7u;
			const bool valid = q < nrun;
			// (the levels' first nodes, a level per lane: read across lanes where the loop needs them, not from LDS)
			const u32 ls_reg = lstart[min(lane, 25u)];
			auto lsAt = [&](u32 lv) -> u32 { return (u32)__builtin_amdgcn_readlane((int)ls_reg, (int)__builtin_amdgcn_readfirstlane((int)lv)); };
			u32 lb = l;  // the run's last level
			while (lb < L && lsAt(lb + 2u) - lsAt(lb + 1u) == nrun) ++lb;
			struct Lvl {
				u32 i, f, ci, ch, p, cc;
				float v;
				bool root;
			};
			auto loadLvl = [&](u32 lv) -> Lvl {
				Lvl x;
				x.i = lsAt(lv) + (valid ? q : 0u);
				x.v = nocc[x.i][sub];
				x.cc = COLOR ? nrgb[COLOR ? x.i : 0u][sub] : 0u;
				x.f = nflags[x.i];
				const u64 lk = nk[x.i];
				x.ci = (u32)(lk & 7);
				x.root = 1 == lk;
				x.ch = chld[x.i];
				x.p = npar[x.i];
				return x;
			};
			const unsigned long long dg1 = clock64();
			Lvl cur = loadLvl(l);
			// what step8 of the level below left for the run's first blocks
			bool ev = valid && 0 != dirty[cur.i];
			const unsigned long long tt0 = top64[cur.i];
			unsigned long long tt_hi = tt0 & 0xFFFFFF0000000000ull;
			u32 tc = ((u32)(tt0 >> 32) & 255u) - 1u;
			const u32 who = min((u32)tt0, UFO_UPPER_MAX - 1u);
			u32 lub = out_bits[who];
			float luo = out_pre[who];
			for (;;) {  // (uniform)
				++dg_run;
				const bool more = l < lb;  // the level above belongs to the run as well
				Lvl nxt = cur;
				if (more) nxt = loadLvl(l + 1u);  // (asked for now, looked at after this level's arithmetic)
				if (0 == __ballot(ev)) {
					ended = true;
					break;
				}
				// updateNode of the group's block (OMB:1195-1224), as in step8
				const bool reached = ev && 0 != (lub & 4u);
				const float m = grpMax(cur.v, 0);
				const bool eq = grpAllEq(cur.v, 0, lane) && (!COLOR || grpAllEqU(cur.cc, 0, lane));
				const u32 rgb = COLOR ? grpRgb(cur.cc, 0) : 0u;
				const float pm = grpMax((reached && sub == tc) ? luo : cur.v, 0);
				const u32 fl = ((cur.f & F_CFREE) ? 1u : 0u) | ((cur.f & F_CUNK) ? 2u : 0u);
				u32 pfl = fl;
				if (reached) {
					const u32 fsub = (cur.f & ~((1u << tc) | (1u << (8 + tc)))) | ((lub & 1u) << tc) | (((lub >> 1) & 1u) << (8 + tc));
					pfl = ((fsub & F_CFREE) ? 1u : 0u) | ((fsub & F_CUNK) ? 2u : 0u);
				}
				const bool dead = ev && reached && eq && 0 == (cur.f & F_INNER);
				const bool reach_out = reached && !(pm == m && pfl == fl);
				if (dead) cur.f |= F_DEAD;
				if (ev && cur.root && 0 == sub) {
					t.root->occ = m;
					t.root->flags = fl;
					if (COLOR) t.root->rgb = rgb;
				}
				// the block goes back to LDS (step 5 stores it to the table)
				if (valid) {
					nocc[cur.i][sub] = cur.v;
					if (COLOR) nrgb[COLOR ? cur.i : 0u][sub] = cur.cc;
					if (0 == sub) nflags[cur.i] = cur.f;
				}
				const u32 ob = ev ? ((reach_out ? 4u : 0u) | (pfl & 3u)) : 0u;
				const float op = ev ? (reached ? pm : m) : 0.f;
				if (!more) {
					// the run's last level hands over through LDS, as step8 does: the level above merges chains (or there is none)
					if (valid && 0 == sub) {
						const u32 pp = cur.p, ci = cur.ci;
						if (ev && !cur.root) {
							if (dead) atomicAnd(&nflags[pp], ~(1u << (16 + ci)));
							const u32 fp = nflags[pp];
							const u32 old_fl = ((fp >> ci) & 1u) | (((fp >> (8 + ci)) & 1u) << 1);
							const bool changed = nocc[pp][ci] != m || old_fl != fl || (COLOR && nrgb[COLOR ? pp : 0u][ci] != rgb);
							nocc[pp][ci] = m;
							if (COLOR) nrgb[COLOR ? pp : 0u][ci] = rgb;
							if (old_fl != fl) {
								const u32 setm = ((fl & 1u) << ci) | (((fl >> 1) & 1u) << (8 + ci));
								const u32 clrm = ((1u << ci) | (1u << (8 + ci))) & ~setm;
								if (setm) atomicOr(&nflags[pp], setm);
								if (clrm) atomicAnd(&nflags[pp], ~clrm);
							}
							if (changed || reach_out) dirty[pp] = 1;
						}
						out_bits[cur.i] = ob;
						out_pre[cur.i] = op;
						if (pp != NONE) atomicMax(&top64[pp], tt_hi | ((unsigned long long)(ci + 1u) << 32) | cur.i);
					}
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
					++l;
					break;
				}
				// ... every other level through registers: the parent's group fetches what its child's group computed
				const u32 cq = valid ? (nxt.ch - lsAt(l)) & 7u : 0u;  // the child's group (chld: filled beside step 2)
				const int src = (int)(8u * cq);
				const bool c_ev = 0 != __shfl(ev ? 1 : 0, src);
				const float c_m = __shfl(m, src);
				const u32 c_fl = (u32)__shfl((int)fl, src);
				const u32 c_rgb = COLOR ? (u32)__shfl((int)rgb, src) : 0u;
				const bool c_dead = 0 != __shfl(dead ? 1 : 0, src);
				const bool c_ro = 0 != __shfl(reach_out ? 1 : 0, src);
				const u32 c_ob = (u32)__shfl((int)ob, src);
				const float c_op = __shfl(op, src);
				const u32 c_ci = (u32)__shfl((int)cur.ci, src);
				const u32 thi_lo = (u32)__shfl((int)(u32)(tt_hi >> 32), src);
				tt_hi = (unsigned long long)thi_lo << 32;
				cur = nxt;
				ev = false;
				if (valid && c_ev) {
					const u32 old_fl = ((cur.f >> c_ci) & 1u) | (((cur.f >> (8 + c_ci)) & 1u) << 1);
					bool changed = old_fl != c_fl;
					if (sub == c_ci) {
						changed = changed || cur.v != c_m || (COLOR && cur.cc != c_rgb);
						cur.v = c_m;
						if (COLOR) cur.cc = c_rgb;
					}
					changed = 0 != grpOr(changed ? 1u : 0u, 0);
					cur.f = (cur.f & ~((1u << c_ci) | (1u << (8 + c_ci)))) | ((c_fl & 1u) << c_ci) | (((c_fl >> 1) & 1u) << (8 + c_ci));
					if (c_dead) cur.f &= ~(1u << (16 + c_ci));
					ev = changed || c_ro;
				}
				tc = c_ci;
				lub = c_ob;
				luo = c_op;
				++l;
			}
			dg_runc += clock64() - dg1;
		}
		if (0 == threadIdx.x) {  // (diagnostics: levels by step8 / in runs, their clocks)
			ctl->dbg[24] = dg_step | ((unsigned long long)dg_run << 32);
			ctl->dbg[25] = dg_stepc;
			ctl->dbg[26] = dg_runc;
		}
	}
	__syncthreads();
	if (0 == threadIdx.x) ctl->dbg[18] = wall_clock64();  // (diagnostics: ufomap_map_debug)
	// ---- 5. every block back to the table, once ----
	for (u32 i = threadIdx.x; i < U; i += blockDim.x) {
		const u32 s = nslot[i];
		if (s == NONE) continue;
		float4* po = reinterpret_cast<float4*>(t.occ(s));
		const float4* li = reinterpret_cast<const float4*>(nocc[i]);
		po[0] = li[0];
		po[1] = li[1];
		if (COLOR) {
			uint4* pc = reinterpret_cast<uint4*>(t.rgb + 8 * (size_t)s);
			const uint4* lc = reinterpret_cast<const uint4*>(nrgb[i]);
			pc[0] = lc[0];
			pc[1] = lc[1];
		}
		t.flags(s) = nflags[i];
		if (ncreated[i]) t.parent(s) = (npar[i] != NONE) ? nslot[npar[i]] : NONE;
	}
	// this kernel is the tile bitmaps' last reader: leave them empty for their sets' next scans
	for (u32 b = 0; b < B; ++b)
		for (u32 j = threadIdx.x; j < nwords3; j += blockDim.x) UFO_DESC(b).tile_bits[j] = 0;
	if (up_bits)
		for (u32 j = threadIdx.x; j < nwords; j += blockDim.x) up_bits[j] = 0;
	if (0 == threadIdx.x) {
		ctl->dbg[19] = wall_clock64();
		ctl->dbg[20] = U | ((unsigned long long)l << 32);
		ctl->dbg[45] = B;      // (scans this walk applied: the host's statistics)
		tsMark(p->ts, f, 6, wall_clock64());
	}
	// The finished control blocks go to the host's pinned copies from here (no read-back copy, no stream synchronisation on
	// the host: it polls the word behind a block and reads), and the device copies return to the start state of a scan
	// (no upload before their sets' next scans). A walk that stood back has left above: the host falls back to copies.
	// What the host reads is the block up to the diagnostics (504 bytes; the clocks behind them only when somebody asked:
	// option "ctl_dbg" -- every word is a write across PCIe that the kernel's last barrier waits for).
	__syncthreads();
	const u32 e = __hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (raised during the walk: ERR_TABLE_FULL)
	{
		const u32* init = reinterpret_cast<const u32*>(ctl_init);
		const u32 used = sh_used, used_g = sh_ng, used_u = sh_nu;
		constexpr u32 W = sizeof(ScanCtl) / 4u, W_USED = offsetof(ScanCtl, used_now) / 4u, W_ERR = offsetof(ScanCtl, err) / 4u;
		constexpr u32 W_USEDG = offsetof(ScanCtl, used_g_now) / 4u, W_USEDU = offsetof(ScanCtl, used_u_now) / 4u;
		constexpr u32 W_CORE = offsetof(ScanCtl, dbg) / 4u, W_WALK = offsetof(ScanCtl, walk_scans) / 4u;
		const u32 wrep = report_dbg ? W : W_CORE;
		for (u32 k = threadIdx.x; k < B * W; k += blockDim.x) {
			const u32 b = k / W, w = k % W;
			u32* dev = reinterpret_cast<u32*>(UFO_DESC(b).ctl);
			if (w < wrep) {
				u32* host = reinterpret_cast<u32*>(UFO_DESC(b).host_result);
				u32 x = __hip_atomic_load(&dev[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (W_USED == w) x = used;   // the table's fill after the walk: in every scan's block (the host reads whichever it joins)
				if (W_USEDG == w) x = used_g;  // ... per region too (a scan that is not the walk's last would report 0 / 0 and the host
				if (W_USEDU == w) x = used_u;  // would size the next update as if the table were empty: ADVICE r4)
				if (W_ERR == w) x |= e;      // a failed walk has failed for all of its scans
				if (W_WALK == w) x = (b + 1u == B) ? B : 0u;
				host[w] = x;
			}
			if (0 == e) dev[w] = w < W_CORE ? init[w] : 0u;  // (an error raised in this very kernel stays for the host to read)
		}
	}
	__syncthreads();  // (every thread's stores to the pinned blocks have been acknowledged: the barrier waits for them)
	if (report_dbg && threadIdx.x + 1u == B) UFO_DESC(threadIdx.x).host_result->dbg[23] = wall_clock64();  // (diagnostics: the report's duration)
	if (threadIdx.x < B) {  // (the kernel's last actions; the walk enqueued behind this one looks at the status words)
		const ScanDesc& d = UFO_DESC(threadIdx.x);
		p->wstat[(sl.first + threadIdx.x) & (UFO_RING - 1u)] = e ? 1u : 0u;
		__hip_atomic_store(reinterpret_cast<unsigned long long*>(d.host_result + 1), d.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}
#undef UFO_DESC

// Stream-to-stream hand-overs without events: a one-thread kernel at the end of the producing stream's work stores the
// scan's number into a word of device memory (k_signal), a one-wave kernel in front of the consuming stream's work waits
// for it (k_gate; a single wave cannot keep anything from being scheduled). Measured per hand-over, stream idle time
// included (scripts/micro/stream_wait*.hip and rocprofv3 traces of the pipeline): event record + wait 9-15 us,
// hipStreamWaitValue64 on signal memory 5-7 us (it is a polling kernel too, on host-coherent memory), this 2-3 us.
__global__ void k_signal(unsigned long long* flag, unsigned long long value, unsigned long long* host_flag, unsigned long long* ts, unsigned long long f)
{
	tsMark(ts, f, 0, wall_clock64());
	__hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
	// (pinned host memory: the call that enqueued the scan returns once k_fhits has consumed the caller's cloud)
	if (host_flag) __hip_atomic_store(host_flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// (The wait is bounded: a tool that serialises kernels across streams -- rocprofv3 --pmc does -- would keep the producer
// from ever running while this wave spins. The host does not use gates when it sees such a tool, ufomap_hip.hip:
// useGates; should one slip through, the gate gives up after ~2 s and flags the scan, which then leaves the map alone.)
__device__ __forceinline__ void gateWait(const unsigned long long* flag, unsigned long long value, ScanCtl* ctl, unsigned long long max_ticks,
                                         unsigned long long* ts, unsigned long long f)
{
	const unsigned long long t0 = wall_clock64();
	tsMark(ts, f, 1, t0);
	while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < value) {
		__builtin_amdgcn_s_sleep(1);
		if (wall_clock64() - t0 > max_ticks) {  // 100 MHz clock
			atomicOr(&ctl->err, ERR_GATE);
			return;
		}
	}
	tsMark(ts, f, 2, wall_clock64());
}
__global__ void k_gate(const unsigned long long* flag, unsigned long long value, ScanCtl* ctl, unsigned long long max_ticks, unsigned long long* ts,
                       unsigned long long f)
{
	gateWait(flag, value, ctl, max_ticks, ts, f);
}
// Round 6: a gate that waits for the HOST -- a pageable cloud is copied into the set's pinned staging buffer by a helper thread while the
// calling thread enqueues the scan (ufomap_hip.hip: uploadCloud); the asynchronous H2D copy of the staging buffer sits behind this
// one-wave kernel on the prep stream, which polls the pinned word the helper stores when the copy is complete. (Nothing a tool that
// serialises kernels could deadlock: the producer is a host thread.)
__global__ void k_host_gate(const unsigned long long* flag, unsigned long long value, unsigned long long max_ticks, u32* err_out)
{
	if (0 != threadIdx.x) return;
	const unsigned long long t0 = wall_clock64();
	while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < value) {
		__builtin_amdgcn_s_sleep(8);
		if (wall_clock64() - t0 > max_ticks) {  // (100 MHz clock; cannot happen: the call joins the helper before it returns)
			if (err_out) atomicOr(err_out, ERR_GATE);
			return;
		}
	}
}
// A pageable cloud on its way from the helper thread's staging buffer (pinned) to HBM, piece by piece: the workgroups of piece k wait
// for the helper's word to say that piece k has been copied, then read it across PCIe themselves -- the cloud is in HBM one piece's
// transfer after the helper's last byte (a call that waits for its scan: 40 us sooner than behind a DMA transfer of the whole cloud).
// Measured alternatives (profiles/r06_ab_experiments.log): hipMemcpyAsync per piece -- every extra copy costs the stream ~20 us;
// more than ~32 workgroups -- slower, 128 of them 3x; the word passed on through HBM by one polling wave -- no different.
__global__ __launch_bounds__(256) void k_stage_copy(const unsigned long long* flag, unsigned long long job, u32 first_piece, const uint8_t* __restrict__ src,
                                                    uint8_t* __restrict__ dst, unsigned long long bytes, unsigned long long chunk, u32 wgs_per_piece,
                                                    unsigned long long max_ticks)
{
	const u32 piece = blockIdx.x / wgs_per_piece, w = blockIdx.x % wgs_per_piece;
	if (0 == threadIdx.x) {
		const unsigned long long want = (job << 8) | (unsigned long long)(first_piece + piece + 1u), t0 = wall_clock64();
		while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
			__builtin_amdgcn_s_sleep(8);
			if (wall_clock64() - t0 > max_ticks) break;  // (100 MHz clock; cannot happen: the call joins the helper before it returns)
		}
	}
	__syncthreads();
	const unsigned long long lo = (unsigned long long)piece * chunk, hi = min(lo + chunk, bytes);  // (chunk: a multiple of 16)
	if (lo >= hi) return;
	const unsigned long long n16 = (hi - lo) >> 4;
	const uint4* s4 = reinterpret_cast<const uint4*>(src + lo);
	uint4* d4 = reinterpret_cast<uint4*>(dst + lo);
	for (unsigned long long i = (unsigned long long)w * blockDim.x + threadIdx.x; i < n16; i += (unsigned long long)wgs_per_piece * blockDim.x) d4[i] = s4[i];
	if (0 == w && threadIdx.x < (u32)((hi - lo) & 15ull)) dst[lo + (n16 << 4) + threadIdx.x] = src[lo + (n16 << 4) + threadIdx.x];
}
// The end of one scan half and the gate of the next in ONE launch (asynchronous calls in a row: the host keeps the
// descriptor of scan i back and hands it over with the gate of scan i+1 -- one one-wave kernel per scan on the scan stream
// instead of two; whatever needs scan i before another scan arrives publishes it with k_scan_done, ufomap_hip.hip:
// publishScanDone).
__global__ void k_done_gate(Pipe* p, ScanDesc d, const unsigned long long* flag, unsigned long long value, ScanCtl* ctl, unsigned long long max_ticks,
                            unsigned long long f)
{
	tsMark(p->ts, d.fseq, 3, wall_clock64());
	p->ring[d.fseq & (UFO_RING - 1u)] = d;
	__hip_atomic_store(&p->scan_done, d.fseq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
	gateWait(flag, value, ctl, max_ticks, p->ts, f);
}

}  // namespace ufo
