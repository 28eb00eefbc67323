// vol_kernels.h -- the VOLUME path: a depth-0 scan whose ray grid is far beyond anything the steady-state path of
// fast_kernels.h holds (a 2 mm RGB-D frame at 5 m: 1 500 x 1 800 x 1 400 cells, 7 M depth-3 tiles, 5 * 10^8 DDA steps, 3.5 * 10^8
// voxels updated -- BASELINE configs[2] at insert depth 0, the one bandwidth-bound configuration), through the SAME tiled
// tree update (k_tile / k_up / k_ftail) instead of the general path's update list:
//
//   k_classify, k_select, k_reduce_boxes   head loops, first point per voxel, boxes (scan_kernels.h; the boxes are read back)
//   k_vhits      the voxels that receive a hit -> brick grid H
//   k_vcut       (round 5) every ray cut into segments of ~192 cells: the exact state at each cut from the three addition chains
//   k_vwalk      (round 5) freeSpace (occupancy_map_base.h:1229-1301; computeRayInit / computeRayTakeStep octree.h:1192-1233): one
//                lane per SEGMENT, the reference's FP64 recurrence; marks go to brick grid M through a register that collects the
//                bits of the brick the ray is in, a per-lane queue and a per-wave write-combining table
//                (k_vdda: round 4's form, one lane per whole ray -- the cross-check, option vol_mode bit 4)
//   k_vlist      the tiles that hold a ray cell -> list
//   k_tile<VOL>  one wave per listed tile (fast_kernels.h): everything beneath depth 3, each block record read and written once
//   k_up x n     levels 4, 5, ... in parallel, eight lanes per block, until what is left above fits k_ftail's LDS
//   k_ftail      the rest, up to the root; finished control block to pinned memory
//
// The grids are TILE-MAJOR: per depth-3 tile (8x8x8 cells) eight 64-bit words, one per 4x4x4-cell brick (brick = bx | by << 1 |
// bz << 2 inside the tile, bit = x | y << 2 | z << 4 inside the brick). A ray stays inside a brick for 4-6 steps whatever its
// direction (a row-major bit grid keeps only rays along x together), a tile's bits are ONE 64-byte line for k_tile, and the
// active-tile list is a streaming pass. What the general path does with these scans -- a 16-byte update-list entry per node
// block, find-or-create and read-modify-write of 5 * 10^7 hashed 64-byte records, a launch per tree level -- was 27 of its
// 40 ms.
#pragma once
#include "fast_kernels.h"

namespace ufo
{
// the tile grid of the scan (FastGeo::tbase / nt / ntiles, tl = 3) and the cell coordinate of its corner
struct VolGeo {
	i32 cbase[3];  // = 8 * tbase
	u32 nt[3];
	u32 ntiles;
};

// the tile grids of the levels the walk runs in parallel (host side: host_vol.inl)
struct VolPlan {
	FastGeo lv[24];  // lv[0]: the depth-3 tiles; lv[k]: the cells of level 3 + k (k_up's k-th launch writes them)
	int n = 0;       // levels in lv[]; k_ftail starts above lv[n - 1]
	u64 rec_total = 0;  // hand-over records of all levels
	VolGeo vg{};
};

// words of one XCD's tile bitmap: whole 256-byte pieces, so that no cache line holds bits of two XCDs (each XCD's atomics stay in
// its own L2 until the kernel ends)
__host__ __device__ inline u32 volTbWords(u32 ntiles) { return (((ntiles + 31u) >> 5) + 63u) & ~63u; }

__device__ __forceinline__ void volWordBit(const VolGeo& vg, u32 x, u32 y, u32 z, u32* word, u32* bit)
{
	const u32 tile = ((z >> 3) * vg.nt[1] + (y >> 3)) * vg.nt[0] + (x >> 3);
	*word = tile * 8u + (((x >> 2) & 1u) | (((y >> 2) & 1u) << 1) | (((z >> 2) & 1u) << 2));
	*bit = (x & 3u) | ((y & 3u) << 2) | ((z & 3u) << 4);
}

// the voxels that receive a hit (k_select's list of first points' codes) -> H
__global__ __launch_bounds__(256) void k_vhits(MapGeom g, VolGeo vg, const u64* __restrict__ hit_code, const ScanCtl* ctl_in, u64* __restrict__ H, ScanCtl* ctl)
{
	const u32 n = ctl_in->n_hits;
	for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const u64 c = hit_code[i];
		const i32 x = (i32)compact3(c) - vg.cbase[0], y = (i32)compact3(c >> 1) - vg.cbase[1], z = (i32)compact3(c >> 2) - vg.cbase[2];
		if (x < 0 || y < 0 || z < 0 || (u32)x >= 8u * vg.nt[0] || (u32)y >= 8u * vg.nt[1] || (u32)z >= 8u * vg.nt[2]) {
			atomicOr(&ctl->err, ERR_GRID_OOB);  // (cannot happen: the hit box lies inside the ray box)
			continue;
		}
		u32 w, b;
		volWordBit(vg, (u32)x, (u32)y, (u32)z, &w, &b);
		atomicOr(reinterpret_cast<unsigned long long*>(&H[w]), 1ull << b);
	}
}

// ---- rays bundled by direction -----------------------------------------------------------------------------------------
// The marks of a scan are a set: the order the rays are cast in changes nothing but speed. Cast in the cloud's order, a wave's
// 64 rays are a strip of an image row (or a piece of a LiDAR ring): far from the sensor they are two cells apart, so few
// of them are in the same brick at the same time. Sorted by direction -- a counting sort over a cube map of 6 x 128 x 128 bins --
// a wave's rays form a 2D bundle (an 8 x 8 pixel patch of a 640 x 480 frame): they share bricks all the way, and what they
// flush is merged in LDS before it reaches the L2 (k_vdda: the per-wave table).
#define UFO_VBINS (6u * 128u * 128u)
__device__ __forceinline__ u32 dirBin(const D3& sensor, const D3& end)
{
	const double dx = end.x - sensor.x, dy = end.y - sensor.y, dz = end.z - sensor.z;
	const double ax = fabs(dx), ay = fabs(dy), az = fabs(dz);
	u32 face;
	double m, a, b;
	if (ax >= ay && ax >= az) {
		face = dx < 0 ? 1u : 0u;
		m = ax;
		a = dy;
		b = dz;
	} else if (ay >= az) {
		face = dy < 0 ? 3u : 2u;
		m = ay;
		a = dx;
		b = dz;
	} else {
		face = dz < 0 ? 5u : 4u;
		m = az;
		a = dx;
		b = dy;
	}
	if (!(m > 0)) return 0u;
	const int ia = min(127, max(0, (int)((a / m + 1.0) * 64.0))), ib = min(127, max(0, (int)((b / m + 1.0) * 64.0)));
	// (8 x 8 bins form a block of 64 consecutive numbers: neighbouring bins of a sparse scan -- a LiDAR's -- stay neighbours)
	return face * 16384u + ((u32)(ib >> 3) * 16u + (u32)(ia >> 3)) * 64u + (u32)(ib & 7) * 8u + (u32)(ia & 7);
}
__global__ __launch_bounds__(256) void k_vbin_count(D3 sensor, const D3* __restrict__ ray_end, const ScanCtl* ctl_in, u32* __restrict__ bin_of, u32* __restrict__ hist)
{
	const u32 n = ctl_in->n_rays;
	const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n) return;
	const u32 b = dirBin(sensor, ray_end[r]);
	bin_of[r] = b;
	atomicAdd(&hist[b], 1u);
}
// exclusive prefix over the bins, in two steps (round 5; the single workgroup below walked 96 bins per thread with stride-96 loads:
// 58 us): every workgroup scans ITS 1024 bins in place (coalesced) and leaves their sum; the scatter adds the sums of the
// workgroups before (96 values, scanned by every workgroup of the scatter for itself)
__global__ __launch_bounds__(1024) void k_vbin_scan1(u32* __restrict__ hist, u32* __restrict__ tot)
{
	__shared__ u32 wsum[16];
	const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6, i = blockIdx.x * 1024u + t;
	const u32 c = hist[i];
	u32 incl = c;
	for (int o = 1; o < 64; o <<= 1) {
		const u32 v = __shfl_up(incl, o);
		if ((int)lane >= o) incl += v;
	}
	if (63u == lane) wsum[wave] = incl;
	__syncthreads();
	u32 base = 0, all = 0;
	for (u32 w = 0; w < 16u; ++w) {
		if (w < wave) base += wsum[w];
		all += wsum[w];
	}
	hist[i] = base + incl - c;
	if (0 == t) tot[blockIdx.x] = all;
}
// (the round-4 form, kept for reference and for tests of the new one)
__global__ __launch_bounds__(1024) void k_vbin_scan(u32* __restrict__ hist)
{
	__shared__ u32 wsum[16];
	constexpr u32 PER = UFO_VBINS / 1024u;
	const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6;
	u32 loc = 0;
	for (u32 k = 0; k < PER; ++k) loc += hist[t * PER + k];
	u32 incl = loc;
	for (int o = 1; o < 64; o <<= 1) {
		const u32 v = __shfl_up(incl, o);
		if ((int)lane >= o) incl += v;
	}
	if (63u == lane) wsum[wave] = incl;
	__syncthreads();
	u32 base = incl - loc;
	for (u32 w = 0; w < wave; ++w) base += wsum[w];
	for (u32 k = 0; k < PER; ++k) {
		const u32 c = hist[t * PER + k];
		hist[t * PER + k] = base;
		base += c;
	}
}
__global__ __launch_bounds__(256) void k_vbin_scatter(const ScanCtl* ctl_in, const u32* __restrict__ bin_of, u32* __restrict__ offs, u32* __restrict__ order,
                                                      const u32* __restrict__ tot)
{
	// (tot: the sums of k_vbin_scan1's workgroups, UFO_VBINS / 1024 = 96 of them; nullptr: offs holds the full prefix already)
	__shared__ u32 bbase[UFO_VBINS / 1024u];
	if (tot) {
		if (threadIdx.x < 64u) {
			// two values per lane of the first wave: exclusive prefix over 96 sums
			const u32 a = tot[threadIdx.x], b = (threadIdx.x + 64u < UFO_VBINS / 1024u) ? tot[threadIdx.x + 64u] : 0u;
			u32 ia = a;
			for (int o = 1; o < 64; o <<= 1) {
				const u32 v = __shfl_up(ia, o);
				if ((int)threadIdx.x >= o) ia += v;
			}
			const u32 first64 = __shfl(ia, 63);
			u32 ib = b;
			for (int o = 1; o < 64; o <<= 1) {
				const u32 v = __shfl_up(ib, o);
				if ((int)threadIdx.x >= o) ib += v;
			}
			bbase[threadIdx.x] = ia - a;
			if (threadIdx.x + 64u < UFO_VBINS / 1024u) bbase[threadIdx.x + 64u] = first64 + ib - b;
		}
		__syncthreads();
	}
	const u32 n = ctl_in->n_rays;
	const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n) return;
	const u32 bin = bin_of[r];
	order[(tot ? bbase[bin >> 10] : 0u) + atomicAdd(&offs[bin], 1u)] = r;
}

// The XCD this wave runs on (HW_REG_XCC_ID, bits 3:0). Waves that read the same value share one L2.
__device__ __forceinline__ u32 xccId() { return (u32)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; }

// What the per-XCD copies rest on, checked once per process before the volume path is used (volSelfTest, host_vol.inl): (1) XCC_ID
// reads 0 .. 7; (2) read-modify-write atomics at WORKGROUP scope issued by different workgroups that read the same XCC_ID on one
// word lose no update -- i.e. they are executed by one L2, the XCD's -- here: every lane adds 1 to the counter of its XCD,
// UFO_VSELF_ADDS times, from 8 * UFO_VSELF_BLOCKS workgroups; the counters must add up to the number of additions, and the plain
// loads of the NEXT kernel (the host's copy) must see them: the lines are written back when the kernel ends. A part (or a compiler
// mode) on which this does not hold fails the test, and the volume path stays off: scans take the general path.
#define UFO_VSELF_BLOCKS 64u
#define UFO_VSELF_ADDS 16u
__global__ __launch_bounds__(256) void k_vselftest(u32* __restrict__ cnt, u32* __restrict__ bad)
{
	const u32 raw = (u32)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
	if ((raw & 15u) > 7u) atomicOr(bad, 1u);
	u32* c = cnt + (raw & 7u) * 64u;  // (a 256-byte piece per XCD, like the copies)
	for (u32 k = 0; k < UFO_VSELF_ADDS; ++k) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// freeSpaceNormal, one lane per ray (the general path's k_dda, marks collected per brick).
// Where the marks go: a device-scope atomic is executed at the memory side of the fabric (the eight XCDs' L2s are not
// coherent with each other) -- 10-35 per ns on this part, and 1e8 of them were 9 of this kernel's 9.6 ms. So every XCD marks
// a copy of M OF ITS OWN, picked by the hardware's XCC id (a fact about where the wave runs, not an assumption about
// dispatch), with atomics that need no more than the XCD's L2 to be atomic (workgroup scope: no sc1, executed by the L2 all
// CUs of the XCD share); the lines are written back when the kernel ends like any plain store's. k_tile ORs the copies a
// tile was marked in -- the per-XCD tile bitmaps say which -- and leaves them zeroed. Blocks are mapped to rays so that
// (with the usual block b -> XCD b % 8 placement; speed only) an XCD takes one contiguous eighth of the cloud: neighbouring
// rays share bricks, and a tile is marked in one or two copies, not eight.
#define UFO_VWC 256u  // entries of a wave's write-combining table (k_vdda)
#define UFO_VSTAGE 4u  // (brick word, bits) entries a lane of k_vwalk parks before the wave sends them on
__global__ __launch_bounds__(256) void k_vdda(MapGeom g, D3 sensor, Grid gr, VolGeo vg, u64* __restrict__ Mx, u32* __restrict__ tbx, const D3* __restrict__ ray_end,
                                              const ScanCtl* ctl_in, ScanCtl* ctl, u32 mode, const u32* __restrict__ order)
{
	// (mode, a measuring aid: bit 1 = blocks take the rays in launch order, bit 3 = no write-combining table; order == nullptr:
	// the rays in the cloud's order. Round 4's bit 0 -- ONE copy for all XCDs, marked with the same workgroup-scope atomics -- is gone:
	// by this file's own argument that is wrong across XCDs.)
	// The wave's write-combining table: (brick word, bits) pairs on their way to the XCD's copy. A lane that leaves a brick ORs
	// its bits into the brick's entry if there is one; else it takes the entry over and sends what was in it to the L2 -- the
	// rays of a bundle enter and leave the same bricks within a few steps of one another, so most flushes end here.
	__shared__ u32 wc_key[4][UFO_VWC];
	__shared__ unsigned long long wc_mask[4][UFO_VWC];
	const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	const bool use_wc = 0 == (mode & 8u);
	for (u32 k = lane; k < UFO_VWC; k += 64u) {
		wc_key[wave][k] = 0xFFFFFFFFu;
		wc_mask[wave][k] = 0ull;
	}
	const u32 n = ctl_in->n_rays;
	const u32 per = gridDim.x >> 3;  // (the grid is a multiple of 8 blocks)
	u32 i = ((mode & 2u) ? blockIdx.x : ((blockIdx.x & 7u) * per + (blockIdx.x >> 3))) * blockDim.x + threadIdx.x;
	const bool live = i < n;
	if (live && order) i = order[i];
	const u32 xcc = xccId();
	u64* const M = Mx + (size_t)xcc * volCopyWords(vg.ntiles);
	u32* const tb = tbx + (size_t)xcc * (size_t)volTbWords(vg.ntiles);
	// a lane's (word, bits) on its way out: through the table, or straight to the L2. Called by any subset of a wave's lanes at
	// the same program point; LDS operations of one wave are executed in program order.
	auto flush = [&](u32 w, u64 bits) {
		if (!use_wc) {
			__hip_atomic_fetch_or(&M[w], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			return;
		}
		const u32 slot = (w * 0x9E3779B1u) >> 24;  // 8 bits
		const u32 k = wc_key[wave][slot];
		const bool hit = k == w;
		// 1. lanes whose brick holds the entry: their bits join it (before anybody takes the entry over, below)
		if (hit) __hip_atomic_fetch_or(&wc_mask[wave][slot], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		u32 prev = k;
		bool won = false;
		// 2. the others try to take it over: one lane per entry wins ...
		if (!hit) {
			prev = atomicCAS(&wc_key[wave][slot], k, w);
			won = prev == k;
		}
		// ... gets what was in it and sends that on its way
		if (won) {
			const u64 old = atomicExch(&wc_mask[wave][slot], (unsigned long long)bits);
			if (k != 0xFFFFFFFFu && old) __hip_atomic_fetch_or(&M[k], old, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		// 3. who lost to a lane with the SAME brick joins the new entry (after the winner's exchange); who lost to another
		// brick goes straight to the L2
		if (!hit && !won) {
			if (prev == w) __hip_atomic_fetch_or(&wc_mask[wave][slot], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			else __hip_atomic_fetch_or(&M[w], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
	};
	unsigned long long steps = 0;
	u32 err = 0;
	if (live) {
		RayState r;
		raySetup(g, sensor, 0u, gr, ray_end[i], r);
		if (3 == r.status) {
			err |= ERR_VOL;  // clipped at the map cube / outside the grid's interior: the checked walk of the general path
		} else if (1 == r.status) {
			u32 w, b;
			volWordBit(vg, (u32)(r.start[0] - vg.cbase[0]), (u32)(r.start[1] - vg.cbase[1]), (u32)(r.start[2] - vg.cbase[2]), &w, &b);
			flush(w, 1ull << b);
			__hip_atomic_fetch_or(&tb[w >> 8], 1u << ((w >> 3) & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			steps = 1;
		} else if (2 == r.status) {
			u32 x = (u32)(r.start[0] - vg.cbase[0]), y = (u32)(r.start[1] - vg.cbase[1]), z = (u32)(r.start[2] - vg.cbase[2]);
			const u32 gx = (u32)(r.goal[0] - vg.cbase[0]), gy = (u32)(r.goal[1] - vg.cbase[1]), gz = (u32)(r.goal[2] - vg.cbase[2]);
			const u32 sx = (u32)(i32)r.s[0], sy = (u32)(i32)r.s[1], sz = (u32)(i32)r.s[2];
			double tmx = r.tm[0], tmy = r.tm[1], tmz = r.tm[2];
			const double tdx = r.td[0], tdy = r.td[1], tdz = r.td[2];
			const long long idist = __double_as_longlong(r.dist);  // (non-negative doubles order like their bit patterns)
			const u32 budget = (u32)min(3ull * (1ull << g.L) + 8ull, 0xFFFFFFF0ull);
			const u32 nt0 = vg.nt[0], nt1 = vg.nt[1];
			u32 cnt = 0, curw = 0xFFFFFFFFu;
			u64 acc = 0;
			bool go;
			do {
				++cnt;
				const u32 tile = ((z >> 3) * nt1 + (y >> 3)) * nt0 + (x >> 3);
				const u32 w = tile * 8u + (((x >> 2) & 1u) | (((y >> 2) & 1u) << 1) | (((z >> 2) & 1u) << 2));
				const u32 b = (x & 3u) | ((y & 3u) << 2) | ((z & 3u) << 4);
				if (w != curw) {
					if (acc) flush(curw, acc);
					// (a new tile: its bit in the XCD's tile bitmap -- looked at first: bits only appear during this kernel, the CU's L1
					// was invalidated when it started, and a stale 0 costs one more atomic; 6e7 atomics were 2 ms of this kernel)
					if (((w ^ curw) >> 3) && !((tb[tile >> 5] >> (tile & 31u)) & 1u))
						__hip_atomic_fetch_or(&tb[tile >> 5], 1u << (tile & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					curw = w;
					acc = 0;
				}
				acc |= 1ull << b;
				// minElementIndex (vector3.h:244-251): x if tx <= ty and tx <= tz, else y if ty <= tz, else z; only the chosen
				// axis' t_max is touched (octree.h:1227-1233)
				const bool cxy = tmx <= tmy, cxz = tmx <= tmz, cyz = tmy <= tmz;
				const bool selx = cxy & cxz;
				const bool sely = !cxy & cyz;
				const bool selz = !(selx | sely);
				x += selx ? sx : 0u;
				y += sely ? sy : 0u;
				z += selz ? sz : 0u;
				const double nx = tmx + tdx, ny = tmy + tdy, nz = tmz + tdz;
				tmx = selx ? nx : tmx;
				tmy = sely ? ny : tmy;
				tmz = selz ? nz : tmz;
				// t_max.min() <= distance (OMB:1300)
				const bool more = (__double_as_longlong(tmx) <= idist) | (__double_as_longlong(tmy) <= idist) | (__double_as_longlong(tmz) <= idist);
				go = (((x ^ gx) | (y ^ gy) | (z ^ gz)) != 0u) & more & (cnt < budget);
			} while (go);
			if (acc) flush(curw, acc);
			if (cnt >= budget) err |= ERR_RUNAWAY;
			steps = cnt;
		}
	}
	// what is left in the wave's table (all of its lanes are here: the loops above have ended for every one of them)
	if (use_wc)
		for (u32 k = lane; k < UFO_VWC; k += 64u) {
			const u32 key = wc_key[wave][k];
			const u64 bits = wc_mask[wave][k];
			if (key != 0xFFFFFFFFu && bits) __hip_atomic_fetch_or(&M[key], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
	waveAddU64(&ctl->n_steps, steps);
	if (err) atomicOr(&ctl->err, err);
}

// ---- the rays cut into segments of equal work (round 5) -------------------------------------------------------------------
// k_vdda gives a lane one whole ray: 1 500 cells on average in a 2 mm frame, 2 500 the longest, and a wave is as slow as its
// longest ray, a SIMD as slow as the waves it happened to get (measured: 0.30 cells per CU and clock, lanes 43 % busy; the longest
// ray ALONE is a third of the kernel's time). The traversal is a 3-way merge of three independent addition chains (scan_kernels.h,
// "K2' dda, segmented"; k_fcast cuts its rays the same way inside a workgroup), so the state after the k0-th pop of the ray's
// dominant axis is rebuilt exactly with ONE addition per skipped cell instead of one DDA step:
//   k_vcut    one lane per ray: set-up, segments of ~K cells (cuts every w pops of the dominant axis, counted from the SENSOR's
//             end: the m-th segment from the sensor of every ray of a bundle covers the same stretch of space); the three chains
//             run side by side in registers -- the dominant one up to the cut, the two others while their elements precede the
//             cut's (strictly smaller, or equal when the axis wins ties, vector3.h:244-251). The records of a wave's 64 bundled
//             rays go into the segment list m-major -- all rays' segment m, then m - 1, ... -- as arrays per field: every store a
//             run of consecutive words, and 64 consecutive entries are 64 segments that run through the same bricks side by side.
//   k_vwalk   one lane per segment, the reference's step (k_vdda's loop) from the cut to the next cut's cell -- a path never
//             revisits a cell, so "the next segment starts here" is the goal test; marks as in k_vdda
// Same cells, same step count as the sequential walk (tests: ray cells of both forms against each other and against the port).
// (First form of the round: the dominant chain in one kernel, the other axes two lanes per ray in a second one that read and wrote
// 40-byte records scattered over the list -- 0.12 + 0.30 ms of sector-sized accesses; this one 0.4 ms less.)
struct VRay {
	double td[3], dist;
	u32 nseg;     // 0: nothing to walk (a ray inside one cell is marked by k_vcut)
	int8_t s[3];
	uint8_t ax;   // dominant axis
	u32 pad[6];
};
static_assert(sizeof(VRay) == 64, "VRay");
struct VSegs {  // the segment list, one array per field (8 regions of seg_cap entries)
	double *tmx, *tmy, *tmz;  // t_max at the segment's first cell
	u32 *cx, *cy, *cz;        // ... the cell, relative to the grid's corner (VolGeo::cbase)
	u32 *ex, *ey, *ez;        // where the segment ends: the next segment's first cell, the goal for the ray's last segment
	u32* ray;                 // position of the ray in the bundled order; bit 31: the ray's first segment (its first cell is always marked)
};
#define UFO_VSEG_BYTES 52u
#define UFO_VSEG_CNT_STRIDE 32u  // 32-bit words between two regions' segment counters
__host__ inline VSegs volSegViews(void* p, size_t cap)
{
	VSegs v;
	char* c = static_cast<char*>(p);
	v.tmx = reinterpret_cast<double*>(c);
	v.tmy = v.tmx + cap;
	v.tmz = v.tmy + cap;
	v.cx = reinterpret_cast<u32*>(v.tmz + cap);
	v.cy = v.cx + cap;
	v.cz = v.cy + cap;
	v.ex = v.cz + cap;
	v.ey = v.ex + cap;
	v.ez = v.ey + cap;
	v.ray = v.ez + cap;
	return v;
}

__global__ __launch_bounds__(256) void k_vcut(MapGeom g, D3 sensor, Grid gr, VolGeo vg, u64* __restrict__ Mx, u32* __restrict__ tbx, const D3* __restrict__ ray_end,
                                              const ScanCtl* ctl_in, ScanCtl* ctl, const u32* __restrict__ order, u32 K, u32 per, u32 seg_cap, VRay* __restrict__ rays, VSegs sg,
                                              u32* __restrict__ cnt)
{
	const u32 n = ctl_in->n_rays;
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
	const u32 region = min(i / per, 7u);  // (per is a multiple of the block size: uniform)
	const bool live = i < n;
	u32 nseg = 0, w = 1, r0 = 0, ax = 0, err = 0;
	unsigned long long steps = 0;
	RayState r{};
	if (live) {
		raySetup(g, sensor, 0u, gr, ray_end[order ? order[i] : i], r);
		if (3 == r.status) {
			err |= ERR_VOL;  // clipped at the map cube / outside the grid's interior: the checked walk of the general path
		} else if (1 == r.status) {
			const u32 xcc = xccId();
			u32 wd, b;
			volWordBit(vg, (u32)(r.start[0] - vg.cbase[0]), (u32)(r.start[1] - vg.cbase[1]), (u32)(r.start[2] - vg.cbase[2]), &wd, &b);
			__hip_atomic_fetch_or(&Mx[(size_t)xcc * volCopyWords(vg.ntiles) + wd], 1ull << b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			__hip_atomic_fetch_or(&tbx[(size_t)xcc * (size_t)volTbWords(vg.ntiles) + (wd >> 8)], 1u << ((wd >> 3) & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			steps = 1;
		} else if (2 == r.status) {
			const u32 dxn = (u32)abs(r.goal[0] - r.start[0]), dyn = (u32)abs(r.goal[1] - r.start[1]), dzn = (u32)abs(r.goal[2] - r.start[2]);
			ax = (dxn >= dyn && dxn >= dzn) ? 0u : (dyn >= dzn ? 1u : 2u);
			const u32 dmax = ax == 0 ? dxn : (ax == 1 ? dyn : dzn);
			const u32 l1 = dxn + dyn + dzn;
			w = (u32)(((u64)dmax * K) / l1);
			if (w < 1u) w = 1u;
			nseg = (dmax + w - 1u) / w;    // >= 1 (start and goal differ); <= l1 / (K - 3) + 1
			r0 = dmax - (nseg - 1u) * w;   // pops of the ray's FIRST segment (the short one: the cuts are counted from the sensor's end)
		}
	}
	// the wave's run of the region's segment list
	u32 total = nseg, maxn = nseg;
	for (int o = 32; o > 0; o >>= 1) {
		total += __shfl_xor(total, o);
		maxn = max(maxn, (u32)__shfl_xor((int)maxn, o));
	}
	u32 base = 0;
	if (0u == lane && total) base = atomicAdd(&cnt[region * UFO_VSEG_CNT_STRIDE], total);
	base = __shfl(base, 0);
	if (base + total > seg_cap) {  // (cannot happen: the host sizes a region for l1 / (K - 3) + 2 segments per ray)
		err |= ERR_VOL;
		nseg = 0;
		maxn = 0;
	}
	if (live) {
		VRay vr;
		vr.td[0] = r.td[0];
		vr.td[1] = r.td[1];
		vr.td[2] = r.td[2];
		vr.dist = r.dist;
		for (int a = 0; a < 3; ++a) vr.s[a] = r.s[a];
		vr.nseg = nseg;
		vr.ax = (uint8_t)ax;
		for (int a = 0; a < 6; ++a) vr.pad[a] = 0;
		rays[i] = vr;
	}
	// The three chains. a* = the dominant axis; b0, b1 = the two others in axis order (b0 < b1). After k0 pops of a*: element
	// A[k0 - 1] (= v) was popped and t_max_a* = A[k0]; of axis b the elements before v were popped -- strictly smaller, or equal
	// when b wins the tie (b < a*).
	const u32 c0x = (u32)(r.start[0] - vg.cbase[0]), c0y = (u32)(r.start[1] - vg.cbase[1]), c0z = (u32)(r.start[2] - vg.cbase[2]);
	double ta = ax == 0 ? r.tm[0] : (ax == 1 ? r.tm[1] : r.tm[2]), v = ta;
	const double tda = ax == 0 ? r.td[0] : (ax == 1 ? r.td[1] : r.td[2]);
	double t0 = ax == 0 ? r.tm[1] : r.tm[0], t1 = ax == 2 ? r.tm[1] : r.tm[2];
	const double d0 = ax == 0 ? r.td[1] : r.td[0], d1 = ax == 2 ? r.td[1] : r.td[2];
	const bool pri0 = ax != 0u, pri1 = ax == 2u;
	const i32 sa = ax == 0 ? (i32)r.s[0] : (ax == 1 ? (i32)r.s[1] : (i32)r.s[2]);
	const i32 s0 = ax == 0 ? (i32)r.s[1] : (i32)r.s[0], s1 = ax == 2 ? (i32)r.s[1] : (i32)r.s[2];
	u32 k0 = 0, n0 = 0, n1 = 0, guard = 0;
	auto advance = [&](double& tb, const double dbt, const bool pri, u32& cb) {
		for (;;) {  // four candidates per iteration (the same sequence of additions)
			const double q1 = tb + dbt, q2 = q1 + dbt, q3 = q2 + dbt;
			const bool p0 = pri ? (tb <= v) : (tb < v);
			const bool p1 = p0 & (pri ? (q1 <= v) : (q1 < v)), p2 = p1 & (pri ? (q2 <= v) : (q2 < v)), p3 = p2 & (pri ? (q3 <= v) : (q3 < v));
			if (p3) {
				tb = q3 + dbt;
				cb += 4u;
				if (++guard > (1u << 22)) {
					err |= ERR_RUNAWAY;  // (cannot trip: an axis has fewer cells than that)
					break;
				}
				continue;
			}
			tb = p2 ? q3 : (p1 ? q2 : (p0 ? q1 : tb));
			cb += (p0 ? 1u : 0u) + (p1 ? 1u : 0u) + (p2 ? 1u : 0u);
			break;
		}
	};
	const size_t rb = (size_t)region * seg_cap;
	u32 run = 0;
	size_t prev = 0;
	for (int m = (int)maxn - 1; m >= 0; --m) {  // (uniform; the list is m-major: the segments far from the sensor first)
		const bool act = nseg > (u32)m;
		const u64 have = __ballot(act);
		if (act) {
			const u32 j = nseg - 1u - (u32)m;
			if (j > 0u) {
				u32 np = (1u == j) ? r0 : w;
				k0 += np;
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
			}
			const u32 pa = (u32)(sa * (i32)k0), p0 = (u32)(s0 * (i32)n0), p1 = (u32)(s1 * (i32)n1);
			const u32 x = c0x + (ax == 0 ? pa : p0), y = c0y + (ax == 0 ? p0 : (ax == 1 ? pa : p1)), z = c0z + (ax == 2 ? pa : p1);
			const size_t pos = rb + base + run + (u32)__popcll(have & ((1ull << lane) - 1ull));
			sg.tmx[pos] = ax == 0 ? ta : t0;
			sg.tmy[pos] = ax == 0 ? t0 : (ax == 1 ? ta : t1);
			sg.tmz[pos] = ax == 2 ? ta : t1;
			sg.cx[pos] = x;
			sg.cy[pos] = y;
			sg.cz[pos] = z;
			sg.ray[pos] = i | (0u == j ? 0x80000000u : 0u);
			if (j > 0u) {  // the segment before ends where this one starts
				sg.ex[prev] = x;
				sg.ey[prev] = y;
				sg.ez[prev] = z;
			}
			prev = pos;
		}
		run += (u32)__popcll(have);
	}
	if (nseg) {
		sg.ex[prev] = (u32)(r.goal[0] - vg.cbase[0]);
		sg.ey[prev] = (u32)(r.goal[1] - vg.cbase[1]);
		sg.ez[prev] = (u32)(r.goal[2] - vg.cbase[2]);
	}
	waveAddU64(&ctl->n_steps, steps);
	if (err) atomicOr(&ctl->err, err);
}

// (A table keyed by TILE -- an entry holds the tile's eight brick words and goes to the L2 as up to eight atomics and one bit of the
// tile bitmap -- was measured too: 1.78 ms against 1.65 for this one on the 2 mm frame. What the walk costs is instruction issue,
// 62 instructions per step outside the table's code, not what reaches the L2; profiles/r05_ab_experiments.log.)
__global__ __launch_bounds__(256) void k_vwalk(MapGeom g, VolGeo vg, u64* __restrict__ Mx, u32* __restrict__ tbx, const VRay* __restrict__ rays, VSegs sg,
                                               const u32* __restrict__ cnt, u32 seg_cap, const ScanCtl* ctl_in, ScanCtl* ctl, u32 mode, unsigned long long* __restrict__ steps_sh)
{
	// (steps_sh: 64 counters a cache line apart for the waves' step counts -- 5e4 waves adding to ONE word queue at ~12 ns each, and
	// they finish in generations; k_vlist folds them into the control block)
	__shared__ u32 wc_key[4][UFO_VWC];
	__shared__ unsigned long long wc_mask[4][UFO_VWC];
	// A lane that leaves a brick PARKS (brick word, bits) in a small queue of its own -- two plain LDS stores -- and the wave sends
	// everybody's parked entries through the write-combining table together, when some lane's queue is full: the table's protocol
	// (three dependent LDS round trips, a compare-and-swap, the L2 atomics, the tile bitmap's look-at-first) is ~50 instructions that
	// the whole wave used to issue on nearly EVERY step, because in nearly every step some lane of 64 leaves its brick. Now once per
	// UFO_VSTAGE-th brick change of the busiest lane (k_vdda keeps the old form: 1.64 -> ... ms, profiles/r05_ab_experiments.log).
	__shared__ u32 st_w[4][UFO_VSTAGE][64];
	__shared__ unsigned long long st_m[4][UFO_VSTAGE][64];
	const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	const bool use_wc = 0 == (mode & 8u);
	for (u32 k = lane; k < UFO_VWC; k += 64u) {
		wc_key[wave][k] = 0xFFFFFFFFu;
		wc_mask[wave][k] = 0ull;
	}
	if (ctl_in->err) return;  // (a scan the cut kernels flagged -- it takes the general path --: its records are not to be followed)
	// (blocks b, b + 8, ... share an XCD with the usual placement -- speed only: they take the segments of one contiguous eighth
	// of the bundled rays, so that a tile is marked in one or two copies; the copy a wave marks is the one of the XCD it RUNS on)
	const u32 region = blockIdx.x & 7u, G = gridDim.x >> 3;
	const u32 count = min(cnt[region * UFO_VSEG_CNT_STRIDE], seg_cap);
	const u32 xcc = xccId();
	u64* const M = Mx + (size_t)xcc * volCopyWords(vg.ntiles);
	u32* const tb = tbx + (size_t)xcc * (size_t)volTbWords(vg.ntiles);
	auto flush = [&](u32 w, u64 bits) {  // (k_vdda's: the wave's write-combining table)
		if (!use_wc) {
			__hip_atomic_fetch_or(&M[w], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			return;
		}
		const u32 slot = (w * 0x9E3779B1u) >> 24;
		const u32 k = wc_key[wave][slot];
		const bool hit = k == w;
		if (hit) __hip_atomic_fetch_or(&wc_mask[wave][slot], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		u32 prev = k;
		bool won = false;
		if (!hit) {
			prev = atomicCAS(&wc_key[wave][slot], k, w);
			won = prev == k;
		}
		if (won) {
			const u64 old = atomicExch(&wc_mask[wave][slot], (unsigned long long)bits);
			if (k != 0xFFFFFFFFu && old) __hip_atomic_fetch_or(&M[k], old, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		if (!hit && !won) {
			if (prev == w) __hip_atomic_fetch_or(&wc_mask[wave][slot], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			else __hip_atomic_fetch_or(&M[w], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
	};
	// a tile's bit in the XCD's tile bitmap (looked at first: bits only appear during this kernel, the CU's L1 was invalidated when
	// it started, and a stale 0 costs one more atomic)
	auto markTile = [&](u32 tile) {
		if (!((tb[tile >> 5] >> (tile & 31u)) & 1u)) __hip_atomic_fetch_or(&tb[tile >> 5], 1u << (tile & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	};
	u32 nst = 0;  // entries parked in the lane's queue
	// The lanes' parked entries on their way. Called by the lanes that are still walking (the call inside the walk loop sits in
	// divergent code: a lane whose segment has ended is not there) and, behind the loop, by the whole wave: correct either way ONLY
	// because flush and markTile use no cross-lane operation -- a lane sends its own entries; LDS operations of a wave run in program
	// order, which is all the write-combining table's protocol needs. (do not add a shuffle or a ballot to either.)
	auto drain = [&]() {
		u32 last_tile = 0xFFFFFFFFu;
#pragma unroll
		for (u32 k = 0; k < UFO_VSTAGE; ++k) {
			if (k < nst) {
				const u32 w = st_w[wave][k][lane];
				flush(w, st_m[wave][k][lane]);
				if ((w >> 3) != last_tile) markTile(w >> 3);
				last_tile = w >> 3;
			}
		}
		nst = 0;
	};
	auto park = [&](u32 w, u64 bits) {
		// (the queue holds UFO_VSTAGE entries per lane; the ballot behind every step drains a full one before the next park -- enforced,
		// not assumed: an entry that finds the queue full goes out directly)
		if (nst >= UFO_VSTAGE) {
			flush(w, bits);
			markTile(w >> 3);
			return;
		}
		st_w[wave][nst][lane] = w;
		st_m[wave][nst][lane] = bits;
		++nst;
	};
	unsigned long long steps = 0;
	u32 err = 0;
	const u32 nt0 = vg.nt[0], nt1 = vg.nt[1];
	for (u32 qi = (blockIdx.x >> 3) * blockDim.x + threadIdx.x; qi < count; qi += G * blockDim.x) {
		const size_t pos = (size_t)region * seg_cap + qi;
		const u32 rid = sg.ray[pos];
		const VRay vr = rays[rid & 0x7FFFFFFFu];
		const u32 gx = sg.ex[pos], gy = sg.ey[pos], gz = sg.ez[pos];
		u32 x = sg.cx[pos], y = sg.cy[pos], z = sg.cz[pos];
		const u32 sx = (u32)(i32)vr.s[0], sy = (u32)(i32)vr.s[1], sz = (u32)(i32)vr.s[2];
		double tmx = sg.tmx[pos], tmy = sg.tmy[pos], tmz = sg.tmz[pos];
		const double tdx = vr.td[0], tdy = vr.td[1], tdz = vr.td[2];
		const long long idist = __double_as_longlong(vr.dist);
		// the ray's first cell is always marked (the reference's do-while); a later segment starts where the sequential walk has
		// just stepped to: it goes on iff t_max.min() <= distance there (OMB:1300; its cell is not the goal's: fewer pops of the
		// dominant axis)
		bool go = (0 != (rid >> 31)) || ((__double_as_longlong(tmx) <= idist) | (__double_as_longlong(tmy) <= idist) | (__double_as_longlong(tmz) <= idist));
		u32 c = 0, curw = 0xFFFFFFFFu;
		u64 acc = 0;
		while (go) {
			++c;
			const u32 tile = ((z >> 3) * nt1 + (y >> 3)) * nt0 + (x >> 3);
			const u32 w = tile * 8u + (((x >> 2) & 1u) | (((y >> 2) & 1u) << 1) | (((z >> 2) & 1u) << 2));
			const u32 b = (x & 3u) | ((y & 3u) << 2) | ((z & 3u) << 4);
			if (w != curw) {
				if (acc) park(curw, acc);
				curw = w;
				acc = 0;
			}
			acc |= 1ull << b;
			if (__ballot(nst >= UFO_VSTAGE)) drain();  // (the lanes still walking: see drain)
			const bool cxy = tmx <= tmy, cxz = tmx <= tmz, cyz = tmy <= tmz;
			const bool selx = cxy & cxz;
			const bool sely = !cxy & cyz;
			const bool selz = !(selx | sely);
			x += selx ? sx : 0u;
			y += sely ? sy : 0u;
			z += selz ? sz : 0u;
			const double nx = tmx + tdx, ny = tmy + tdy, nz = tmz + tdz;
			tmx = selx ? nx : tmx;
			tmy = sely ? ny : tmy;
			tmz = selz ? nz : tmz;
			const bool more = (__double_as_longlong(tmx) <= idist) | (__double_as_longlong(tmy) <= idist) | (__double_as_longlong(tmz) <= idist);
			go = (((x ^ gx) | (y ^ gy) | (z ^ gz)) != 0u) & more & (c < (1u << 16));
		}
		if (acc) park(curw, acc);  // (room for it: the queue is drained the moment it is full)
		if (c >= (1u << 16)) err |= ERR_RUNAWAY;  // (a segment is ~K cells by construction)
		steps += c;
	}
	drain();  // (all of the wave's lanes are here: the loops above have ended for every one of them)
	if (use_wc)
		for (u32 k = lane; k < UFO_VWC; k += 64u) {
			const u32 key = wc_key[wave][k];
			const u64 bits = wc_mask[wave][k];
			if (key != 0xFFFFFFFFu && bits) __hip_atomic_fetch_or(&M[key], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
	for (int o = 32; o > 0; o >>= 1) steps += __shfl_xor(steps, o);
	if (0 == lane && steps) atomicAdd(&steps_sh[((blockIdx.x * 4u + wave) & 63u) * 16u], steps);
	if (err) atomicOr(&ctl->err, err);
}

// the tiles some XCD has marked -> list (tile, copies it was marked in); the bitmaps are left clean. One thread per word of
// the bitmaps (32 tiles); count in *n_out.
// (... and where each tile's level-3 block was when a walk last left a record for it: k_tile's guess of the tile's group arrives with
// the list entry, its records are asked for together with the brick words -- one dependent round trip less per wave)
__global__ __launch_bounds__(256) void k_vlist(u32* __restrict__ tbx, u32 ntiles, u32* __restrict__ list, uint8_t* __restrict__ copies, u32* n_out, const TileRec* __restrict__ recs,
                                               u32* __restrict__ slots, unsigned long long* __restrict__ steps_sh, ScanCtl* ctl)
{
	if (steps_sh && 0 == blockIdx.x && threadIdx.x < 64u) {  // (the walkers' step counts, k_vwalk)
		unsigned long long v = steps_sh[threadIdx.x * 16u];
		steps_sh[threadIdx.x * 16u] = 0ull;
		for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
		if (0 == threadIdx.x && v) atomicAdd(&ctl->n_steps, v);
	}
	__shared__ u32 wsum[8];
	const u32 nwords = volTbWords(ntiles);  // (the padding words are never marked)
	for (u32 w0 = blockIdx.x * blockDim.x; w0 < nwords; w0 += gridDim.x * blockDim.x) {  // (uniform)
		const u32 w = w0 + threadIdx.x;
		u32 c[8], any = 0;
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			c[k] = w < nwords ? tbx[(size_t)k * nwords + w] : 0u;
			any |= c[k];
		}
#pragma unroll
		for (int k = 0; k < 8; ++k)
			if (c[k]) tbx[(size_t)k * nwords + w] = 0u;
		// (room in the list: ONE atomic per workgroup and pass -- 3 600 waves adding to the one word queued for 43 us of this kernel's 57)
		u32 pos;
		{
			const u32 cnt = (u32)__popc(any), lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
			u32 incl = cnt;
			for (int o = 1; o < 64; o <<= 1) {
				const u32 v = __shfl_up(incl, o);
				if ((int)lane >= o) incl += v;
			}
			if (63u == lane) wsum[wv] = incl;
			__syncthreads();
			if (0 == threadIdx.x) {
				const u32 tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
				wsum[4] = tot ? atomicAdd(n_out, tot) : 0u;
			}
			__syncthreads();
			u32 before = 0;
			for (u32 k = 0; k < wv; ++k) before += wsum[k];
			pos = wsum[4] + before + incl - cnt;
			__syncthreads();  // (wsum is reused by the next pass)
		}
		while (any) {
			const u32 bit = (u32)__ffs(any) - 1u;
			any &= any - 1u;
			u32 cm = 0;
#pragma unroll
			for (int k = 0; k < 8; ++k) cm |= ((c[k] >> bit) & 1u) << k;
			list[pos] = 32u * w + bit;
			copies[pos] = (uint8_t)cm;
			slots[pos] = recs[32u * w + bit].slot;
			++pos;
		}
	}
}

// node blocks the listed tiles' ray cells touch beneath depth 3 (level 1: 2x2x2 cells with a mark, level 2: bricks with a mark,
// level 3: the tile): what a walk into an empty map creates -- the table is sized by it. Eight lanes per tile (lane = brick).
__global__ __launch_bounds__(256) void k_vcount(const u64* __restrict__ Mx, u32 ntiles, const u32* __restrict__ list, const uint8_t* __restrict__ copies, u32 count,
                                                unsigned long long* out)
{
	const u32 i = (blockIdx.x * blockDim.x + threadIdx.x) >> 3, br = threadIdx.x & 7u;
	u32 nblk = 0;
	if (i < count) {
		const u32 tile = list[i];
		u32 cm = copies[i];
		u64 m = 0;
		while (cm) {
			const u32 k = (u32)__ffs(cm) - 1u;
			cm &= cm - 1u;
			m |= Mx[(size_t)k * volCopyWords(ntiles) + (size_t)tile * 8u + br];
		}
		// a 2x2x2 block of the brick holds a mark: fold x pairs, y pairs, z pairs onto the block's first cell
		u64 f = m | (m >> 1);
		f |= f >> 4;
		f |= f >> 16;
		// (cells with even x, y, z: bits 0, 2, 8, 10, and the same + 32)
		nblk = (u32)__popcll(f & 0x0000050500000505ull) + (m ? 1u : 0u) + (0 == br ? 1u : 0u);
	}
	unsigned long long v = nblk;
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	if (0 == (threadIdx.x & 63u) && v) atomicAdd(out, v);
}

// After the table has been exchanged for a larger one in the middle of a walk: the records of the tiles that are done
// carry slots of the old table (k_up links new level-3 blocks through them), and their creations are part of the new
// table's fill already. One thread per listed tile; *n_done counts the tiles that are done.
__global__ __launch_bounds__(256) void k_vfix(Table t, MapGeom g, FastGeo fg, const u32* __restrict__ list, u32 count, TileRec* __restrict__ recs, u32 scan_id, u32* n_done)
{
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	bool done = false;
	if (i < count) {
		const u32 tile = list[i];
		TileRec r = recs[tile];
		if (r.seq == scan_id) {
			done = true;
			u64 lk3;
			if (tileKey(g, fg, tile, &lk3, nullptr)) {
				r.slot = tableFind(t, lk3);  // (NONE: the block had collapsed and was left behind)
				r.counts &= (1u << 25) - 1u;
				recs[tile] = r;
			}
		}
	}
	const u32 c = (u32)__popcll(__ballot(done));
	if (0 == (threadIdx.x & 63u) && c) atomicAdd(n_done, c);
}
// ray cells of the scan as codes (ufomap_map_last_misses), from the merged words k_tile left for the listed tiles: one thread per
// brick word of a listed tile
__global__ __launch_bounds__(256) void k_vcodes(VolGeo vg, const u64* __restrict__ Mm, const u32* __restrict__ list, u32 count, u64* __restrict__ codes, u32 cap,
                                                ScanCtl* ctl)
{
	const u64 nwords = (u64)count * 8u;
	for (u64 w0 = (u64)blockIdx.x * blockDim.x; w0 < nwords; w0 += (u64)gridDim.x * blockDim.x) {  // (uniform: whole waves append)
		const u64 w = w0 + threadIdx.x;
		const u32 tile = w < nwords ? list[w >> 3] : 0u, br = (u32)(w & 7u);
		u64 m = w < nwords ? Mm[(size_t)tile * 8u + br] : 0ull;
		const u32 cnt = (u32)__popcll(m);
		u32 pos = waveAppendN(&ctl->n_codes, cnt);
		if (0 == m) continue;
		const u32 tx = tile % vg.nt[0], tr = tile / vg.nt[0];
		const u32 ty = tr % vg.nt[1], tz = tr / vg.nt[1];
		while (m) {
			const u32 bit = (u32)__ffsll((unsigned long long)m) - 1u;
			m &= m - 1ull;
			const u32 x = (u32)(vg.cbase[0] + (i32)(8u * tx + 4u * (br & 1u) + (bit & 3u)));
			const u32 y = (u32)(vg.cbase[1] + (i32)(8u * ty + 4u * ((br >> 1) & 1u) + ((bit >> 2) & 3u)));
			const u32 z = (u32)(vg.cbase[2] + (i32)(8u * tz + 4u * (br >> 2) + (bit >> 4)));
			if (pos < cap) codes[pos] = morton3(x, y, z);
			++pos;
		}
	}
}
// the reserve's counters -> MapRoot::used has them already through k_up / k_ftail; this only clears them and the walk's flag
__global__ void k_vreset(u32* resv, ScanCtl* ctl, u32 clear_err)
{
	if (threadIdx.x < 64u) resv[threadIdx.x * UFO_RESV_STRIDE] = 0;
	if (0 == threadIdx.x && clear_err) atomicAnd(&ctl->err, ~clear_err);
}
}  // namespace ufo
