// map_kernels.h -- kernels that update the GPU-resident linear-hashed octree from a scan's update
// list, propagate inner-node summaries, and read the map back.
//
// Replaces (reference, ufomap/include/ufo/map/): updateValue (occupancy_map_base.h:1063-1083) with
// Octree::createNode/createChildren (octree.h:997-1058), updateOccupancy (OMB:1139-1145),
// updateAllChildren (OMB:1085-1120), updateParents/updateNode (OMB:1126-1133, 1179-1224),
// isNodeCollapsible/deleteChildren (octree.h:1145-1162, 1060-1086), and for colour maps
// updateValue(code,update,color) / updateNodeColor / updateNode / getAverageChildColor
// (occupancy_map_color.h:269-287, src/map/occupancy_map_color.cpp:115-222).
//
// Batch formulation of the reference's one-update-at-a-time semantics:
//   ensure  : every node block on the path of every touched cell exists (created or revived)
//   init    : new blocks inherit the value of their deepest pre-existing ancestor (octree.h:1044-1054)
//   apply   : all hits (clamp), then all misses (clamp), one thread per 8-child node block
//   propagate: level by level, only where a child's summary changed (OMB:1126-1133 early exit)
#pragma once
#include "expf_ref.h"
#include "scan_kernels.h"

namespace ufo
{
struct Summ {
	float occ;
	u32 fl;  // bit0 contains_free, bit1 contains_unknown
	u32 rgb;
	bool collapsible;
};

__device__ inline u32 levelOf(const MapGeom& g, u64 lk) { return g.L - (u32)((63 - __clzll((long long)lk)) / 3); }

// float exp as the reference's toProb sees it: std::exp(float) (OMB:911 with LogitType=float) -- glibc's expf, bit for bit
// (expf_ref.h; swept against the host's libm over every float32 a clamped log-odds can take: tests/test_toprob_sweep.py)
__device__ inline double toProbF(float logit) { return 1.0 / (1.0 + (double)ufoExpfRef(-logit)); }
// (diagnostics, ufomap_dev_expf: the device's std::exp(float) on an array, in place)
__global__ void k_dev_expf(float* __restrict__ x, u32 n)
{
	for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] = ufoExpfRef(x[i]);
}

// updateNode for a non-leaf node (OMB:1191-1224) + colour average (OMC.cpp:177-222), read-only part.
// oc >= 0 substitutes (o_occ, o_fl, o_rgb) for child oc: the summary the node had before that child's
// last update.
__device__ inline Summ blockSummary(const Table& t, const MapGeom& g, u32 s, u32 level, u32 f, int oc = -1,
                                    float o_occ = 0.f, u32 o_fl = 0, u32 o_rgb = 0)
{
	Summ r;
	const float4* pv = reinterpret_cast<const float4*>(t.occ(s));
	float4 a = pv[0], b = pv[1];
	float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
	if (oc >= 0) {
#pragma unroll
		for (int i = 0; i < 8; ++i)
			if (i == oc) v[i] = o_occ;
		f = (f & ~((1u << oc) | (1u << (8 + oc)))) | ((o_fl & 1u) << oc) | (((o_fl >> 1) & 1u) << (8 + oc));
	}
	float m = v[0];
	bool eq = true;
#pragma unroll
	for (int i = 1; i < 8; ++i) {
		m = fmaxf(m, v[i]);
		eq = eq && (v[i] == v[0]);
	}
	r.occ = m;
	if (1 == level) {
		u32 fl = 0;
#pragma unroll
		for (int i = 0; i < 8; ++i) fl |= (isFreeV(g, v[i]) ? 1u : 0u) | (isUnknownV(g, v[i]) ? 2u : 0u);
		r.fl = fl;
	} else {
		r.fl = ((f & F_CFREE) ? 1u : 0u) | ((f & F_CUNK) ? 2u : 0u);
	}
	r.rgb = 0;
	if (g.color) {
		const u32* pc = t.rgb + 8 * (size_t)s;
		double rr = 0, gg = 0, bb = 0;
		int cnt = 0;
		u32 c0 = pc[0];
		if (0 == oc) c0 = o_rgb;
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			u32 c = (i == oc) ? o_rgb : pc[i];
			eq = eq && (c == c0);
			if (c) {
				double cr = (double)(c & 0xFF), cg = (double)((c >> 8) & 0xFF), cb = (double)((c >> 16) & 0xFF);
				rr += cr * cr;
				gg += cg * cg;
				bb += cb * cb;
				++cnt;
			}
		}
		if (cnt) {
			double num = (double)cnt;
			u32 R = (u32)(uint8_t)sqrt(rr / num), G = (u32)(uint8_t)sqrt(gg / num), B = (u32)(uint8_t)sqrt(bb / num);
			r.rgb = R | (G << 8) | (B << 16);
		}
	}
	r.collapsible = eq && (1 == level || 0 == (f & F_INNER));
	return r;
}

// Write a block's summary into the slot that holds the node's own value. Returns "changed"
// (the bool updateNode returns, OMB:1215-1223 / OMC.cpp:118-121).
// p_known: the caller already holds t.parent(s) (the wave-resident chain of k_propagate_tail prefetches it)
template <bool WG = false>
__device__ inline bool writeToParent(const Table& t, const MapGeom& g, u32 s, u64 lk, const Summ& sm, u32 p_known = NONE)
{
	if (1 == lk) {
		MapRoot* r = t.root;
		bool ch = r->occ != sm.occ || (r->flags & 3u) != sm.fl || (g.color && r->rgb != sm.rgb);
		r->occ = sm.occ;
		r->flags = sm.fl;
		r->rgb = sm.rgb;
		return ch;
	}
	u32 p = (p_known != NONE) ? p_known : t.parent(s);
	u32 ci = (u32)(lk & 7);
	float* po = t.occ(p) + ci;
	u32 fp = aLoad<WG>(&t.flags(p));
	u32 old_fl = ((fp >> ci) & 1u) | (((fp >> (8 + ci)) & 1u) << 1);
	bool ch = (*po != sm.occ) || (old_fl != sm.fl);
	if (g.color) {
		u32* pc = t.rgb + 8 * (size_t)p + ci;
		ch = ch || (*pc != sm.rgb);
		*pc = sm.rgb;
	}
	*po = sm.occ;
	if (old_fl != sm.fl) {
		u32 setm = ((sm.fl & 1u) << ci) | (((sm.fl >> 1) & 1u) << (8 + ci));
		u32 clrm = ((1u << ci) | (1u << (8 + ci))) & ~setm;
		if (setm) aOr<WG>(&t.flags(p), setm);
		if (clrm) aAnd<WG>(&t.flags(p), ~clrm);
	}
	return ch;
}

// What the parent's slot currently holds for this node (= the node's stored value and flags).
__device__ inline Summ readStored(const Table& t, const MapGeom& g, u32 s, u64 lk)
{
	Summ r;
	r.collapsible = false;
	if (1 == lk) {
		r.occ = t.root->occ;
		r.fl = t.root->flags & 3u;
		r.rgb = t.root->rgb;
		return r;
	}
	u32 p = t.parent(s);
	u32 ci = (u32)(lk & 7);
	u32 fp = __hip_atomic_load(&t.flags(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	r.occ = t.occ(p)[ci];
	r.fl = ((fp >> ci) & 1u) | (((fp >> (8 + ci)) & 1u) << 1);
	r.rgb = g.color ? t.rgb[8 * (size_t)p + ci] : 0u;
	return r;
}

// The node became a leaf again (deleteChildren, octree.h:1060-1066): mark the block DEAD and clear
// the parent's "child is inner" bit.
template <bool WG = false>
__device__ inline void collapseBlock(const Table& t, u32 s, u64 lk, u32 p_known = NONE)
{
	aOr<WG>(&t.flags(s), F_DEAD);
	if (1 != lk) aAnd<WG>(&t.flags((p_known != NONE) ? p_known : t.parent(s)), ~(1u << (16 + (u32)(lk & 7))));
}

__device__ inline bool sameSumm(const MapGeom& g, const Summ& a, const Summ& b)
{
	return a.occ == b.occ && a.fl == b.fl && (!g.color || a.rgb == b.rgb);
}

// ---- last-update chain ----------------------------------------------------------------------------
// The reference applies updates one at a time; each one first re-expands its whole path (createNode,
// octree.h:997-1016) and only its own upward pass (updateParents, OMB:1126-1133: stops at the first
// unchanged node) can collapse a node again. Hence after a phase a node N is collapsed iff the LAST
// update beneath N reached N (every node below N on its path changed by that single update) and found
// it collapsible. "Last" is the reference's application order: cloud order for hits (OMB:1351-1354),
// ascending code order for misses (CodeMap iteration within a subtree; the oracle port applies them
// sorted). Each node therefore publishes, per phase: whether its last update reached it and changed its
// summary, and the summary it had just before that update; tmax tells the parent which child carries
// the last update.
#define UFO_TAG(phase) ((u64)((phase)&0xFFFFFFu))
// ONE record per block, in the block's own slot (8 bytes, 12 with colour); the parent pulls the record of the child that
// carries the last update (tmax names it) with one hash lookup. (Round 1 kept eight records per slot in the parent's
// arrays -- no lookup, but 64-96 bytes of every table slot for words that only live during a phase.)
// lu_fl: bits 0-1 flags of the pre-last summary, bit 8 "reached and changed", bits 9.. phase tag.
__device__ inline void publishLast(const Table& t, const MapGeom& g, u32 s, u64 lk, u32 phase, bool reachchg, const Summ& pre,
                                   u32 p_known = NONE)
{
	if (1 == lk) return;
	(void)p_known;
	t.lu_occ[s] = pre.occ;
	if (g.color) t.lu_rgb[s] = pre.rgb;
	t.lu_fl[s] = (pre.fl & 3u) | (reachchg ? 0x100u : 0u) | ((phase & 0x3FFFFFu) << 9);
}
// carry the time of the last update beneath block s up the tree (max-reduction with early exit)
__device__ inline void carryTime(const Table& t, u32 s, u64 lk, u32 phase, u64 time)
{
	u32 b = s;
	while (1 != lk) {
		u32 p = t.parent(b);
		u64 val = (UFO_TAG(phase) << 40) | (time << 3) | (lk & 7);
		u64 old = atomicMax((unsigned long long*)&t.tmax[p], (unsigned long long)val);
		if (old >= val) break;
		b = p;
		lk >>= 3;
	}
}
// For block s (location key lk): did the last update beneath it reach it? If so *pre = summary before it.
__device__ inline bool lastReached(const Table& t, const MapGeom& g, u32 s, u64 lk, u32 level, u32 f, u32 phase, Summ* pre,
                                   const u64* tv_known = nullptr)
{
	u64 tv = tv_known ? *tv_known : t.tmax[s];
	if ((tv >> 40) != UFO_TAG(phase)) return false;
	int c = (int)(tv & 7);
	const u32 cs = tableFindChild(t, s, lk, (u32)c);  // the child's block: updated in this phase, so it is there (maybe just collapsed)
	if (cs == NONE) return false;
	u32 lf = t.lu_fl[cs];
	if ((lf >> 9) != (phase & 0x3FFFFFu) || !(lf & 0x100u)) return false;
	*pre = blockSummary(t, g, s, level, f, c, t.lu_occ[cs], lf & 3u, g.color ? t.lu_rgb[cs] : 0u);
	return true;
}

// Queue block p for the next propagation level. `want` may differ per lane; the append is one atomic
// per wave, so every active lane of the wave must make this call.
template <bool WG = false>
__device__ inline void markDirty(const Table& t, bool want, u32 p, u32* __restrict__ wl, u32* wl_count)
{
	bool first = false;
	if (want) first = !(aOr<WG>(&t.flags(p), F_DIRTY) & F_DIRTY);
	u32 pos = waveAppend<WG>(wl_count, first);
	if (first) wl[pos] = p;
}

// Phase tags start over (ufomap_hip.hip: phaseGuard): no block may look created, reached or timed "in this phase"
// to a phase that reuses an old number. One thread per slot.
__global__ __launch_bounds__(256) void k_reset_tags(Table t)
{
	// (grid-stride: the host caps its launches at 4096 workgroups -- one thread per slot left the tags of a table of more
	// than a million slots in place beyond the first million; round 4's tile-major tables are that large for small maps too)
	for (u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x; s <= (u64)t.mask; s += (u64)gridDim.x * blockDim.x) {
		t.stamp((u32)s) = 0;
		t.tmax[s] = 0;
		t.lu_fl[s] = 0;
	}
}

// a few counters from the host, by value (no host buffer whose lifetime anybody has to think about)
struct SmallCounts {
	u32 v[256];
};
__global__ void k_store_counts(SmallCounts c, u32 n, u32* __restrict__ out)
{
	if (threadIdx.x < n) out[threadIdx.x] = c.v[threadIdx.x];
}

// The colour section of an update list (ufomap_keys_info::reserved bit 1): for each of the first n records, the colours
// of its hit voxels = the colour of the first point in the voxel (what k_apply_leaf finds through the hit hash,
// occupancy_map_color.h:225-233), 0 for the other children.
__global__ __launch_bounds__(256) void k_list_colors(MapGeom g, const Entry* __restrict__ entries, u32 n, HitHash hh, const uint8_t* __restrict__ rgb_in,
                                                     u32* __restrict__ out)
{
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const Entry e = entries[i];
	const u64 pcode = (e.lk ^ (1ULL << (3 * (g.L - 1)))) << 3;  // depth-0 code of child 0
	u32 col[8];
#pragma unroll
	for (int c = 0; c < 8; ++c) {
		col[c] = 0;
		if (!((e.hit >> c) & 1)) continue;
		const u32 hs = hitHashFind(hh, pcode | (u64)c);
		if (hs == NONE) continue;
		const u32 pt = hh.minidx[hs];
		col[c] = (u32)rgb_in[3 * (size_t)pt] | ((u32)rgb_in[3 * (size_t)pt + 1] << 8) | ((u32)rgb_in[3 * (size_t)pt + 2] << 16);
	}
	uint4* o = reinterpret_cast<uint4*>(out + 8 * (size_t)i);
	o[0] = make_uint4(col[0], col[1], col[2], col[3]);
	o[1] = make_uint4(col[4], col[5], col[6], col[7]);
}

// clear flag bits of a control block (the host re-runs an update that had stood back: ERR_PREV only -- a flag the
// scan raised itself must survive)
__global__ void k_ctl_clear(ScanCtl* ctl, u32 bits) { atomicAnd(&ctl->err, ~bits); }

// ------------------------------------------------------------------------------------------------
// S1 ensure: createNode (octree.h:997-1016) for every entry, batched: the thread that creates or
// revives a block also links it to its parent and continues upward; stops at the first block that
// already existed (its ancestors exist by induction).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ensure(Table t, MapGeom g, const Entry* __restrict__ entries,
                                                const u32* n_entries_p, u32 cap_h, u32 cap_m, u32 scan_id,
                                                u32* __restrict__ ent_slot, u32* __restrict__ newlist, u32 newcap,
                                                ScanCtl::PhaseCtr* pc, ScanCtl* ctl, const ScanCtl* prev)
{
	u32 n = *n_entries_p;
	const u32 max_probe = (t.mask >> 1) + 1;
	u32 n_created = 0;
	if (prev && prev->err) {
		// This update was enqueued before its predecessor had been checked by the host (so that the two run back to
		// back). The predecessor flagged itself -- it has left the map alone and will be repeated -- hence this one
		// must not reach the map before it: stand back, the host re-runs both in order.
		if (0 == blockIdx.x && 0 == threadIdx.x) atomicOr(&ctl->err, ERR_PREV);
		return;
	}
	if (ctl->n_entries[0] > cap_h || ctl->n_entries[1] > cap_m) {
		// one of the two update lists did not fit its buffer: NOTHING may be applied, in either phase
		// (both lists are extracted before the first phase starts); the host retries with the exact sizes
		if (0 == blockIdx.x && 0 == threadIdx.x) atomicOr(&ctl->err, ERR_ENTRIES);
		return;
	}
	if (0 == ctl->err) {  // e.g. a runaway ray: leave the map untouched
		for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
			u64 lk = entries[i].lk;
			bool cr;
			u32 s = tableEnsure(t, lk, scan_id, max_probe, &cr, &n_created);
			ent_slot[i] = s;
			if (s == NONE) {
				atomicOr(&ctl->err, ERR_TABLE_FULL);
				continue;
			}
			while (cr) {
				// divergent loop: aggregate the append over the lanes that are in this iteration
				const u64 act = __ballot(1);
				const int leader = __ffsll((unsigned long long)act) - 1;
				u32 base = 0;
				if ((int)__lane_id() == leader) base = atomicAdd(&pc->n_new, (u32)__popcll(act));
				base = __shfl(base, leader);
				u32 pos = base + (u32)__popcll(act & ((1ULL << __lane_id()) - 1ULL));
				if (pos < newcap) newlist[pos] = s;
				else atomicOr(&ctl->err, ERR_TABLE_FULL);
				if (1 == lk) {
					t.parent(s) = NONE;
					break;
				}
				u64 plk = lk >> 3;
				bool pcr;
				u32 ps = tableEnsure(t, plk, scan_id, max_probe, &pcr, &n_created);
				if (ps == NONE) {
					atomicOr(&ctl->err, ERR_TABLE_FULL);
					break;
				}
				t.parent(s) = ps;
				atomicOr(&t.flags(ps), 1u << (16 + (u32)(lk & 7)));
				s = ps;
				lk = plk;
				cr = pcr;
			}
		}
	}
	for (int o = 32; o > 0; o >>= 1) n_created += __shfl_xor(n_created, o);
	if (__lane_id() == 0 && n_created) atomicAdd(&t.root->used, n_created);
}

// How many entries would have to create (or revive) their block: exact count used by the host before it
// decides to grow the table (the a-priori bound assumes every entry is new, which is far off on a warm map).
__global__ __launch_bounds__(256) void k_count_missing(Table t, const Entry* __restrict__ entries, const u32* n_entries_p, u32 cap,
                                                       u32* __restrict__ out)
{
	const u32 n = min(*n_entries_p, cap);
	u32 miss = 0;
	for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		u32 s = tableFind(t, entries[i].lk);
		if (s == NONE || (t.flags(s) & F_DEAD)) ++miss;
	}
	for (int o = 32; o > 0; o >>= 1) miss += __shfl_xor(miss, o);
	if (__lane_id() == 0 && miss) atomicAdd(out, miss);
}

// How many level-2 and level-3 blocks the missing level-1 entries can make new: the distinct missing parents /
// grandparents, counted through bitmaps over the scan's box (level-1 block coordinates b0 .. b0 + nb - 1). The a-priori
// bound takes min(entries, cells of the box) per level, which for a surface scanned at 2 mm is 8x too many at level 2
// (49 M entries, 6 M distinct parents) and makes the table twice as large as it has to be.
struct ParentBox {
	i32 lo2[3], lo3[3];  // first level-2 / level-3 block coordinate of the box
	u32 n2[3], n3[3];
};
__global__ __launch_bounds__(256) void k_mark_parents(Table t, MapGeom g, const Entry* __restrict__ entries, const u32* n_entries_p, u32 cap,
                                                      ParentBox pb, u32* __restrict__ bits2, u32* __restrict__ bits3)
{
	const u32 n = min(*n_entries_p, cap);
	for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const u64 lk = entries[i].lk;
		const u32 s = tableFind(t, lk);
		if (s != NONE && !(t.flags(s) & F_DEAD)) continue;  // the block is there: so are its ancestors
		const u64 p = lk ^ (1ULL << (3 * (g.L - 1)));
		const i32 x = (i32)compact3(p), y = (i32)compact3(p >> 1), z = (i32)compact3(p >> 2);
		const u32 s2 = tableFind(t, lk >> 3);
		if (s2 == NONE || (t.flags(s2) & F_DEAD)) {
			const u32 a = (u32)((x >> 1) - pb.lo2[0]), b = (u32)((y >> 1) - pb.lo2[1]), c = (u32)((z >> 1) - pb.lo2[2]);
			if (a < pb.n2[0] && b < pb.n2[1] && c < pb.n2[2]) {
				const u64 idx = ((u64)c * pb.n2[1] + b) * pb.n2[0] + a;
				atomicOr(&bits2[idx >> 5], 1u << (idx & 31u));
			}
			const u32 s3 = tableFind(t, lk >> 6);
			if (s3 == NONE || (t.flags(s3) & F_DEAD)) {
				const u32 a3 = (u32)((x >> 2) - pb.lo3[0]), b3 = (u32)((y >> 2) - pb.lo3[1]), c3 = (u32)((z >> 2) - pb.lo3[2]);
				if (a3 < pb.n3[0] && b3 < pb.n3[1] && c3 < pb.n3[2]) {
					const u64 idx = ((u64)c3 * pb.n3[1] + b3) * pb.n3[0] + a3;
					atomicOr(&bits3[idx >> 5], 1u << (idx & 31u));
				}
			}
		}
	}
}
__global__ __launch_bounds__(256) void k_popcount(const u32* __restrict__ bits, u64 nwords, unsigned long long* __restrict__ out)
{
	unsigned long long c = 0;
	for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (u64)gridDim.x * blockDim.x) c += (unsigned long long)__popc(bits[w]);
	for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
	if (__lane_id() == 0 && c) atomicAdd(out, c);
}

// S2 init: children of a new block inherit the whole value of the node (createChildren,
// octree.h:1044-1054). The node's value is found in the first ancestor block that is not new.
__global__ __launch_bounds__(256) void k_init_new(Table t, MapGeom g, const u32* __restrict__ newlist, u32 newcap,
                                                  u32 scan_id, const ScanCtl::PhaseCtr* pc, const ScanCtl* ctl)
{
	u32 n = min(pc->n_new, newcap);
	if (ctl->err) return;
	for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		u32 s = newlist[i];
		u32 a = s;
		float v;
		u32 c = 0;
		for (;;) {
			u64 lk = t.key(a);
			if (1 == lk) {
				v = t.root->occ;
				c = t.root->rgb;
				break;
			}
			u32 p = t.parent(a);
			if (t.stamp(p) != scan_id) {
				u32 ci = (u32)(lk & 7);
				v = t.occ(p)[ci];
				if (g.color) c = t.rgb[8 * (size_t)p + ci];
				break;
			}
			a = p;
		}
		float4 vv = make_float4(v, v, v, v);
		float4* po = reinterpret_cast<float4*>(t.occ(s));
		po[0] = vv;
		po[1] = vv;
		if (g.color) {
			uint4 cc = make_uint4(c, c, c, c);
			uint4* pc = reinterpret_cast<uint4*>(t.rgb + 8 * (size_t)s);
			pc[0] = cc;
			pc[1] = cc;
		}
		// leaf children carry the flags of a leaf with this value (OMB:1181-1189)
		u32 masks = (isFreeV(g, v) ? F_CFREE : 0u) | (isUnknownV(g, v) ? F_CUNK : 0u);
		u32 f = t.flags(s);
		t.flags(s) = (f & F_INNER) | masks;
	}
}

// updateNodeColor (OMC.cpp:142-171)
__device__ inline u32 blendColor(const MapGeom& g, u32 cur, u32 upd, float occ_old)
{
	if (cur == upd) return cur;
	if (0 == cur) return upd;
	double prob = g.prob_hit_f;
	double total = prob + toProbF(occ_old);
	prob /= total;
	double inv = 1.0 - prob;
	u32 out = 0;
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		double c = (double)((cur >> (8 * k)) & 0xFF), u = (double)((upd >> (8 * k)) & 0xFF);
		u32 r = (u32)(uint8_t)sqrt(((c * c) * inv) + ((u * u) * prob));
		out |= r << (8 * k);
	}
	return out;
}

// ------------------------------------------------------------------------------------------------
// S3 apply (level-1 blocks): one phase of insertPointCloudHelper (OMB:1351-1365) -- either all hits
// of the scan or all misses -- with updateOccupancy's clamp (OMB:1139-1145), followed by this
// block's own updateNode (OMB:1195-1224) and the hand-off to its parent. One thread per node
// block: the 8 children are one 32-byte record. Hits and misses are separate phases, each followed
// by its own propagation, because the reference joins the hit thread before the first miss lands
// (OMB:1361) and pruning depends on which ancestors were re-evaluated in which phase.
// ------------------------------------------------------------------------------------------------
// `mode`: 0 = the misses of the scan, 1 = its hits (two separate phases, each with its own propagation: used
// for insert depth > 0 and for update lists that arrive from other GPUs), 2 = MERGED: hits, then misses, in
// one pass over a list that holds every touched block once. One pass is enough because after the scan a
// node's value depends on the final values beneath it only, and its collapse state on the LAST update
// beneath it only (last-update chain above) -- and "last" is well defined across the two phases: every miss
// comes after every hit (the reference joins the hit thread before the first miss lands, OMB:1361).
#define UFO_MISS_TIME (1ull << 29)  // later than any hit of the scan (hit time = point index < 2^29)
__global__ __launch_bounds__(256) void k_apply_leaf(Table t, MapGeom g, const Entry* __restrict__ entries,
                                                    const u32* n_entries_p, const u32* __restrict__ ent_slot, float upd_hit,
                                                    float upd_miss, u32 mode, u32 phase, HitHash hh,
                                                    const uint8_t* __restrict__ rgb_in, u32* __restrict__ wl,
                                                    ScanCtl::PhaseCtr* pc, ScanCtl* ctl, ChangeLog cl)
{
	u32 n = *n_entries_p;
	if (ctl->err) return;
	// Parents to queue for the next level are STAGED in LDS and appended to the worklist eight iterations' worth at a time:
	// the worklist counter is one word, an atomic on it costs ~12 ns whoever issues it, and one per wave and iteration was
	// 0.77 M of them = 9 of this kernel's 11 ms on a 49 M-entry list (C3 at insert depth 0).
	__shared__ u32 stage[256][8];
	u32 nstage = 0;
	const u32 stride = gridDim.x * blockDim.x;
	const u32 iters = (n + stride - 1) / stride;  // (uniform trip count: the flush is a wave-wide operation)
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	for (u32 it = 0; it < iters; ++it, i += stride) {
	if (i < n) {
		Entry e = entries[i];
		u32 s = ent_slot[i];
		float4* po = reinterpret_cast<float4*>(t.occ(s));
		float4 a = po[0], b = po[1];
		float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
		const u32 hmask = (0 != mode) ? e.hit : 0u;
		const u32 mmask = (1 != mode) ? e.miss : 0u;
		// the child updated last: for misses the highest code (ascending code order), for hits the latest
		// first-point (cloud order; k_hitmark); a miss is later than any hit
		const bool last_is_miss = 0 != mmask;
		const int c_last = (2 == mode) ? (last_is_miss ? (31 - __clz((int)mmask)) : (int)e.c_last) : (int)e.c_last;
		const u64 t_last = (2 == mode) ? (last_is_miss ? UFO_MISS_TIME : (u64)e.t_last) : (u64)e.t_last;
		const bool blend = hmask && g.color && rgb_in;
		u32 oldc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		if (blend) {
			const u64 pcode = (e.lk ^ (1ULL << (3 * (g.L - 1)))) << 3;  // depth-0 code of child 0
#pragma unroll
			for (int c = 0; c < 8; ++c) {
				if (!((hmask >> c) & 1)) continue;
				u32 hs = hitHashFind(hh, pcode | (u64)c);
				if (hs == NONE) continue;
				u32 pt = hh.minidx[hs];
				// updateValue(code, update, color): colour first, with the OLD occupancy (OMC.h:275-277)
				u32 u = (u32)rgb_in[3 * (size_t)pt] | ((u32)rgb_in[3 * (size_t)pt + 1] << 8) | ((u32)rgb_in[3 * (size_t)pt + 2] << 16);
				u32* pc = t.rgb + 8 * (size_t)s + c;
				oldc[c] = *pc;
				*pc = blendColor(g, oldc[c], u, v[c]);
			}
		}
		u32 old_rgb_last = 0;
#pragma unroll
		for (int c = 0; c < 8; ++c)
			if (c == c_last) old_rgb_last = oldc[c];
		// colour of the last-updated child just before that update: misses leave the colour alone
		if (g.color && (last_is_miss || !blend)) old_rgb_last = t.rgb[8 * (size_t)s + c_last];
		float v_old_last = 0.f;
		u32 chg = 0;  // children whose value a hit or a miss changed: updateOccupancy returned true (OMB:1069-1072)
#pragma unroll
		for (int c = 0; c < 8; ++c) {
			float x = v[c];
			if ((hmask >> c) & 1) {
				if (c == c_last && !last_is_miss) v_old_last = x;
				const float y = clampAdd(x, upd_hit, g.cmin, g.cmax);
				chg |= (y != x) ? (1u << c) : 0u;
				x = y;
			}
			if ((mmask >> c) & 1) {
				if (c == c_last) v_old_last = x;  // last_is_miss
				const float y = clampAdd(x, upd_miss, g.cmin, g.cmax);
				chg |= (y != x) ? (1u << c) : 0u;
				x = y;
			}
			v[c] = x;
		}
		po[0] = make_float4(v[0], v[1], v[2], v[3]);
		po[1] = make_float4(v[4], v[5], v[6], v[7]);
		logChanges(t, cl, (e.lk ^ (1ULL << (3 * (g.L - 1)))) << 3, 0u, chg);
		Summ sm = blockSummary(t, g, s, 1, 0);
		// summary just before the last update of this block (level 1 is always reached: OMB:1128 starts at 1)
		Summ pre = blockSummary(t, g, s, 1, 0, c_last, v_old_last, 0, old_rgb_last);
		const bool reachchg = !sameSumm(g, pre, sm);
		publishLast(t, g, s, e.lk, phase, reachchg, pre);
		carryTime(t, s, e.lk, phase, t_last);
		if (sm.collapsible) collapseBlock(t, s, e.lk);
		// the parent is re-evaluated when the stored summary changed over the whole pass, or when the last
		// update alone changed it (its upward walk reached the parent even if the net change is nil)
		const bool changed = writeToParent(t, g, s, e.lk, sm);
		if (changed || reachchg) {
			const u32 par = t.parent(s);
			if (!(atomicOr(&t.flags(par), F_DIRTY) & F_DIRTY)) stage[threadIdx.x][nstage++] = par;
		}
	}
		if (7u == (it & 7u) || it + 1u == iters) {
			const u32 pos = waveAppendN(&pc->wl_cnt[2], nstage);
			for (u32 k = 0; k < nstage; ++k) wl[pos + k] = stage[threadIdx.x][k];
			nstage = 0;
		}
	}
}

// ------------------------------------------------------------------------------------------------
// S3b batch of scans (update lists of several scans, insert depth 0; ufomap_map_apply_keys_batch): the tree is
// walked ONCE for the whole batch. The lists are applied to the voxels in the reference's order -- scan by
// scan, hits before misses -- by one light launch each (k_apply_values: values only). A block remembers its
// last update (child, the child's value just before, time) in the record arrays of ITS OWN slot, which are
// otherwise unused at level 1 (the children of a level-1 block are voxels and publish nothing); the touched
// blocks are queued once, and k_finish_leaf then does for each what k_apply_leaf does at its end. Time orders
// the whole batch: (scan << 30) | (miss << 29) | point index, so the last-update chain above works unchanged. A list is a scan's
// hits, its misses, or both merged (one record per block with both masks, as k_apply_leaf mode 2).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_apply_values(Table t, MapGeom g, const Entry* __restrict__ entries, const u32* n_entries_p,
                                                      const u32* __restrict__ ent_slot, float upd_hit, float upd_miss, u32 mode, u32 phase,
                                                      u64 time_hi, u32* __restrict__ wl, ScanCtl::PhaseCtr* pc, const ScanCtl* ctl, ChangeLog cl,
                                                      const u32* __restrict__ list_rgb)
{
	// list_rgb (colour maps): 8 colours per record, the colour of the first point of every hit voxel -- what k_apply_leaf
	// looks up through the scan's hit hash travels with the list here (the list may come from another GPU)
	const u32 n = *n_entries_p;
	if (ctl->err) return;
	const u32 stride = gridDim.x * blockDim.x;
	const u32 iters = (n + stride - 1) / stride;  // uniform trip count for the wave-aggregated append
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	for (u32 it = 0; it < iters; ++it, i += stride) {
		bool first = false;
		u32 s = 0;
		if (i < n) {
			const Entry e = entries[i];
			s = ent_slot[i];
			float4* po = reinterpret_cast<float4*>(t.occ(s));
			float4 a = po[0], b = po[1];
			float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
			// mode as in k_apply_leaf: 0 = a list of misses, 1 = a list of hits, 2 = one scan's merged list
			const u32 hmask = (0 != mode) ? e.hit : 0u;
			const u32 mmask = (1 != mode) ? e.miss : 0u;
			const bool last_is_miss = 0 != mmask;
			const int c_last = (2 == mode && last_is_miss) ? (31 - __clz((int)mmask)) : (int)e.c_last;
			const u64 t_last = last_is_miss ? UFO_MISS_TIME : (u64)e.t_last;
			float v_old_last = 0.f;
			u32 chg = 0;
			if (g.color && list_rgb && hmask) {
				// updateValue(code, update, color): colour first, with the OLD occupancy (OMC.h:275-277)
				u32 old_rgb_last = 0;
#pragma unroll
				for (int c = 0; c < 8; ++c) {
					if (!((hmask >> c) & 1)) continue;
					u32* pcol = t.rgb + 8 * (size_t)s + c;
					const u32 oldc = *pcol;
					*pcol = blendColor(g, oldc, list_rgb[8 * (size_t)i + c], v[c]);
					if (c == c_last) old_rgb_last = oldc;
				}
				// colour of the last-updated child just before that update (a miss leaves the colour alone: k_finish_leaf then
				// takes the colour as it is)
				if (!last_is_miss) t.lu_rgb[s] = old_rgb_last;
			}
#pragma unroll
			for (int c = 0; c < 8; ++c) {
				float x = v[c];
				if ((hmask >> c) & 1) {
					if (c == c_last && !last_is_miss) v_old_last = x;
					const float y = clampAdd(x, upd_hit, g.cmin, g.cmax);
					chg |= (y != x) ? (1u << c) : 0u;
					x = y;
				}
				if ((mmask >> c) & 1) {
					if (c == c_last) v_old_last = x;  // last_is_miss
					const float y = clampAdd(x, upd_miss, g.cmin, g.cmax);
					chg |= (y != x) ? (1u << c) : 0u;
					x = y;
				}
				v[c] = x;
			}
			po[0] = make_float4(v[0], v[1], v[2], v[3]);
			po[1] = make_float4(v[4], v[5], v[6], v[7]);
			logChanges(t, cl, (e.lk ^ (1ULL << (3 * (g.L - 1)))) << 3, 0u, chg);
			t.lu_occ[s] = v_old_last;
			t.tmax[s] = (UFO_TAG(phase) << 40) | ((time_hi | t_last) << 3) | (u64)c_last;
			first = !(atomicOr(&t.flags(s), F_DIRTY) & F_DIRTY);
		}
		const u32 pos = waveAppend(&pc->wl_cnt[1], first);
		if (first) wl[pos] = s;
	}
}

__global__ __launch_bounds__(256) void k_finish_leaf(Table t, MapGeom g, const u32* __restrict__ wl_in, u32* __restrict__ wl_out, u32 phase,
                                                     ScanCtl::PhaseCtr* pc, const ScanCtl* ctl)
{
	if (ctl->err) return;
	const u32 n = pc->wl_cnt[1];
	const u32 stride = gridDim.x * blockDim.x;
	const u32 iters = (n + stride - 1) / stride;
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	for (u32 it = 0; it < iters; ++it, i += stride) {
		bool want = false;
		u32 par = NONE;
		if (i < n) {
			const u32 s = wl_in[i];
			atomicAnd(&t.flags(s), ~F_DIRTY);
			const u64 lk = t.key(s);
			const u64 tv = t.tmax[s];
			const int c_last = (int)(tv & 7);
			const u64 time = (tv >> 3) & ((1ull << 37) - 1ull);
			const Summ sm = blockSummary(t, g, s, 1, 0);
			// summary just before the block's last update (level 1 is always reached: OMB:1128 starts at 1)
			// (colour just before the last update: a miss does not touch it; a hit's old colour was parked by k_apply_values)
			const bool last_was_miss = 0 != (time & UFO_MISS_TIME);
			const u32 pre_rgb = g.color ? (last_was_miss ? t.rgb[8 * (size_t)s + c_last] : t.lu_rgb[s]) : 0u;
			const Summ pre = blockSummary(t, g, s, 1, 0, c_last, t.lu_occ[s], 0, pre_rgb);
			const bool reachchg = !sameSumm(g, pre, sm);
			publishLast(t, g, s, lk, phase, reachchg, pre);
			carryTime(t, s, lk, phase, time);
			if (sm.collapsible) collapseBlock(t, s, lk);
			const bool changed = writeToParent(t, g, s, lk, sm);
			want = changed || reachchg;
			par = t.parent(s);
		}
		markDirty(t, want, par, wl_out, &pc->wl_cnt[2]);
	}
}

// ------------------------------------------------------------------------------------------------
// S3c coarse misses (insert depth d > 0): the update list holds level d+1 blocks ("holders") whose
// masked children are depth-d nodes. A child that is a leaf is updated in place (OMB:1067-1072); a child
// that has been expanded gets the update on every leaf below it (updateAllChildren, OMB:1085-1120).
// The reference recurses per node; here the subtrees of ALL entries are walked level-synchronously:
//   k_coarse_begin  per entry: leaf children updated, expanded children pushed onto the visit list
//   k_coarse_down   per level, top-down: leaves of every visited block updated, inner children pushed
//   k_coarse_up     per level, bottom-up: blocks with a changed child re-evaluated (updateNode) and their
//                   summary handed to the parent -- `changed && updateNode(node)` of OMB:1119
//   k_coarse_end    per entry: the holder's own updateNode, reproducing the reference's child-by-child
//                   order (each child is a separate updateValue; the walk up from a LEAF child starts only
//                   on a flag change, OMB:1126-1133 + 1181-1189, so the stored max can be stale -- kept)
// (A per-thread depth-first walk of each subtree took 800 ms on config C3 at depth 6.)
// ------------------------------------------------------------------------------------------------
struct CoarseRec {
	float old_occ[8];  // children values before the phase
	u32 old_flags;     // holder's flags word before the phase
	u32 leaf_trig;     // leaf children whose flags changed (bit per child)
	u32 inner_mask;    // miss children that were expanded
	u32 pad;
};

__global__ __launch_bounds__(256) void k_coarse_begin(Table t, MapGeom g, const Entry* __restrict__ entries, const u32* n_entries_p,
                                                      const u32* __restrict__ ent_slot, float miss, CoarseRec* __restrict__ rec,
                                                      u32* __restrict__ dlist, u32 dcap, ScanCtl* ctl, ChangeLog cl)
{
	u32 n = *n_entries_p;
	if (ctl->err) return;
	// (uniform trip count: the visit list is appended to once per wave and iteration -- a returning atomic on one hot word
	// per expanded child was eight dependent round trips to the memory side per thread)
	for (u32 i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gridDim.x * blockDim.x) {
		const u32 i = i0 + threadIdx.x;
		const bool valid = i < n;
		u32 cs[8];
		u32 npush = 0;
		Entry e{};
		u32 s = NONE, f = 0;
		if (valid) {
			e = entries[i];
			s = ent_slot[i];
			f = t.flags(s);
#pragma unroll
			for (int c = 0; c < 8; ++c) {
				cs[c] = NONE;
				if (((e.miss >> c) & 1) && ((f >> (16 + c)) & 1u)) cs[c] = tableFindChild(t, s, e.lk, (u32)c);  // (the lookups in flight together)
				npush += (cs[c] != NONE) ? 1u : 0u;
			}
		}
		u32 pos = waveAppendN(&ctl->dl_total, npush);
		if (!valid) continue;
		u32 chg = 0;
		CoarseRec r;
		r.old_flags = f;
		r.leaf_trig = 0;
		r.inner_mask = 0;
		r.pad = 0;
		u32 nf = f;
#pragma unroll
		for (int c = 0; c < 8; ++c) {
			float* pv = t.occ(s) + c;
			float v = *pv;
			r.old_occ[c] = v;
			if (!((e.miss >> c) & 1)) continue;
			if ((f >> (16 + c)) & 1u) {
				if (cs[c] != NONE) {
					r.inner_mask |= 1u << c;
					if (pos < dcap) dlist[pos] = cs[c];
					else atomicOr(&ctl->err, ERR_TABLE_FULL);
					++pos;
				}
			} else {
				float nv = clampAdd(v, miss, g.cmin, g.cmax);
				*pv = nv;
				chg |= (nv != v) ? (1u << c) : 0u;  // updateValue on a leaf: the update's own code (OMB:1069-1072)
				u32 nbits = (isFreeV(g, nv) ? (1u << c) : 0u) | (isUnknownV(g, nv) ? (1u << (8 + c)) : 0u);
				u32 obits = f & ((1u << c) | (1u << (8 + c)));
				if (nbits != obits) {
					r.leaf_trig |= 1u << c;
					nf = (nf & ~((1u << c) | (1u << (8 + c)))) | nbits;
				}
			}
		}
		if (nf != f) {
			// only CFREE/CUNK bits of this block change here; nobody else touches this word in this kernel
			t.flags(s) = nf;
		}
		rec[i] = r;
		logChanges(t, cl, (e.lk ^ (1ULL << (3 * (g.L - (u32)e.level)))) << 3, (u32)e.level - 1u, chg);
	}
}

// records where the next level's part of the visit list starts (single thread, between levels)
__global__ void k_coarse_mark(ScanCtl* ctl, u32 level) { ctl->dl_start[level] = ctl->dl_total; }

__global__ __launch_bounds__(256) void k_coarse_down(Table t, MapGeom g, u32 level, float miss, u32* __restrict__ dlist, u32 dcap,
                                                     ScanCtl* ctl, ChangeLog cl, u32 ins_depth)
{
	if (ctl->err) return;
	const u32 lo = ctl->dl_start[level + 1], hi = min(ctl->dl_start[level], dcap);
	for (u32 i0 = lo + blockIdx.x * blockDim.x; i0 < hi; i0 += gridDim.x * blockDim.x) {  // (uniform: one append per wave and iteration, see k_coarse_begin)
		const u32 i = i0 + threadIdx.x;
		const bool valid = i < hi;
		u32 s = NONE, f = 0;
		u64 lk = 0;
		u32 cs[8];
		u32 npush = 0;
		if (valid) {
			s = dlist[i];
			lk = t.key(s);
			f = t.flags(s);
			if (level > 1) {
#pragma unroll
				for (int c = 0; c < 8; ++c) {
					cs[c] = ((f >> (16 + c)) & 1u) ? tableFindChild(t, s, lk, (u32)c) : NONE;  // (the lookups in flight together)
					npush += (cs[c] != NONE) ? 1u : 0u;
				}
			}
		}
		u32 pos = (level > 1) ? waveAppendN(&ctl->dl_total, npush) : 0u;
		if (!valid) continue;
		float4* po = reinterpret_cast<float4*>(t.occ(s));
		float4 a = po[0], b = po[1];
		float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
		u32 nf = f;
		bool changed = false;
#pragma unroll
		for (int c = 0; c < 8; ++c) {
			if (level > 1 && ((f >> (16 + c)) & 1u)) {
				if (cs[c] != NONE) {
					if (pos < dcap) dlist[pos] = cs[c];
					else atomicOr(&ctl->err, ERR_TABLE_FULL);
					++pos;
				}
				continue;
			}
			float nv = clampAdd(v[c], miss, g.cmin, g.cmax);
			if (nv != v[c]) {
				v[c] = nv;
				changed = true;
				if (level > 1) {
					// updateNode on a leaf inner node: flags from its own value (OMB:1181-1189)
					nf &= ~((1u << c) | (1u << (8 + c)));
					nf |= (isFreeV(g, nv) ? (1u << c) : 0u) | (isUnknownV(g, nv) ? (1u << (8 + c)) : 0u);
				}
			}
		}
		if (changed) {
			po[0] = make_float4(v[0], v[1], v[2], v[3]);
			po[1] = make_float4(v[4], v[5], v[6], v[7]);
			t.flags(s) = nf | F_SUB;  // this block's word is private to this thread until the up pass
			// updateAllChildren records the code of the node whose leaf child changed (OMB:1094-1095, 1106-1107). Below the
			// update's own node the reference derives that code with Code::getChild (code.h:257-267), which ADDS the child
			// index to a code whose centre-offset bits (octree.h:321: all three bits of digit depth-1 set) are still in
			// place: the sum carries, and what lands in the set is the node's code + 7 in that digit. Reproduced as is.
			const u64 own = lk ^ (1ULL << (3 * (g.L - level)));
			logChange(t, cl, level < ins_depth ? own + (7ULL << (3 * (ins_depth - 1 - level))) : own, level);
		}
	}
}

__global__ __launch_bounds__(256) void k_coarse_up(Table t, MapGeom g, u32 level, const u32* __restrict__ dlist, u32 dcap,
                                                   ScanCtl* ctl)
{
	if (ctl->err) return;
	const u32 lo = ctl->dl_start[level + 1], hi = min(ctl->dl_start[level], dcap);
	for (u32 i = lo + blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += gridDim.x * blockDim.x) {
		const u32 s = dlist[i];
		u32 f = atomicAnd(&t.flags(s), ~F_SUB);
		if (!(f & F_SUB)) continue;  // nothing changed beneath: the reference does not call updateNode either
		const u64 lk = t.key(s);
		Summ sm = blockSummary(t, g, s, level, f);
		if (sm.collapsible) collapseBlock(t, s, lk);
		if (writeToParent(t, g, s, lk, sm)) atomicOr(&t.flags(t.parent(s)), F_SUB);
	}
}

__global__ __launch_bounds__(256) void k_coarse_end(Table t, MapGeom g, const Entry* __restrict__ entries, const u32* n_entries_p,
                                                    const u32* __restrict__ ent_slot, const CoarseRec* __restrict__ rec, u32 phase,
                                                    u32* __restrict__ wl, ScanCtl::PhaseCtr* pc, ScanCtl* ctl)
{
	u32 n = *n_entries_p;
	if (ctl->err) return;
	for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const Entry e = entries[i];
		const u32 s = ent_slot[i];
		const u32 level = e.level;
		const CoarseRec r = rec[i];
		u32 f = atomicAnd(&t.flags(s), ~F_SUB);  // SUB may have been set by expanded children; not needed here
		f &= ~F_SUB;
		const int c_last = (int)e.c_last;  // highest miss child = the reference's last updateValue on this block
		// which children "triggered" a walk up to this block (see header): leaf -> flags changed;
		// expanded -> its summary (the slot in this block) changed
		u32 trig = r.leaf_trig;
		float cur[8];
		for (int c = 0; c < 8; ++c) {
			cur[c] = t.occ(s)[c];
			if ((r.inner_mask >> c) & 1u) {
				u32 ob = r.old_flags & ((1u << c) | (1u << (8 + c))), nb = f & ((1u << c) | (1u << (8 + c)));
				if (cur[c] != r.old_occ[c] || ob != nb) trig |= 1u << c;
			}
		}
		Summ pre = readStored(t, g, s, e.lk), fin = pre;
		bool last_reached = false;
		if (trig) {
			const int c_t = 31 - __clz((int)trig);
			// the last updateNode of this block ran right after child c_t: later children still had their old state
			float m = 0.f;
			u32 ff = 0, fu = 0;
			for (int c = 0; c < 8; ++c) {
				const bool later = c > c_t && ((e.miss >> c) & 1);
				const float v = later ? r.old_occ[c] : cur[c];
				const u32 wf = later ? r.old_flags : f;
				m = (0 == c) ? v : fmaxf(m, v);
				ff |= (wf >> c) & 1u;
				fu |= (wf >> (8 + c)) & 1u;
			}
			Summ sm = blockSummary(t, g, s, level, f);  // colour average and collapsibility on the current state
			sm.occ = m;
			sm.fl = ff | (fu << 1);
			if (c_t == c_last) {
				// last child triggered: the walk reached this block on its final state
				last_reached = true;
				if (sm.collapsible) collapseBlock(t, s, e.lk);
				// summary just before: child c_last with its old state
				float pm = 0.f;
				u32 pf = 0, pu = 0;
				for (int c = 0; c < 8; ++c) {
					const float v = (c == c_last) ? r.old_occ[c] : cur[c];
					const u32 wf = (c == c_last) ? r.old_flags : f;
					pm = (0 == c) ? v : fmaxf(pm, v);
					pf |= (wf >> c) & 1u;
					pu |= (wf >> (8 + c)) & 1u;
				}
				pre = sm;
				pre.occ = pm;
				pre.fl = pf | (pu << 1);
			}
			fin = sm;
			if (writeToParent(t, g, s, e.lk, sm) && 1 != e.lk) {
				u32 pp = t.parent(s);
				if (!(atomicOr(&t.flags(pp), F_DIRTY) & F_DIRTY)) wl[atomicAdd(&pc->wl_cnt[level + 1], 1u)] = pp;
			}
		}
		publishLast(t, g, s, e.lk, phase, last_reached && !sameSumm(g, pre, fin), pre);
		carryTime(t, s, e.lk, phase, 0);
	}
}

// one block's updateNode (collapse only when the last-update chain reaches it) + hand-off of its summary.
// Returns true when the summary changed, i.e. the parent (*par) has to be re-evaluated.
template <bool WG>
__device__ inline bool propagateCore(const Table& t, const MapGeom& g, u32 s, u32 old, u32 phase, u32* par)
{
	const u64 lk = t.key(s);
	const u32 level = levelOf(g, lk);
	Summ sm = blockSummary(t, g, s, level, old);
	Summ pre = sm;
	const bool reached = lastReached(t, g, s, lk, level, old, phase, &pre);
	if (reached && sm.collapsible) collapseBlock<WG>(t, s, lk);
	const bool changed = writeToParent<WG>(t, g, s, lk, sm);
	const bool reachchg = reached && !sameSumm(g, pre, sm);
	publishLast(t, g, s, lk, phase, reachchg, pre);
	const bool want = (changed || reachchg) && 1 != lk;
	if (want) *par = t.parent(s);
	return want;
}

// worklist flavour; `valid` false lanes only take part in the wave-aggregated append
// the same with the block's key, its tmax word and its parent slot already in registers: every remaining load
// depends on s alone, so one level of the wave-resident chain is ONE round of loads instead of three
template <bool WG>
__device__ inline bool propagateCoreKnown(const Table& t, const MapGeom& g, u32 s, u32 old, u32 phase, u64 lk, u64 tv, u32 ps)
{
	const u32 level = levelOf(g, lk);
	Summ sm = blockSummary(t, g, s, level, old);
	Summ pre = sm;
	const bool reached = lastReached(t, g, s, lk, level, old, phase, &pre, &tv);
	if (reached && sm.collapsible) collapseBlock<WG>(t, s, lk, ps);
	const bool changed = writeToParent<WG>(t, g, s, lk, sm, ps);
	const bool reachchg = reached && !sameSumm(g, pre, sm);
	publishLast(t, g, s, lk, phase, reachchg, pre, ps);
	return (changed || reachchg) && 1 != lk;
}

template <bool WG>
__device__ inline void propagateOne(const Table& t, const MapGeom& g, bool valid, u32 s, u32 phase, u32* __restrict__ wl_out,
                                    u32* cnt_out)
{
	bool want = false;
	u32 par = NONE;
	if (valid) {
		const u32 old = aAnd<WG>(&t.flags(s), ~F_DIRTY);
		want = propagateCore<WG>(t, g, s, old, phase, &par);
	}
	markDirty<WG>(t, want, par, wl_out, cnt_out);
}

__global__ __launch_bounds__(256) void k_propagate(Table t, MapGeom g, const u32* __restrict__ wl_in, u32* __restrict__ wl_out,
                                                   u32 level, u32 phase, ScanCtl::PhaseCtr* pc, const ScanCtl* ctl)
{
	if (ctl->err) return;
	const u32 n = pc->wl_cnt[level];
	const u32 stride = gridDim.x * blockDim.x;
	const u32 iters = (n + stride - 1) / stride;
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	for (u32 it = 0; it < iters; ++it, i += stride) propagateOne<false>(t, g, i < n, i < n ? wl_in[i] : 0u, phase, wl_out, &pc->wl_cnt[level + 1]);
}

// All remaining levels in ONE single-workgroup launch: above the first two or three levels the
// worklists hold a few thousand blocks at most, and 14 dependent launches would cost more than the
// work (MI355X_MICROARCH.md "boundary": ~1.5-2 us each). Only this workgroup touches the queued
// blocks, their parents and the counters during the launch, so all atomics are workgroup-scope (done in
// the XCD's L2 instead of an sc1 round trip to the memory side per hop).
__global__ __launch_bounds__(1024) void k_propagate_tail(Table t, MapGeom g, u32* __restrict__ wl_a, u32* __restrict__ wl_b,
                                                         u32 first_level, u32 phase, ScanCtl::PhaseCtr* pc, ScanCtl* ctl, u32 dbg_at)
{
	if (0 == threadIdx.x) ctl->used_now = t.root->used;  // blocks are only created by k_ensure, long done
	if (threadIdx.x < 64u) {
		u32 ng, nu;
		tableCounts(t, threadIdx.x, &ng, &nu);
		if (0 == threadIdx.x) {
			ctl->used_g_now = ng;
			ctl->used_u_now = nu;
		}
	}
	if (ctl->err) return;
	for (u32 level = first_level; level <= g.L; ++level) {
		if (0 == threadIdx.x) ctl->dbg[dbg_at + level] = wall_clock64();
		u32* in = (level & 1) ? wl_b : wl_a;
		u32* out = (level & 1) ? wl_a : wl_b;
		// level `first_level` was filled by an earlier launch; the later ones by this workgroup
		const u32 n = aLoad<true>(&pc->wl_cnt[level]);
		if (0 == n) break;  // uniform
		if (n <= 64u) {
			// Narrow from here to the root (each level has at most as many dirty blocks as the one below): one
			// wave carries the items in registers -- no worklist, no DIRTY bits, no counters, no block barrier.
			// Parents are de-duplicated with ballots; a workgroup-scope fence orders a level's stores before
			// the next level's loads (same wave, same CU).
			if (threadIdx.x >= 64u) return;
			const u32 lane = threadIdx.x;
			u32 s = lane < n ? in[lane] : NONE;
			// key, tmax word and parent slot of the lane's block travel with it; those of the NEXT block (the parent)
			// are loaded while this one is evaluated -- they do not change during propagation (tmax is final after the
			// apply kernels, the key of a parent is the child's key >> 3) -- so a level costs one round of loads
			u64 lk = 1, tv = 0;
			u32 ps = NONE;
			if (s != NONE) {
				lk = t.key(s);
				tv = t.tmax[s];
				ps = (1 != lk) ? t.parent(s) : NONE;
			}
			bool first = true;
			for (u32 guard = 0; guard < 32u; ++guard) {
				const bool valid = s != NONE;
				if (0 == __ballot(valid)) break;
				bool want = false;
				u64 tv_n = 0;
				u32 pp_n = NONE;
				if (valid) {
					if (1 != lk) {
						tv_n = t.tmax[ps];
						pp_n = (1 != (lk >> 3)) ? t.parent(ps) : NONE;
					}
					// items of the first round came off the worklist (DIRTY set); later rounds were never queued
					const u32 old = first ? aAnd<true>(&t.flags(s), ~F_DIRTY) : aLoad<true>(&t.flags(s));
					want = propagateCoreKnown<true>(t, g, s, old, phase, lk, tv, ps);
				}
				first = false;
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				u32 next = NONE;
				u64 todo = __ballot(want);
				while (todo) {
					const int ld = __ffsll((unsigned long long)todo) - 1;
					const u32 pl = __shfl(ps, ld);
					const u64 same = __ballot(want && ps == pl);
					if ((int)lane == ld) next = pl;
					todo &= ~same;
				}
				// the leader of a parent continues with what it prefetched for it (every child holds the same values)
				s = next;
				lk >>= 3;
				tv = tv_n;
				ps = pp_n;
			}
			if (0 == lane) ctl->dbg[dbg_at + 31] = wall_clock64();
			return;
		}
		const u32 iters = (n + blockDim.x - 1) / blockDim.x;
		u32 i = threadIdx.x;
		for (u32 it = 0; it < iters; ++it, i += blockDim.x) {
			u32 s = 0;
			if (i < n) s = in[i];
			propagateOne<true>(t, g, i < n, s, phase, out, &pc->wl_cnt[level + 1]);
		}
		__syncthreads();  // workgroup-scope release/acquire: the next level reads what this one wrote
	}
	if (0 == threadIdx.x) ctl->dbg[dbg_at + 31] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------
// Robot clearing: OccupancyMapBase::setValueVolume(AABB, value, min_depth)
// (occupancy_map_base.h:492-518) with setValueVolumeRecurs (986-1031) as the server calls it after every scan
// (ufomap_mapping/src/server.cpp:152-155). The reference recurses from the root through every node whose box
// intersects the volume (createChildren on each: a leaf on the way is expanded by inheritance), sets the
// intersecting children at min_depth (or the voxels), and on the way back calls updateNode on a node only if
// something beneath it changed -- returning "true" also when NOTHING changed (`return !changed || updateNode`),
// which the caller counts as a change. Here the recursion is level-synchronous: one launch per level down
// (k_vol_down: a record per visited node, its centre carried along because the reference adds +-half sizes
// level by level, octree.h:625-633), a breadth-first kill of the subtrees that deleteChildren removes
// (k_vol_kill), one launch per level up (k_vol_up).
// ------------------------------------------------------------------------------------------------
struct VolRec {
	u64 lk;       // key of the visited node's children block
	double c[3];  // centre of the node
	u32 parent;   // record of the parent node (NONE for the root)
	u32 slot;     // table slot of the children block
	u32 changed;  // "changed" of the reference's loop: own children, plus what the recursion into them returned
	u32 pad;
};
struct VolArgs {
	double vc[3], vh[3];  // the volume: AABB(min, max) = centre, half size (geometry/aabb.h:62-65)
	float val;            // clamp(float(toLogit(occupancy_value))) (occupancy_map_base.h:1151-1157)
	u32 min_depth;
};
// geometry::intersects(AABB, AABB) (collision_checks.cpp:256-264) on getMin/getMax (aabb.h:67-69)
__device__ inline bool volIntersects(const VolArgs& a, const double c[3], double h)
{
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		const double min1 = a.vc[k] - a.vh[k], max1 = a.vc[k] + a.vh[k], min2 = c[k] - h, max2 = c[k] + h;
		if (!(min1 <= max2) || !(min2 <= max1)) return false;
	}
	return true;
}

__global__ void k_vol_begin(VolRec* __restrict__ rec, ScanCtl* ctl, u32 L)
{
	VolRec r;
	r.lk = 1;
	r.c[0] = r.c[1] = r.c[2] = 0.0;
	r.parent = NONE;
	r.slot = NONE;
	r.changed = 0;
	r.pad = 0;
	rec[0] = r;
	ctl->dl_start[L] = 0;
	ctl->dl_total = 1;
	ctl->dl_start[L - 1] = 1;
}

// One level of the descent: the records of depth `cd` nodes are [dl_start[cd], dl_start[cd-1]).
// kill: blocks whose subtree deleteChildren removes (children at min_depth that had been expanded).
// (lo, hi: the level's records; the children's records are reserved at next_base + *next_cnt -- k_vol_down: the running total in
// the control block, k_vol_all: a counter per level in LDS)
__device__ inline void volDownLevel(const Table& t, const MapGeom& g, const VolArgs& a, u32 cd, VolRec* __restrict__ rec, u32 rcap, u32* __restrict__ kill,
                                    u32 kcap, u32 scan_id, ScanCtl* ctl, u32 lo, u32 hi, u32* next_cnt, u32 next_base)
{
	const u32 max_probe = (t.mask >> 1) + 1;
	const u32 child_depth = cd - 1;
	const double chs = g.hs[child_depth];
	u32 n_created = 0;
	for (u32 r = lo + blockIdx.x * blockDim.x + threadIdx.x; r < hi; r += gridDim.x * blockDim.x) {
		VolRec me = rec[r];
		// createChildren (octree.h:1022-1058): a leaf node gets its 8 children, each a copy of the node
		bool created;
		const u32 s = tableEnsure(t, me.lk, scan_id, max_probe, &created, &n_created);
		if (s == NONE) {
			atomicOr(&ctl->err, ERR_TABLE_FULL);
			continue;
		}
		rec[r].slot = s;
		if (created) {
			float v;
			u32 col = 0;
			if (1 == me.lk) {
				t.parent(s) = NONE;
				v = t.root->occ;
				col = t.root->rgb;
			} else {
				const u32 p = rec[me.parent].slot, ci = (u32)(me.lk & 7);
				t.parent(s) = p;
				atomicOr(&t.flags(p), 1u << (16 + ci));
				v = t.occ(p)[ci];
				if (g.color) col = t.rgb[8 * (size_t)p + ci];
			}
			float4 vv = make_float4(v, v, v, v);
			float4* po = reinterpret_cast<float4*>(t.occ(s));
			po[0] = vv;
			po[1] = vv;
			if (g.color) {
				uint4 cc = make_uint4(col, col, col, col);
				uint4* pc = reinterpret_cast<uint4*>(t.rgb + 8 * (size_t)s);
				pc[0] = cc;
				pc[1] = cc;
			}
			// leaf children carry the flags of a leaf with this value (as k_init_new; OMB:1181-1189)
			t.flags(s) = (isFreeV(g, v) ? F_CFREE : 0u) | (isUnknownV(g, v) ? F_CUNK : 0u);
		}
		u32 changed = 0;
		// the children the volume reaches; those that are descended into get their records with ONE reservation (a returning
		// atomic per child was up to eight dependent round trips to the memory side per node and level)
		u32 inter = 0;
#pragma unroll
		for (u32 i = 0; i < 8; ++i) {
			const double ci[3] = {me.c[0] + ((i & 1) ? chs : -chs), me.c[1] + ((i & 2) ? chs : -chs), me.c[2] + ((i & 4) ? chs : -chs)};
			inter |= volIntersects(a, ci, chs) ? (1u << i) : 0u;
		}
		const bool descend = 0 != child_depth && a.min_depth < child_depth;
		u32 next = (descend && inter) ? next_base + atomicAdd(next_cnt, (u32)__popc(inter)) : 0u;
		for (u32 i = 0; i < 8; ++i) {
			double cc[3] = {me.c[0], me.c[1], me.c[2]};  // getChildCenter (octree.h:625-633)
			cc[0] += ((i & 1) ? chs : -chs);
			cc[1] += ((i & 2) ? chs : -chs);
			cc[2] += ((i & 4) ? chs : -chs);
			if (!((inter >> i) & 1u)) continue;
			float* pv = t.occ(s) + i;
			if (0 == child_depth) {
				if (*pv != a.val) changed = 1;  // setOccupancy (OMB:1151-1157)
				*pv = a.val;
			} else if (a.min_depth < child_depth) {
				const u32 pos = next++;
				if (pos < rcap) {
					VolRec ch;
					ch.lk = (me.lk << 3) | (u64)i;
					ch.c[0] = cc[0];
					ch.c[1] = cc[1];
					ch.c[2] = cc[2];
					ch.parent = r;
					ch.slot = NONE;
					ch.changed = 0;
					ch.pad = 0;
					rec[pos] = ch;
				} else {
					atomicOr(&ctl->err, ERR_ENTRIES);
				}
			} else {
				// deleteChildren(child) + setOccupancy(child) + updateNode(child), the child now a leaf (OMB:1019-1026)
				const u32 f = __hip_atomic_load(&t.flags(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (f & (1u << (16 + i))) {
					const u32 cs = tableFind(t, (me.lk << 3) | (u64)i);
					if (cs != NONE) {
						const u32 kp = atomicAdd(&ctl->n_codes, 1u);
						if (kp < kcap) kill[kp] = cs;
						else atomicOr(&ctl->err, ERR_ENTRIES);
					}
					atomicAnd(&t.flags(s), ~(1u << (16 + i)));
				}
				if (*pv != a.val) changed = 1;
				*pv = a.val;
				const u32 nf = isFreeV(g, a.val) ? 1u : 0u, nu = isUnknownV(g, a.val) ? 1u : 0u;
				if (((f >> i) & 1u) != nf || ((f >> (8 + i)) & 1u) != nu) changed = 1;  // updateNode, leaf branch (OMB:1181-1189)
				atomicAnd(&t.flags(s), ~((1u << i) | (1u << (8 + i))));
				if (nf | nu) atomicOr(&t.flags(s), (nf << i) | (nu << (8 + i)));
			}
		}
		rec[r].changed = changed;
	}
	for (int o = 32; o > 0; o >>= 1) n_created += __shfl_xor(n_created, o);
	if (__lane_id() == 0 && n_created) atomicAdd(&t.root->used, n_created);
}

__global__ __launch_bounds__(256) void k_vol_down(Table t, MapGeom g, VolArgs a, u32 cd, VolRec* __restrict__ rec, u32 rcap,
                                                  u32* __restrict__ kill, u32 kcap, u32 scan_id, ScanCtl* ctl)
{
	volDownLevel(t, g, a, cd, rec, rcap, kill, kcap, scan_id, ctl, ctl->dl_start[cd], min(ctl->dl_start[cd - 1], rcap), &ctl->dl_total, 0u);
}

// Breadth-first removal of subtrees: blocks kill[lo, hi) die, their live child blocks are appended.
// The range bounds live in ctl->dbg[56] (lo) / ctl->n_codes (end of the list); k_vol_kill_mark advances lo.
__global__ void k_vol_kill_mark(ScanCtl* ctl, u32 which)
{
	if (0 == which) {
		ctl->dbg[56] = 0;
		ctl->dbg[57] = ctl->n_codes;
	} else {
		ctl->dbg[56] = ctl->dbg[57];
		ctl->dbg[57] = ctl->n_codes;
	}
}
__global__ __launch_bounds__(256) void k_vol_kill(Table t, u32* __restrict__ kill, u32 kcap, ScanCtl* ctl)
{
	const u32 lo = (u32)ctl->dbg[56], hi = min((u32)ctl->dbg[57], kcap);
	for (u32 k = lo + blockIdx.x * blockDim.x + threadIdx.x; k < hi; k += gridDim.x * blockDim.x) {
		const u32 s = kill[k];
		const u32 f = t.flags(s);
		const u64 lk = t.key(s);
		for (u32 i = 0; i < 8; ++i) {
			if (!(f & (1u << (16 + i)))) continue;
			const u32 cs = tableFindChild(t, s, lk, i);
			if (cs == NONE) continue;
			const u32 kp = atomicAdd(&ctl->n_codes, 1u);
			if (kp < kcap) kill[kp] = cs;
			else atomicOr(&ctl->err, ERR_ENTRIES);
		}
		t.flags(s) = (f & ~(F_INNER | F_DIRTY | F_SUB)) | F_DEAD;  // a revived block starts without inner children
	}
}

// One level of the way back: `return !changed || updateNode(node, depth)` (OMB:1030) for the depth-`cd` records.
__device__ inline void volUpLevel(const Table& t, const MapGeom& g, u32 cd, VolRec* __restrict__ rec, const ScanCtl* ctl, u32 lo, u32 hi, u32* marked = nullptr)
{
	if (ctl->err) return;
	for (u32 r = lo + blockIdx.x * blockDim.x + threadIdx.x; r < hi; r += gridDim.x * blockDim.x) {
		const VolRec me = rec[r];
		bool ret = true;
		if (me.changed) {
			// updateNode of a node with children (OMB:1191-1224): summary, collapse, compare
			const u32 f = t.flags(me.slot);
			const Summ sm = blockSummary(t, g, me.slot, cd, f);
			if (sm.collapsible) collapseBlock(t, me.slot, me.lk);
			ret = writeToParent(t, g, me.slot, me.lk, sm);
		}
		if (ret && me.parent != NONE) {
			atomicOr(&rec[me.parent].changed, 1u);
			if (marked) *marked = 1u;  // (k_vol_all, LDS: the level above has something to do)
		}
	}
}

__global__ __launch_bounds__(256) void k_vol_up(Table t, MapGeom g, u32 cd, VolRec* __restrict__ rec, u32 rcap, const ScanCtl* ctl)
{
	volUpLevel(t, g, cd, rec, ctl, ctl->dl_start[cd], min(ctl->dl_start[cd - 1], rcap));
}
// A volume of a few thousand nodes at most (the robot's own box, cleared after every scan: server.cpp:122-160) at
// min_depth 0: the whole descent and the way back by ONE workgroup, a barrier per level -- instead of three launches per
// level (48 for 16 levels, ~4 us each and nothing to do in most of them).
__global__ __launch_bounds__(1024) void k_vol_all(Table t, MapGeom g, VolArgs a, u32 L, VolRec* __restrict__ rec, u32 rcap, u32* __restrict__ kill, u32 kcap,
                                                  u32 scan_id, ScanCtl* ctl, ScanCtl* host_result, unsigned long long done_value, u32 init_ctl)
{
	auto levelSync = [] {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__syncthreads();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	};
	// (round 6: where a level's records start and how many children it has reserved live in LDS -- a returning atomic on the control
	// block and a second barrier per level, to publish the next level's start, cost the descent a trip to memory per level)
	__shared__ u32 lstart[26], lcnt[26], lmark[26];  // records of depth cd: [lstart[cd], lstart[cd - 1]); lcnt[cd]: reserved for depth cd
	if (threadIdx.x < 26u) lcnt[threadIdx.x] = lmark[threadIdx.x] = 0;
	if (init_ctl) {
		// (the control block's start state: zero but for the box of changes -- written here instead of uploaded by the host)
		u32* cw = reinterpret_cast<u32*>(ctl);
		for (u32 w = threadIdx.x; w < (u32)(sizeof(ScanCtl) / 4u); w += blockDim.x) cw[w] = 0u;
		levelSync();
		if (threadIdx.x < 3u) ctl->aabb_min[threadIdx.x] = ~0ull;
	}
	if (0 == threadIdx.x) {
		VolRec r;
		r.lk = 1;
		r.c[0] = r.c[1] = r.c[2] = 0.0;
		r.parent = NONE;
		r.slot = NONE;
		r.changed = 0;
		r.pad = 0;
		rec[0] = r;
	}
	levelSync();
	u32 lo = 0, hi = 1;  // (uniform: the level's records)
	for (u32 cd = L; cd > a.min_depth; --cd) {
		volDownLevel(t, g, a, cd, rec, rcap, kill, kcap, scan_id, ctl, lo, min(hi, rcap), &lcnt[cd - 1], hi);
		levelSync();
		if (0 == threadIdx.x) lstart[cd] = lo;  // (for the way back)
		const u32 n = lcnt[cd - 1];             // (complete: nobody adds to it after this level)
		lo = hi;
		hi += n;
	}
	if (0 == threadIdx.x) lstart[a.min_depth] = lo;
	levelSync();
	for (u32 cd = a.min_depth + 1; cd <= L; ++cd) {
		volUpLevel(t, g, cd, rec, ctl, lstart[cd], min(lstart[cd - 1], rcap), &lmark[cd + 1]);
		levelSync();
		// (a record is only ever marked from the level below: when none was, nothing is left to do on the way to the root)
		if (0 == lmark[cd + 1]) break;
	}
	// Round 6: the finished control block -- with the table's fill, which the host sizes the next update by -- goes to the host's
	// pinned copy from here, followed by the word the host polls (as k_ftail does for a scan): the call needed a stream
	// synchronisation, a read-back copy, a counting kernel and two more copies with a second synchronisation for it (60 us).
	if (!host_result) return;
	if (threadIdx.x < 64u) {
		u32 ng, nu;
		tableCounts(t, threadIdx.x, &ng, &nu);
		if (0 == threadIdx.x) {
			ctl->used_now = __hip_atomic_load(&t.root->used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			ctl->used_g_now = ng;
			ctl->used_u_now = nu;
		}
	}
	levelSync();
	{
		constexpr u32 W_CORE = offsetof(ScanCtl, dbg) / 4u;
		const u32* dev = reinterpret_cast<const u32*>(ctl);
		u32* host = reinterpret_cast<u32*>(host_result);
		for (u32 w = threadIdx.x; w < W_CORE; w += blockDim.x) host[w] = __hip_atomic_load(&dev[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	__syncthreads();  // (every thread's stores to the pinned block have been acknowledged: the barrier waits for them)
	if (0 == threadIdx.x) __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_result + 1), done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The same for a volume of at most UFO_VOL_SMALL nodes (the robot's box: ~400), round 6 -- k_vol_all's descent costs a trip to memory or two
// per level whatever the level holds (the node's record, written a level earlier; the hash probe for its block): 47 us for 16 levels.
// WHICH nodes are visited is geometry alone (every child whose box meets the volume, OMB:507-518 -- a leaf on the way gets children), so:
//   G  the records of all levels, level by level, in LDS (the box tests of one level, an LDS reservation, a barrier);
//   T  every record's block found or created -- all levels at once, one round of probes;
//   C  (only if a block was created, i.e. hardly ever after the first scans) the new blocks' contents, top-down: copies of the parent node;
//   V  the values of the depth-0 nodes, all at once;
//   U  updateNode bottom-up, stopping where no parent was marked (k_vol_all's way back).
#define UFO_VOL_SMALL 1024u
__global__ __launch_bounds__(1024) void k_vol_small(Table t, MapGeom g, VolArgs a, u32 L, u32 scan_id, ScanCtl* ctl, ScanCtl* host_result, unsigned long long done_value)
{
	__shared__ VolRec recs[UFO_VOL_SMALL];
	__shared__ u32 lstart[26], lcnt[26], lmark[26], any_created;
	auto levelSync = [] {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__syncthreads();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	};
	if (threadIdx.x < 26u) lcnt[threadIdx.x] = lmark[threadIdx.x] = 0;
	{
		// (the control block's start state: zero but for the box of changes)
		u32* cw = reinterpret_cast<u32*>(ctl);
		for (u32 w = threadIdx.x; w < (u32)(sizeof(ScanCtl) / 4u); w += blockDim.x) cw[w] = 0u;
	}
	if (0 == threadIdx.x) {
		any_created = 0;
		VolRec r;
		r.lk = 1;
		r.c[0] = r.c[1] = r.c[2] = 0.0;
		r.parent = NONE;
		r.slot = NONE;
		r.changed = 0;
		r.pad = 0;
		recs[0] = r;
	}
	levelSync();
	if (threadIdx.x < 3u) ctl->aabb_min[threadIdx.x] = ~0ull;
	// ---- G ----
	u32 lo = 0, hi = 1, overflow = 0;
	for (u32 cd = L; cd >= 1u; --cd) {
		const double chs = g.hs[cd - 1u];
		for (u32 r = lo + threadIdx.x; r < hi; r += blockDim.x) {
			const VolRec me = recs[r];
			u32 inter = 0;
#pragma unroll
			for (u32 i = 0; i < 8; ++i) {
				const double ci[3] = {me.c[0] + ((i & 1) ? chs : -chs), me.c[1] + ((i & 2) ? chs : -chs), me.c[2] + ((i & 4) ? chs : -chs)};
				inter |= volIntersects(a, ci, chs) ? (1u << i) : 0u;
			}
			recs[r].pad = inter;
			if (cd >= 2u && inter) {
				u32 next = hi + atomicAdd(&lcnt[cd - 1u], (u32)__popc(inter));
				for (u32 i = 0; i < 8; ++i) {
					if (!((inter >> i) & 1u)) continue;
					const u32 pos = next++;
					if (pos >= UFO_VOL_SMALL) {
						overflow = 1;  // (cannot happen: the host's bound on the records is at most the array)
						continue;
					}
					VolRec ch;
					ch.lk = (me.lk << 3) | (u64)i;
					ch.c[0] = me.c[0] + ((i & 1) ? chs : -chs);  // getChildCenter (octree.h:625-633)
					ch.c[1] = me.c[1] + ((i & 2) ? chs : -chs);
					ch.c[2] = me.c[2] + ((i & 4) ? chs : -chs);
					ch.parent = r;
					ch.slot = NONE;
					ch.changed = 0;
					ch.pad = 0;
					recs[pos] = ch;
				}
			}
		}
		__syncthreads();
		if (0 == threadIdx.x) lstart[cd] = lo;
		const u32 n = lcnt[cd - 1u];
		lo = hi;
		hi = min(hi + n, (u32)UFO_VOL_SMALL);
	}
	if (0 == threadIdx.x) lstart[0] = lo;
	if (overflow) atomicOr(&ctl->err, ERR_ENTRIES);
	const u32 n_rec = lo;  // (uniform: level 1 is the last one with records)
	// ---- T ----
	{
		const u32 max_probe = (t.mask >> 1) + 1;
		u32 n_created = 0;
		for (u32 r = threadIdx.x; r < n_rec; r += blockDim.x) {
			bool created;
			const u32 s = tableEnsure(t, recs[r].lk, scan_id, max_probe, &created, &n_created);
			if (s == NONE) atomicOr(&ctl->err, ERR_TABLE_FULL);
			recs[r].slot = s;
			if (created) {
				recs[r].pad |= 0x100u;
				any_created = 1u;
			}
		}
		for (int o = 32; o > 0; o >>= 1) n_created += __shfl_xor(n_created, o);
		if (__lane_id() == 0 && n_created) atomicAdd(&t.root->used, n_created);
	}
	levelSync();
	if (__hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
		// (nothing but new, unlinked blocks has been written: the host reports the table as full, as it does for k_vol_all)
	} else {
		// ---- C ----
		if (any_created) {
			for (u32 cd = L; cd >= 1u; --cd) {
				for (u32 r = lstart[cd] + threadIdx.x; r < lstart[cd - 1u]; r += blockDim.x) {
					const VolRec me = recs[r];
					if (!(me.pad & 0x100u)) continue;
					// createChildren (octree.h:1022-1058): a leaf node gets its 8 children, each a copy of the node
					const u32 s = me.slot;
					float v;
					u32 col = 0;
					if (1 == me.lk) {
						t.parent(s) = NONE;
						v = t.root->occ;
						col = t.root->rgb;
					} else {
						const u32 p = recs[me.parent].slot, ci = (u32)(me.lk & 7);
						t.parent(s) = p;
						atomicOr(&t.flags(p), 1u << (16 + ci));
						v = t.occ(p)[ci];
						if (g.color) col = t.rgb[8 * (size_t)p + ci];
					}
					const float4 vv = make_float4(v, v, v, v);
					float4* po = reinterpret_cast<float4*>(t.occ(s));
					po[0] = vv;
					po[1] = vv;
					if (g.color) {
						const uint4 cc = make_uint4(col, col, col, col);
						uint4* pc = reinterpret_cast<uint4*>(t.rgb + 8 * (size_t)s);
						pc[0] = cc;
						pc[1] = cc;
					}
					// leaf children carry the flags of a leaf with this value (as k_init_new; OMB:1181-1189)
					t.flags(s) = (isFreeV(g, v) ? F_CFREE : 0u) | (isUnknownV(g, v) ? F_CUNK : 0u);
				}
				levelSync();
			}
		}
		// ---- V ---- setOccupancy on the depth-0 nodes the volume reaches (OMB:1151-1157)
		for (u32 r = lstart[1] + threadIdx.x; r < lstart[0]; r += blockDim.x) {
			const u32 s = recs[r].slot, inter = recs[r].pad & 0xFFu;
			u32 changed = 0;
			for (u32 i = 0; i < 8; ++i) {
				if (!((inter >> i) & 1u)) continue;
				float* pv = t.occ(s) + i;
				if (*pv != a.val) changed = 1;
				*pv = a.val;
			}
			recs[r].changed = changed;
		}
		levelSync();
		// ---- U ---- (volUpLevel on the LDS records)
		for (u32 cd = 1; cd <= L; ++cd) {
			for (u32 r = lstart[cd] + threadIdx.x; r < lstart[cd - 1u]; r += blockDim.x) {
				const VolRec me = recs[r];
				bool ret = true;
				if (me.changed) {
					// updateNode of a node with children (OMB:1191-1224): summary, collapse, compare
					const u32 f = t.flags(me.slot);
					const Summ sm = blockSummary(t, g, me.slot, cd, f);
					if (sm.collapsible) collapseBlock(t, me.slot, me.lk);
					ret = writeToParent(t, g, me.slot, me.lk, sm);
				}
				if (ret && me.parent != NONE) {
					atomicOr(&recs[me.parent].changed, 1u);
					lmark[cd + 1u] = 1u;
				}
			}
			levelSync();
			if (0 == lmark[cd + 1u]) break;  // (a record is only ever marked from the level below)
		}
	}
	// ---- the finished control block to the host's pinned copy, then the word the host polls (as k_vol_all) ----
	if (threadIdx.x < 64u) {
		u32 ng, nu;
		tableCounts(t, threadIdx.x, &ng, &nu);
		if (0 == threadIdx.x) {
			ctl->used_now = __hip_atomic_load(&t.root->used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			ctl->used_g_now = ng;
			ctl->used_u_now = nu;
		}
	}
	levelSync();
	{
		constexpr u32 W_CORE = offsetof(ScanCtl, dbg) / 4u;
		const u32* dev = reinterpret_cast<const u32*>(ctl);
		u32* host = reinterpret_cast<u32*>(host_result);
		for (u32 w = threadIdx.x; w < W_CORE; w += blockDim.x) host[w] = __hip_atomic_load(&dev[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	__syncthreads();  // (every thread's stores to the pinned block have been acknowledged: the barrier waits for them)
	if (0 == threadIdx.x) __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_result + 1), done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// min_depth == depth_levels: the root itself is set (OMB:505-511) -- every block dies
__global__ __launch_bounds__(256) void k_vol_root(Table t, MapGeom g, float val)
{
	const u32 ncap = t.mask + 1;
	for (u32 s = blockIdx.x * blockDim.x + threadIdx.x; s < ncap; s += gridDim.x * blockDim.x) {
		if (0 == t.key(s)) continue;
		t.flags(s) = (t.flags(s) & ~(F_INNER | F_DIRTY | F_SUB)) | F_DEAD;
	}
	if (0 == blockIdx.x && 0 == threadIdx.x) {
		t.root->occ = val;
		t.root->flags = (isFreeV(g, val) ? 1u : 0u) | (isUnknownV(g, val) ? 2u : 0u);
	}
}

// ------------------------------------------------------------------------------------------------
// Point queries: getState / isOccupied / isFree / isUnknown / containsFree / containsUnknown /
// getOccupancy of a coordinate at a depth (occupancy_map_base.h:599-728), all evaluated on the node that
// Octree::getNode(toCode(coord, depth)) returns (octree.h:974-985) -- whose loop stops one level early: on a fully
// expanded path that is the node at depth + 1 (reported as depth), else the leaf that ends the path (true depth).
// Reproduced as is: the callers of the reference see exactly this. One thread per query, one hash lookup per level.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_query(Table t, MapGeom g, const double* __restrict__ xyz, u32 n, u32 depth,
                                               float* __restrict__ logodds, uint8_t* __restrict__ state)
{
	for (u32 q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
		const u64 code = morton3(toKey1(g, xyz[3 * (size_t)q], depth), toKey1(g, xyz[3 * (size_t)q + 1], depth),
		                         toKey1(g, xyz[3 * (size_t)q + 2], depth));
		float occ = t.root->occ;
		u32 fl = t.root->flags & 3u;
		u32 rd = depth;
		u64 bkey = 1;  // key of the current node's children block
		u32 bs = tableFind(t, bkey);
		bool leaf = bs == NONE || (t.flags(bs) & F_DEAD);
		for (u32 d = g.L - 1; d > depth; --d) {
			if (leaf) {  // !hasChildren (octree.h:979-981)
				rd = d + 1;
				break;
			}
			const u32 ci = (u32)((code >> (3 * d)) & 7);  // getChildIdx (code.h:245-248)
			const u32 f = t.flags(bs);
			occ = t.occ(bs)[ci];
			fl = ((f >> ci) & 1u) | (((f >> (8 + ci)) & 1u) << 1);
			bkey = (bkey << 3) | (u64)ci;
			leaf = true;
			if (f & (1u << (16 + ci))) {
				bs = tableFind(t, bkey);
				leaf = bs == NONE || (t.flags(bs) & F_DEAD);
			}
		}
		logodds[q] = occ;
		uint8_t st = (g.occ_thr < (double)occ) ? 1 : (isFreeV(g, occ) ? 2 : 4);
		const bool cfree = (0 == rd) ? isFreeV(g, occ) : (0 != (fl & 1u));
		const bool cunk = (0 == rd) ? isUnknownV(g, occ) : (0 != (fl & 2u));
		state[q] = st | (cfree ? 8 : 0) | (cunk ? 16 : 0);
	}
}

// ------------------------------------------------------------------------------------------------
// Leaf / tree iteration with bounding volume, state filter and min_depth: what
// beginLeaves / beginTree (occupancy_map_base.h:93-165) yield when run to the end. The reference's iterator walks
// depth-first (iterator/octree.h:255-300), descending into a node only if validNode holds for it
// (iterator/occupancy_map.h:168-190: the node's box intersects the bounding volume AND the state filter -- on the
// contains_* summaries unless the node sits at min_depth and `contains` is off) and returning it if validReturnNode
// holds (192-207: leaves and min_depth nodes for the leaf iterator). Here the same predicates run level by level:
// one record per inner node to descend into, its centre carried along because the reference accumulates child
// centres from the root down (octree.h:625-633); matching nodes are appended unordered and the host sorts them
// into pre-order.
// ------------------------------------------------------------------------------------------------
struct IterArgs {
	double vc[3], vh[3];  // the bounding volume: AABB centre, half size
	u32 has_bv, occ, fre, unk, contains, min_depth, only_leaf, pad;
};
struct IterRec {
	u64 lk;  // key of the node's children block
	double c[3];
};
struct IterOut {
	u64* codes;
	u8* depths;
	float* occ;
	u32* rgb;
	u8* flags;
	u32 cap;
};
__device__ inline bool iterState(const MapGeom& g, const IterArgs& a, float v, u32 cf, u32 cu, u32 depth, bool use_contains)
{
	const bool isocc = g.occ_thr < (double)v, isfree = isFreeV(g, v), isunk = isUnknownV(g, v);
	if (use_contains) {
		// containsOccupied = isOccupied; containsFree / containsUnknown: own state at depth 0, stored flags above (OMB:946-980)
		const bool cfree = (0 == depth) ? isfree : (0 != cf), cunk = (0 == depth) ? isunk : (0 != cu);
		return (a.occ && isocc) || (a.unk && cunk) || (a.fre && cfree);
	}
	return (a.occ && isocc) || (a.unk && isunk) || (a.fre && isfree);
}
__device__ inline void iterVisit(const Table& t, const MapGeom& g, const IterArgs& a, u64 node_code, u32 depth, const double c[3], float v,
                                 u32 cf, u32 cu, u32 rgb, bool has_children, u64 child_lk, IterRec* __restrict__ rec, u32 rcap,
                                 const IterOut& out, ScanCtl* ctl)
{
	if (a.has_bv) {
		VolArgs va;
		for (int k = 0; k < 3; ++k) {
			va.vc[k] = a.vc[k];
			va.vh[k] = a.vh[k];
		}
		if (!volIntersects(va, c, g.hs[depth])) return;  // Base::validNode (iterator/octree.h:214-244)
	}
	if (!iterState(g, a, v, cf, cu, depth, a.contains || a.min_depth != depth)) return;  // validNode
	const bool leaf = !has_children;
	const bool ret = a.only_leaf ? ((a.min_depth == depth || leaf) && iterState(g, a, v, cf, cu, depth, false))
	                             : iterState(g, a, v, cf, cu, depth, 0 != a.contains);
	if (ret) {
		const u32 pos = atomicAdd(&ctl->n_codes, 1u);
		if (pos < out.cap) {
			out.codes[pos] = node_code;
			out.depths[pos] = (u8)depth;
			out.occ[pos] = v;
			out.rgb[pos] = rgb;
			out.flags[pos] = (u8)((0 == depth ? ((isFreeV(g, v) ? 1u : 0u) | (isUnknownV(g, v) ? 2u : 0u)) : ((cf ? 1u : 0u) | (cu ? 2u : 0u))) |
			                      (leaf ? 4u : 0u));
		}
	}
	if (!(depth <= a.min_depth || leaf)) {  // singleIncrement: descend (iterator/octree.h:266-274)
		const u32 rp = atomicAdd(&ctl->dl_total, 1u);
		if (rp < rcap) {
			IterRec r;
			r.lk = child_lk;
			r.c[0] = c[0];
			r.c[1] = c[1];
			r.c[2] = c[2];
			rec[rp] = r;
		} else {
			atomicOr(&ctl->err, ERR_ENTRIES);
		}
	}
}
__global__ void k_iter_root(Table t, MapGeom g, IterArgs a, IterRec* __restrict__ rec, u32 rcap, IterOut out, ScanCtl* ctl)
{
	ctl->dl_start[g.L] = 0;
	if (g.L >= a.min_depth) {  // init(): `current_depth_ >= min_depth_` (iterator/octree.h:203)
		const u32 rs = tableFind(t, 1);
		const bool has_children = rs != NONE && !(t.flags(rs) & F_DEAD);
		const double c[3] = {0.0, 0.0, 0.0};
		iterVisit(t, g, a, 0, g.L, c, t.root->occ, t.root->flags & 1u, t.root->flags & 2u, t.root->rgb, has_children, 1, rec, rcap, out, ctl);
	}
	ctl->dl_start[g.L - 1] = ctl->dl_total;
}
// the records of depth `cd` nodes are [dl_start[cd], dl_start[cd-1]); their children (depth cd-1) are visited
__global__ __launch_bounds__(256) void k_iter_level(Table t, MapGeom g, IterArgs a, u32 cd, IterRec* __restrict__ rec, u32 rcap, IterOut out,
                                                    ScanCtl* ctl)
{
	const u32 lo = ctl->dl_start[cd], hi = min(ctl->dl_start[cd - 1], rcap);
	const u32 n = (hi - lo) * 8u;
	const u32 child_depth = cd - 1;
	const double chs = g.hs[child_depth];
	for (u32 w = blockIdx.x * blockDim.x + threadIdx.x; w < n; w += gridDim.x * blockDim.x) {
		const IterRec me = rec[lo + (w >> 3)];
		const u32 i = w & 7u;
		const u32 s = tableFind(t, me.lk);
		if (s == NONE) continue;
		const u32 f = t.flags(s);
		double cc[3] = {me.c[0], me.c[1], me.c[2]};  // getChildCenter (octree.h:625-633)
		cc[0] += ((i & 1) ? chs : -chs);
		cc[1] += ((i & 2) ? chs : -chs);
		cc[2] += ((i & 4) ? chs : -chs);
		const u64 clk = (me.lk << 3) | (u64)i;
		bool has_children = false;
		if (child_depth >= 1 && ((f >> (16 + i)) & 1u)) {
			const u32 cs = tableFind(t, clk);
			has_children = cs != NONE && !(t.flags(cs) & F_DEAD);
		}
		const u64 code = clk ^ (1ULL << (3 * (g.L - child_depth)));
		iterVisit(t, g, a, code, child_depth, cc, t.occ(s)[i], (f >> i) & 1u, (f >> (8 + i)) & 1u, t.rgb ? t.rgb[8 * (size_t)s + i] : 0u,
		          has_children, clk, rec, rcap, out, ctl);
	}
}

// ------------------------------------------------------------------------------------------------
// read-back
// ------------------------------------------------------------------------------------------------
struct DumpCtl {
	unsigned long long n_out;
	unsigned long long n_live;
	unsigned long long n_leaf;
};

// Room for `cnt` records of the calling thread in a list that ONE 64-bit counter fills: one atomic per workgroup and call (the
// counter is one word: ~12 ns per atomic whoever issues it -- a returning atomic per leaf made the export of the bench map's
// 2.3e5 leaves 2.3 ms and that of a 2 mm RGB-D frame's 3.5e8 leaves 0.23 s). Every thread of the workgroup (256) calls it.
__device__ inline unsigned long long blockReserve(unsigned long long* counter, u32 cnt)
{
	__shared__ u32 wsum[4];
	__shared__ unsigned long long bbase;
	const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	u32 incl = cnt;
	for (int o = 1; o < 64; o <<= 1) {
		const u32 v = (u32)__shfl_up((int)incl, o);
		if ((int)lane >= o) incl += v;
	}
	if (63u == lane) wsum[wave] = incl;
	__syncthreads();
	if (0 == threadIdx.x) {
		const u32 tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
		bbase = tot ? atomicAdd(counter, (unsigned long long)tot) : 0ull;
	}
	__syncthreads();
	unsigned long long pos = bbase + (incl - cnt);
	for (u32 w = 0; w < wave; ++w) pos += wsum[w];
	__syncthreads();  // (wsum / bbase are reused by the next call)
	return pos;
}
#define UFO_EXPORT_SLOTS 4u  // table slots a thread takes per reservation

__global__ __launch_bounds__(256) void k_export_leaves(Table t, MapGeom g, int include_unknown, u64* __restrict__ codes,
                                                       uint8_t* __restrict__ depths, float* __restrict__ occ,
                                                       u32* __restrict__ rgb, unsigned long long cap, DumpCtl* dc)
{
	const u32 ncap = t.mask + 1;
	const u64 span = (u64)gridDim.x * blockDim.x * UFO_EXPORT_SLOTS;
	u32 n_live = 0, n_leaf = 0;
	for (u64 s0 = (u64)blockIdx.x * blockDim.x * UFO_EXPORT_SLOTS; s0 < ncap; s0 += span) {  // (uniform: barriers inside)
		u32 f[UFO_EXPORT_SLOTS], level[UFO_EXPORT_SLOTS], outm[UFO_EXPORT_SLOTS];
		u64 lk[UFO_EXPORT_SLOTS];
		float v[UFO_EXPORT_SLOTS][8];
		u32 cnt = 0;
#pragma unroll
		for (u32 k = 0; k < UFO_EXPORT_SLOTS; ++k) {
			const u64 s = s0 + (u64)k * blockDim.x + threadIdx.x;
			outm[k] = 0;
			lk[k] = s < ncap ? t.key((u32)s) : 0ull;
			f[k] = lk[k] ? t.flags((u32)s) : F_DEAD;
			if (0 == lk[k] || (f[k] & F_DEAD)) continue;
			level[k] = levelOf(g, lk[k]);
			++n_live;
			const float4* po = reinterpret_cast<const float4*>(t.occ((u32)s));
			const float4 a = po[0], b = po[1];
			v[k][0] = a.x; v[k][1] = a.y; v[k][2] = a.z; v[k][3] = a.w;
			v[k][4] = b.x; v[k][5] = b.y; v[k][6] = b.z; v[k][7] = b.w;
#pragma unroll
			for (u32 i = 0; i < 8; ++i) {
				if (level[k] > 1 && ((f[k] >> (16 + i)) & 1u)) continue;
				++n_leaf;
				if (!include_unknown && isUnknownV(g, v[k][i])) continue;
				outm[k] |= 1u << i;
			}
			cnt += (u32)__popc(outm[k]);
		}
		unsigned long long pos = blockReserve(&dc->n_out, cnt);
#pragma unroll
		for (u32 k = 0; k < UFO_EXPORT_SLOTS; ++k) {
			if (0 == outm[k]) continue;
			const u64 s = s0 + (u64)k * blockDim.x + threadIdx.x;
			const u64 p = lk[k] ^ (1ULL << (3 * (g.L - level[k])));
#pragma unroll
			for (u32 i = 0; i < 8; ++i) {
				if (!((outm[k] >> i) & 1u)) continue;
				if (pos < cap) {
					codes[pos] = (p << 3) | (u64)i;
					depths[pos] = (uint8_t)(level[k] - 1);
					occ[pos] = v[k][i];
					rgb[pos] = t.rgb ? t.rgb[8 * (size_t)s + i] : 0u;
				}
				++pos;
			}
		}
	}
	// the two totals: once per wave
	for (int o = 32; o > 0; o >>= 1) {
		n_live += (u32)__shfl_xor((int)n_live, o);
		n_leaf += (u32)__shfl_xor((int)n_leaf, o);
	}
	if (0 == (threadIdx.x & 63u)) {
		if (n_live) atomicAdd(&dc->n_live, (unsigned long long)n_live);
		if (n_leaf) atomicAdd(&dc->n_leaf, (unsigned long long)n_leaf);
	}
}

__global__ __launch_bounds__(256) void k_export_inner(Table t, MapGeom g, u64* __restrict__ codes, uint8_t* __restrict__ depths,
                                                      float* __restrict__ occ, uint8_t* __restrict__ flags,
                                                      u32* __restrict__ rgb, unsigned long long cap, DumpCtl* dc)
{
	const u32 ncap = t.mask + 1;
	const u64 span = (u64)gridDim.x * blockDim.x * UFO_EXPORT_SLOTS;
	for (u64 s0 = (u64)blockIdx.x * blockDim.x * UFO_EXPORT_SLOTS; s0 < ncap; s0 += span) {  // (uniform: barriers inside)
		u64 lk[UFO_EXPORT_SLOTS];
		u32 cnt = 0;
#pragma unroll
		for (u32 k = 0; k < UFO_EXPORT_SLOTS; ++k) {
			const u64 s = s0 + (u64)k * blockDim.x + threadIdx.x;
			lk[k] = s < ncap ? t.key((u32)s) : 0ull;
			if (lk[k] && (t.flags((u32)s) & F_DEAD)) lk[k] = 0ull;
			cnt += lk[k] ? 1u : 0u;
		}
		unsigned long long pos = blockReserve(&dc->n_out, cnt);
#pragma unroll
		for (u32 k = 0; k < UFO_EXPORT_SLOTS; ++k) {
			if (0 == lk[k]) continue;
			const u32 s = (u32)(s0 + (u64)k * blockDim.x + threadIdx.x);
			const unsigned long long at = pos++;
			if (at >= cap) continue;
			const u32 level = levelOf(g, lk[k]);
			codes[at] = lk[k] ^ (1ULL << (3 * (g.L - level)));
			depths[at] = (uint8_t)level;
			if (1 == lk[k]) {
				occ[at] = t.root->occ;
				flags[at] = (uint8_t)(t.root->flags & 3u);
				rgb[at] = t.root->rgb;
			} else {
				u32 p = t.parent(s);
				u32 ci = (u32)(lk[k] & 7);
				u32 fp = t.flags(p);
				occ[at] = t.occ(p)[ci];
				flags[at] = (uint8_t)(((fp >> ci) & 1u) | (((fp >> (8 + ci)) & 1u) << 1));
				rgb[at] = t.rgb ? t.rgb[8 * (size_t)p + ci] : 0u;
			}
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Order-independent fingerprint of the canonical dump (ufomap_map_digest): per record
//   h = mix64(mix64(code >> 3*depth | depth << 58) ^ (float bits | rgb << 32 | flags << 56))
// out[0..2] = count, sum, xor over the leaves; out[3..5] the same over the inner nodes. Maps that are too large to
// export and sort on the host (config C3 at insert depth 0: 3.4e8 leaves) are compared through it, and replicas of
// one map on several GPUs can check each other with 48 bytes.
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline u64 mix64(u64 z)
{
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}
__host__ __device__ inline u64 digestRecord(u64 code_shifted, u32 depth, float occ, u32 rgb, u32 flags)
{
	u32 ob;
	memcpy(&ob, &occ, 4);
	return mix64(mix64(code_shifted | ((u64)depth << 58)) ^ ((u64)ob | ((u64)(rgb & 0xFFFFFFu) << 32) | ((u64)(flags & 0xFFu) << 56)));
}
__global__ __launch_bounds__(256) void k_digest(Table t, MapGeom g, int include_unknown, unsigned long long* __restrict__ out)
{
	u32 ncap = t.mask + 1;
	u64 acc[6] = {0, 0, 0, 0, 0, 0};
	for (u32 s = blockIdx.x * blockDim.x + threadIdx.x; s < ncap; s += gridDim.x * blockDim.x) {
		const u64 lk = t.key(s);
		if (0 == lk) continue;
		const u32 f = t.flags(s);
		if (f & F_DEAD) continue;
		const u32 level = levelOf(g, lk);
		const u64 p = lk ^ (1ULL << (3 * (g.L - level)));
		{
			float v;
			u32 fl, c;
			if (1 == lk) {
				v = t.root->occ;
				fl = t.root->flags & 3u;
				c = t.root->rgb;
			} else {
				const u32 pp = t.parent(s), ci = (u32)(lk & 7), fp = t.flags(pp);
				v = t.occ(pp)[ci];
				fl = ((fp >> ci) & 1u) | (((fp >> (8 + ci)) & 1u) << 1);
				c = t.rgb ? t.rgb[8 * (size_t)pp + ci] : 0u;
			}
			const u64 h = digestRecord(p, level, v, c, fl);
			acc[3] += 1;
			acc[4] += h;
			acc[5] ^= h;
		}
		for (u32 i = 0; i < 8; ++i) {
			if (level > 1 && ((f >> (16 + i)) & 1u)) continue;
			const float v = t.occ(s)[i];
			if (!include_unknown && isUnknownV(g, v)) continue;
			const u64 h = digestRecord((p << 3) | (u64)i, level - 1, v, t.rgb ? t.rgb[8 * (size_t)s + i] : 0u, 0u);
			acc[0] += 1;
			acc[1] += h;
			acc[2] ^= h;
		}
	}
	for (int k = 0; k < 6; ++k) {
		u64 v = acc[k];
		const bool x = (2 == k || 5 == k);
		for (int o = 32; o > 0; o >>= 1) {
			const u64 w = __shfl_xor(v, o);
			v = x ? (v ^ w) : (v + w);
		}
		if (0 == __lane_id() && v) {
			if (x) atomicXor(&out[k], (unsigned long long)v);
			else atomicAdd(&out[k], (unsigned long long)v);
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Map byte stream: the node part of Octree::write / OccupancyMapBase::writeNodes
// (octree.h:833-917, occupancy_map_base.h:1457-1533): pre-order; per inner node of depth >= 2 one byte
// "child i has children", then per child either its subtree or its leaf payload (float log-odds [+ 3 bytes
// r,g,b]); the eight leaves of a depth-1 node follow each other without a mask byte. Pre-order offsets come
// from subtree sizes: sizes bottom-up (one launch per level), offsets and bytes top-down.
// ------------------------------------------------------------------------------------------------
// (The per-level counters are hot words: tens of thousands of live blocks on 16 of them. A workgroup counts in LDS and adds
// its totals with one atomic per level -- straight atomics serialise at ~12 ns each: 0.4 ms for a 30 k-block map.)
// The list of live blocks, level by level, WITHOUT atomics on device memory: every workgroup takes a contiguous share of the
// table's slots, counts its live blocks per level (k_ser_count: one row of 32 counts per workgroup), k_ser_prefix turns the rows
// into where each workgroup's blocks of a level start, k_ser_collect walks the same share again and writes. (Until round 4's
// last day each workgroup added its counts to ONE row of 32 words -- twice 9 000 atomics on two cache lines for the bench map,
// ~12 ns apiece whatever is in flight: most of the 0.2 ms a publish took.)
#define UFO_SER_NB_MAX 4096u
// workgroups of the two listing kernels for a table of ncap slots (a multiple of 32: k_ser_prefix gives 32 lanes to a level)
__host__ __device__ inline u32 serBlocks(u64 ncap)
{
	u64 nb = (ncap + 4095u) / 4096u;
	nb = (nb + 31u) & ~31ull;
	return (u32)(nb < 32u ? 32u : (nb > UFO_SER_NB_MAX ? UFO_SER_NB_MAX : nb));
}
// the share of workgroup b: [lo, hi), whole multiples of the workgroup's 256 threads
__device__ inline void serShare(u32 ncap, u32 nblocks, u32 b, u32* lo, u32* hi)
{
	const u64 per = ((((u64)ncap + nblocks - 1u) / nblocks) + 255u) & ~255ull;
	const u64 a = per * b, e = a + per;
	*lo = (u32)(a < ncap ? a : ncap);
	*hi = (u32)(e < ncap ? e : ncap);
}
__global__ __launch_bounds__(256) void k_ser_count(Table t, MapGeom g, u32* __restrict__ blk_cnt /* [gridDim.x][32] */)
{
	__shared__ u32 cnt[32];
	if (threadIdx.x < 32u) cnt[threadIdx.x] = 0;
	__syncthreads();
	u32 lo, hi;
	serShare(t.mask + 1u, gridDim.x, blockIdx.x, &lo, &hi);
	for (u32 s = lo + threadIdx.x; s < hi; s += blockDim.x) {
		u64 lk = t.key(s);
		if (0 == lk || (t.flags(s) & F_DEAD)) continue;
		atomicAdd(&cnt[levelOf(g, lk)], 1u);
	}
	__syncthreads();
	if (threadIdx.x < 32u) blk_cnt[32u * blockIdx.x + threadIdx.x] = cnt[threadIdx.x];
}
__global__ __launch_bounds__(256) void k_ser_collect(Table t, MapGeom g, const u32* __restrict__ level_off, const u32* __restrict__ blk_base /* [gridDim.x][32], k_ser_prefix */,
                                                     u32* __restrict__ list, u32 list_cap, unsigned long long* __restrict__ pos_out = nullptr,
                                                     unsigned long long* __restrict__ off_out = nullptr)
{
	__shared__ u32 run[32];  // where the workgroup's next block of a level goes, counted from the level's first entry
	if (threadIdx.x < 32u) run[threadIdx.x] = blk_base[32u * blockIdx.x + threadIdx.x];
	__syncthreads();
	u32 lo, hi;
	serShare(t.mask + 1u, gridDim.x, blockIdx.x, &lo, &hi);
	for (u32 s = lo + threadIdx.x; s < hi; s += blockDim.x) {
		u64 lk = t.key(s);
		if (0 == lk || (t.flags(s) & F_DEAD)) continue;
		const u32 l = levelOf(g, lk);
		const u32 at = level_off[l] + atomicAdd(&run[l], 1u);
		if (at < list_cap) list[at] = s;
		// (where the block stands in the list, in the array of subtree sizes: read by k_ser_tail_prep for the blocks of the narrow levels --
		// their sizes are never stored there -- and overwritten by the size for the others before anybody reads it)
		if (pos_out) pos_out[s] = at;
		// (... and "no place in the stream yet" for every block that can get one: the serialiser's offsets need no fill of the whole array)
		if (off_out) off_out[s] = ~0ull;
	}
}
// Bounding volume and min_depth of Octree::write / writeData (octree.h:779-917): only children whose box intersects the
// volume are written (the mask byte is written regardless), and nodes at depth <= min_depth are written as leaves.
struct SerArgs {
	double vc[3], vh[3];
	u32 has_bv, min_depth;
};
// centre of the node whose children block has key lk (node depth = level), accumulated from the root down exactly as
// writeNodesRecurs does through getChildCenter (octree.h:625-633)
__device__ inline void keyCenter(const MapGeom& g, u64 lk, u32 level, double c[3])
{
	c[0] = c[1] = c[2] = 0.0;
	for (u32 d = g.L; d-- > level;) {
		const u32 idx = (u32)((lk >> (3 * (d - level))) & 7u);
		const double hs = g.hs[d];
		c[0] += ((idx & 1) ? hs : -hs);
		c[1] += ((idx & 2) ? hs : -hs);
		c[2] += ((idx & 4) ? hs : -hs);
	}
}
__device__ inline bool serChildIn(const SerArgs& sa, const double c[3], u32 i, double chs)
{
	if (!sa.has_bv) return true;
	const double cc[3] = {c[0] + ((i & 1) ? chs : -chs), c[1] + ((i & 2) ? chs : -chs), c[2] + ((i & 4) ? chs : -chs)};
	VolArgs va;
	for (int k = 0; k < 3; ++k) {
		va.vc[k] = sa.vc[k];
		va.vh[k] = sa.vh[k];
	}
	return volIntersects(va, cc, chs);
}
// EIGHT LANES PER BLOCK (lane = child): the children's lookups -- a hash probe each, a dependent chain of global loads --
// run side by side instead of one after the other (the narrow levels are walked by one workgroup, a barrier per level:
// their time is this chain times the number of levels), the sum is three xor shuffles.
__device__ inline void serSizesLevel(const Table& t, const MapGeom& g, const SerArgs& sa, const u32* __restrict__ list, u32 n, u32 level, u32 D,
                                     u64* __restrict__ size)
{
	const double chs = g.hs[level - 1];
	const u32 ch = threadIdx.x & 7u;
	for (u32 i = (blockIdx.x * blockDim.x + threadIdx.x) >> 3; i < n; i += (gridDim.x * blockDim.x) >> 3) {  // (uniform per group of 8 lanes)
		const u32 s = list[i];
		const u64 lk = t.key(s);
		double c[3] = {0, 0, 0};
		if (sa.has_bv) keyCenter(g, lk, level, c);
		const u32 f = t.flags(s);
		unsigned long long add = 0;
		if (serChildIn(sa, c, ch, chs)) {
			add = D;
			if (level >= 2 && level - 1 > sa.min_depth && ((f >> (16 + ch)) & 1u)) {
				const u32 cs = tableFindChild(t, s, lk, ch);
				if (cs != NONE && !(t.flags(cs) & F_DEAD)) add = size[cs];
			}
		}
		for (int o = 1; o < 8; o <<= 1) add += __shfl_xor(add, o);
		if (0 == ch) size[s] = add + ((1 == level) ? 0ull : 1ull);  // the eight leaves of a depth-1 node follow each other without a mask byte
	}
}
__global__ __launch_bounds__(256) void k_ser_sizes(Table t, MapGeom g, SerArgs sa, const u32* __restrict__ list, u32 n, u32 level, u32 D,
                                                   u64* __restrict__ size)
{
	serSizesLevel(t, g, sa, list, n, level, D, size);
}
// the levels l_from .. l_to (towards the root: few blocks each) by ONE workgroup, a barrier per level instead of a launch
struct SerLevels {
	u32 off[32], cnt[32];
};
__global__ __launch_bounds__(1024) void k_ser_sizes_tail(Table t, MapGeom g, SerArgs sa, const u32* __restrict__ list, SerLevels lv, u32 l_from, u32 l_to,
                                                         u32 D, u64* __restrict__ size, unsigned long long* __restrict__ total_out)
{
	for (u32 l = l_from; l <= l_to; ++l) {
		if (lv.cnt[l]) serSizesLevel(t, g, sa, list + lv.off[l], lv.cnt[l], l, D, size);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__syncthreads();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	}
	// the stream's length: the 0xFF byte + the subtree of the root block (the last level's only block)
	if (0 == threadIdx.x && total_out) *total_out = 1ull + size[list[lv.off[l_to]]];
}
__device__ inline void serPutLeaf(uint8_t* __restrict__ out, u64 at, float v, u32 rgb, u32 D)
{
	memcpy(out + at, &v, 4);  // (one unaligned 32-bit store: global memory takes them, and the stream has no alignment)
	if (D > 4) {
		out[at + 4] = (uint8_t)rgb;
		out[at + 5] = (uint8_t)(rgb >> 8);
		out[at + 6] = (uint8_t)(rgb >> 16);
	}
}
// off[] is pre-set to ~0: a block whose offset nobody wrote lies outside the bounding volume (or below min_depth).
// Eight lanes per block as in serSizesLevel; a child's position is a prefix sum over the children before it.
__device__ inline void serWriteLevel(const Table& t, const MapGeom& g, const SerArgs& sa, const u32* __restrict__ list, u32 n, u32 level, u32 D,
                                     const u64* __restrict__ size, u64* __restrict__ off, uint8_t* __restrict__ out)
{
	const double chs = g.hs[level - 1];
	const u32 ch = threadIdx.x & 7u;
	for (u32 i = (blockIdx.x * blockDim.x + threadIdx.x) >> 3; i < n; i += (gridDim.x * blockDim.x) >> 3) {  // (uniform per group of 8 lanes)
		const u32 s = list[i];
		const u64 lk = t.key(s);
		const u64 at0 = (1 == lk) ? 1ull : off[s];  // the root's subtree starts behind the 0xFF byte of writeNodes
		if (at0 == ~0ull) continue;
		double c[3] = {0, 0, 0};
		if (sa.has_bv) keyCenter(g, lk, level, c);
		const u32 f = t.flags(s);
		u32 cslot = NONE;
		if (level >= 2 && level - 1 > sa.min_depth && ((f >> (16 + ch)) & 1u)) {
			const u32 cs = tableFindChild(t, s, lk, ch);
			if (cs != NONE && !(t.flags(cs) & F_DEAD)) cslot = cs;
		}
		u32 mask = (cslot != NONE) ? (1u << ch) : 0u;
		for (int o = 1; o < 8; o <<= 1) mask |= (u32)__shfl_xor((int)mask, o);
		const bool in = serChildIn(sa, c, ch, chs);
		const unsigned long long w = in ? ((cslot != NONE) ? (unsigned long long)size[cslot] : (unsigned long long)D) : 0ull;
		unsigned long long incl = w;
		for (int o = 1; o < 8; o <<= 1) {
			const unsigned long long v = __shfl_up(incl, o);
			if ((int)ch >= o) incl += v;
		}
		const u64 at = at0 + ((level >= 2) ? 1ull : 0ull) + (incl - w);
		if (0 == ch && level >= 2) out[at0] = (uint8_t)mask;  // written for all eight children, intersecting or not (OMB:1500-1512)
		if (in) {
			if (cslot != NONE) off[cslot] = at;
			else serPutLeaf(out, at, t.occ(s)[ch], t.rgb ? t.rgb[8 * (size_t)s + ch] : 0u, D);
		}
	}
}

__global__ __launch_bounds__(256) void k_ser_write(Table t, MapGeom g, SerArgs sa, const u32* __restrict__ list, u32 n, u32 level, u32 D,
                                                   const u64* __restrict__ size, u64* __restrict__ off, uint8_t* __restrict__ out)
{
	serWriteLevel(t, g, sa, list, n, level, D, size, off, out);
}
// the levels l_from down to l_to (from the root: few blocks each) by ONE workgroup; the first byte of the stream (0xFF,
// writeNodes) comes from here, too
__global__ __launch_bounds__(1024) void k_ser_write_tail(Table t, MapGeom g, SerArgs sa, const u32* __restrict__ list, SerLevels lv, u32 l_from, u32 l_to,
                                                         u32 D, const u64* __restrict__ size, u64* __restrict__ off, uint8_t* __restrict__ out)
{
	if (0 == threadIdx.x) out[0] = 0xFF;
	for (u32 l = l_from; l + 1 > l_to; --l) {
		if (lv.cnt[l]) serWriteLevel(t, g, sa, list + lv.off[l], lv.cnt[l], l, D, size, off, out);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__syncthreads();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		if (0 == l) break;
	}
}

// ---- the same without the host in the middle (maps of up to a few hundred thousand blocks: what a server publishes after
// every scan). The level counts stay on the device (k_ser_prefix turns them into the SerLevels the kernels read), the
// output buffer is sized by the table's fill, the two widest levels get a launch each with a fixed grid, the rest is the
// one-workgroup tail, and the last kernel copies the stream -- whose length only the device knows -- into pinned host
// memory with 16-byte stores: ONE stream synchronisation per serialisation instead of three.
// (1024 threads: 32 lanes per level. blk: k_ser_count's rows in, the workgroups' first entries within their levels out.)
__global__ __launch_bounds__(1024) void k_ser_prefix(u32* __restrict__ cnt /* out: [0..31] live blocks per level, [32..63] first list entry per level */,
                                                     SerLevels* lv, u32 list_cap, u32* __restrict__ blk, u32 nblocks)
{
	__shared__ u32 tot[32];
	const u32 level = threadIdx.x >> 5, sub = threadIdx.x & 31u, per = nblocks >> 5;  // (nblocks: a multiple of 32, serBlocks)
	u32 sum = 0;
	for (u32 k = 0; k < per; ++k) sum += blk[32u * (sub * per + k) + level];
	u32 incl = sum;
	for (int o = 1; o < 32; o <<= 1) {
		const u32 v = (u32)__shfl_up((int)incl, o, 32);
		if ((int)sub >= o) incl += v;
	}
	u32 runv = incl - sum;
	for (u32 k = 0; k < per; ++k) {
		const u32 idx = 32u * (sub * per + k) + level;
		const u32 c = blk[idx];
		blk[idx] = runv;
		runv += c;
	}
	if (31u == sub) tot[level] = incl;
	__syncthreads();
	if (0 != threadIdx.x) return;
	u32 off = 0;
	for (u32 l = 0; l < 32; ++l) off += tot[l];
	// (the host sized the block list from its view of the table's fill; should the map hold more live blocks than that --
	// it cannot after a join, but nothing here depends on it -- the levels are reported empty: the stream's length comes
	// out as 0 and the host takes the long way, which counts first)
	const bool fits = off <= list_cap;
	off = 0;
	for (u32 l = 0; l < 32; ++l) {
		const u32 c = fits ? tot[l] : 0u;
		lv->off[l] = off;
		lv->cnt[l] = c;
		cnt[l] = c;
		cnt[32 + l] = off;
		off += c;
	}
	cnt[96] = cnt[97] = 0u;  // (the stream's length: written by the narrow levels' kernel; no fill of the block in front of the listing)
}
__global__ __launch_bounds__(256) void k_ser_sizes_dev(Table t, MapGeom g, SerArgs sa, const u32* __restrict__ list, const SerLevels* lv, u32 level, u32 D,
                                                       u64* __restrict__ size)
{
	if (lv->cnt[level]) serSizesLevel(t, g, sa, list + lv->off[level], lv->cnt[level], level, D, size);
}
__global__ __launch_bounds__(1024) void k_ser_sizes_tail_dev(Table t, MapGeom g, SerArgs sa, const u32* __restrict__ list, const SerLevels* lvp, u32 l_from,
                                                             u32 l_to, u32 D, u64* __restrict__ size, unsigned long long* __restrict__ total_out)
{
	for (u32 l = l_from; l <= l_to; ++l) {
		if (lvp->cnt[l]) serSizesLevel(t, g, sa, list + lvp->off[l], lvp->cnt[l], l, D, size);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__syncthreads();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	}
	// (no live root block: the root is a leaf -- the host writes that stream itself; 0 tells it)
	if (0 == threadIdx.x) *total_out = lvp->cnt[l_to] ? 1ull + size[list[lvp->off[l_to]]] : 0ull;
}
__global__ __launch_bounds__(256) void k_ser_write_dev(Table t, MapGeom g, SerArgs sa, const u32* __restrict__ list, const SerLevels* lv, u32 level, u32 D,
                                                       const u64* __restrict__ size, u64* __restrict__ off, uint8_t* __restrict__ out, const unsigned long long* total,
                                                       unsigned long long cap)
{
	if (0 == *total || *total > cap) return;  // (uniform)
	if (lv->cnt[level]) serWriteLevel(t, g, sa, list + lv->off[level], lv->cnt[level], level, D, size, off, out);
}
__global__ __launch_bounds__(1024) void k_ser_write_tail_dev(Table t, MapGeom g, SerArgs sa, const u32* __restrict__ list, const SerLevels* lvp, u32 l_from,
                                                             u32 l_to, u32 D, const u64* __restrict__ size, u64* __restrict__ off, uint8_t* __restrict__ out,
                                                             const unsigned long long* total, unsigned long long cap)
{
	if (0 == *total || *total > cap) return;  // (uniform: nothing to write / the host's bound was too small -- it repeats the long way)
	if (0 == threadIdx.x) out[0] = 0xFF;
	for (u32 l = l_from; l + 1 > l_to; --l) {
		if (lvp->cnt[l]) serWriteLevel(t, g, sa, list + lvp->off[l], lvp->cnt[l], l, D, size, off, out);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__syncthreads();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		if (0 == l) break;
	}
}
// The narrow levels (a launch per level and pass would be fourteen launches of a few blocks each): k_ser_tail_prep + k_ser_tail_dev.
// Up to UFO_SER_TAIL_MAX blocks in these levels (else the two separate one-workgroup kernels above).
#define UFO_SER_TAIL_MAX 2048u
#define UFO_SER_TAIL_CHILD 0x80000000u  // gcs: the child's block is one of the narrow levels' -- the low bits are its place among them
#define UFO_SER_TAIL_OPEN 0xFFFFFFFFu   // gcw: ... and inside the volume: its share of the stream is that block's size, known to the one-workgroup pass
// What the narrow levels' pass needs to know about every child of every block in them, found by the whole chip at once (round 6: until
// then the one workgroup looked the children up itself, level after level -- a hash probe and three dependent loads per level and pass
// of 128 blocks: 97 us for the bench map's 1 400 blocks in levels 3 .. 16 against 6 + 12 us now). Eight lanes per block.
__global__ __launch_bounds__(256) void k_ser_tail_prep(Table t, MapGeom g, SerArgs sa, const u32* __restrict__ list, const SerLevels* lvp, u32 l_tail, u32 L, u32 D,
                                                       const u64* __restrict__ size, u32* __restrict__ gcw, u32* __restrict__ gcs)
{
	__shared__ u32 lo[32], ln[32];
	if (threadIdx.x < 32u) {
		lo[threadIdx.x] = lvp->off[threadIdx.x];
		ln[threadIdx.x] = lvp->cnt[threadIdx.x];
	}
	__syncthreads();
	const u32 base = lo[l_tail], nb = lo[L] + ln[L] - base;
	if (nb > UFO_SER_TAIL_MAX) return;  // (uniform; the one-workgroup pass says so to the host)
	const u32 ch = threadIdx.x & 7u;
	for (u32 j = (blockIdx.x * blockDim.x + threadIdx.x) >> 3; j < nb; j += (gridDim.x * blockDim.x) >> 3) {
		u32 l = l_tail;
		while (l < L && base + j >= lo[l] + ln[l]) ++l;
		const u32 s = list[base + j];
		const u64 lk = t.key(s);
		double c[3] = {0, 0, 0};
		if (sa.has_bv) keyCenter(g, lk, l, c);
		const u32 f = t.flags(s);
		u32 cslot = NONE;
		if (l >= 2 && l - 1 > sa.min_depth && ((f >> (16 + ch)) & 1u)) {
			const u32 cs = tableFindChild(t, s, lk, ch);
			if (cs != NONE && !(t.flags(cs) & F_DEAD)) cslot = cs;
		}
		const bool in = serChildIn(sa, c, ch, g.hs[l - 1]);
		u32 w = 0, code = cslot;
		if (cslot != NONE && l - 1 >= l_tail) {
			code = UFO_SER_TAIL_CHILD | ((u32)size[cslot] - base);  // (k_ser_collect's tag: the child block's place in the list)
			if (in) w = UFO_SER_TAIL_OPEN;
		} else if (in) {
			w = (cslot != NONE) ? (u32)size[cslot] : D;
		}
		gcw[8u * j + ch] = w;
		gcs[8u * j + ch] = code;
	}
}
// Both passes over the narrow levels by ONE workgroup in one launch: sizes bottom-up, then offsets and bytes top-down, on what
// k_ser_tail_prep left about every child (its block, its share of the stream) -- copied to LDS, where the sizes and offsets of these
// levels' blocks live too: a level costs LDS reads, a shuffle scan and a barrier, no trip to memory.
__global__ __launch_bounds__(1024) void k_ser_tail_dev(Table t, MapGeom g, const u32* __restrict__ list, const SerLevels* lvp, u32 l_tail, u32 L, u32 D,
                                                       const u32* __restrict__ gcw, const u32* __restrict__ gcs, u64* __restrict__ off, uint8_t* __restrict__ out,
                                                       unsigned long long* __restrict__ total_out, unsigned long long cap)
{
	__shared__ __attribute__((aligned(16))) u32 cw[UFO_SER_TAIL_MAX][8], cs_[UFO_SER_TAIL_MAX][8];
	__shared__ u32 szl[UFO_SER_TAIL_MAX], offl[UFO_SER_TAIL_MAX];  // the blocks' subtree sizes; where their subtrees start in the stream
	__shared__ u32 slotl[UFO_SER_TAIL_MAX];                        // their table slots (a leaf's payload is one dependent load, not two)
	__shared__ u32 lo[32], ln[32];
	if (threadIdx.x < 32u) {
		lo[threadIdx.x] = lvp->off[threadIdx.x];
		ln[threadIdx.x] = lvp->cnt[threadIdx.x];
	}
	__syncthreads();
	const u32 base = lo[l_tail], nb = lo[L] + ln[L] - base;
	if (nb > UFO_SER_TAIL_MAX) {  // (uniform) more blocks than the LDS arrays hold: the host takes the long way
		if (0 == threadIdx.x) *total_out = ~0ull;
		return;
	}
	{
		const uint4* a4 = reinterpret_cast<const uint4*>(gcw);
		const uint4* b4 = reinterpret_cast<const uint4*>(gcs);
		uint4* cw4 = reinterpret_cast<uint4*>(&cw[0][0]);
		uint4* cs4 = reinterpret_cast<uint4*>(&cs_[0][0]);
		for (u32 i = threadIdx.x; i < 2u * nb; i += blockDim.x) {
			cw4[i] = a4[i];
			cs4[i] = b4[i];
		}
		for (u32 i = threadIdx.x; i < nb; i += blockDim.x) {
			offl[i] = 0xFFFFFFFFu;
			slotl[i] = list[base + i];
		}
	}
	__syncthreads();
	const u32 ch = threadIdx.x & 7u;
	// ---- sizes, bottom-up (serSizesLevel, eight lanes per block) ----
	for (u32 l = l_tail; l <= L; ++l) {
		for (u32 i = threadIdx.x >> 3; i < ln[l]; i += blockDim.x >> 3) {
			const u32 j = lo[l] - base + i;
			u32 add = cw[j][ch];
			if (UFO_SER_TAIL_OPEN == add) {
				add = szl[cs_[j][ch] & ~UFO_SER_TAIL_CHILD];
				cw[j][ch] = add;
			}
			for (int o = 1; o < 8; o <<= 1) add += (u32)__shfl_xor((int)add, o);
			if (0 == ch) szl[j] = add + ((1 == l) ? 0u : 1u);
		}
		__syncthreads();
	}
	const unsigned long long total = ln[L] ? 1ull + (unsigned long long)szl[lo[L] - base] : 0ull;  // (no live root block: the root is a leaf, the host writes that stream itself)
	if (0 == threadIdx.x) *total_out = total;
	if (0 == total || total > cap) return;  // (uniform; too large for the host's bound cannot happen)
	// ---- offsets and bytes, top-down (serWriteLevel) ----
	if (0 == threadIdx.x) out[0] = 0xFF;
	for (u32 l = L; l + 1 > l_tail; --l) {
		for (u32 i = threadIdx.x >> 3; i < ln[l]; i += blockDim.x >> 3) {
			const u32 j = lo[l] - base + i;
			const u32 at0 = (l == L) ? 1u : offl[j];  // the root's subtree starts behind the 0xFF byte of writeNodes
			if (at0 == 0xFFFFFFFFu) continue;
			const u32 w = cw[j][ch], code = cs_[j][ch];
			u32 mask = (code != NONE) ? (1u << ch) : 0u;
			for (int o = 1; o < 8; o <<= 1) mask |= (u32)__shfl_xor((int)mask, o);
			u32 incl = w;
			for (int o = 1; o < 8; o <<= 1) {
				const u32 v = (u32)__shfl_up((int)incl, o);
				if ((int)ch >= o) incl += v;
			}
			const u32 at = at0 + ((l >= 2) ? 1u : 0u) + (incl - w);
			if (0 == ch && l >= 2) out[at0] = (uint8_t)mask;
			if (w) {
				if (NONE == code) {
					const u32 s = slotl[j];
					serPutLeaf(out, at, t.occ(s)[ch], t.rgb ? t.rgb[8 * (size_t)s + ch] : 0u, D);
				} else if (code & UFO_SER_TAIL_CHILD) {
					offl[code & ~UFO_SER_TAIL_CHILD] = at;
				} else {
					off[code] = at;
				}
			}
		}
		__syncthreads();
		if (0 == l) break;
	}
}
__global__ __launch_bounds__(256) void k_ser_copy_out(const uint4* __restrict__ out, const unsigned long long* __restrict__ total, unsigned long long cap,
                                                      uint4* __restrict__ h_out, unsigned long long* __restrict__ h_total, const SerLevels* lv, u32 l_tail, u32 L)
{
	const unsigned long long n = *total;
	if (0 == (blockIdx.x | threadIdx.x)) h_total[1] = lv->off[L] + lv->cnt[L] - lv->off[l_tail];  // blocks in the narrow levels: the host's choice next time
	if (n && n <= cap) {
		const unsigned long long n4 = (n + 15ull) >> 4;
		for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (unsigned long long)gridDim.x * blockDim.x) h_out[i] = out[i];
	}
	if (0 == (blockIdx.x | threadIdx.x)) *h_total = n;
}

// ------------------------------------------------------------------------------------------------
// Reading a node stream into the map (second half): OccupancyMapBase::readNodes / readNodesRecurs
// (occupancy_map_base.h:1379-1455). The stream is pre-order with implicit structure, so the host walks it once (sizes
// of earlier siblings decide where a subtree starts) and emits one ReadRec per node that has children in the stream;
// the device then does what the recursion does, level by level: createChildren on the node (a leaf is expanded by
// inheritance), for every child inside the bounding volume either the stream's leaf data (deleteChildren + readData +
// the leaf's own updateNode) or the recursion, and on the way back updateNode on every node of the stream.
// ------------------------------------------------------------------------------------------------
struct ReadRec {
	u64 lk;          // key of the node's children block
	u32 parent;      // record of the parent node (NONE for the root)
	u32 slot;        // table slot, filled by k_read_down
	u32 set_mask;    // children whose leaf data the stream holds
	u32 inner_mask;  // children that have children in the stream
	float val[8];
	u32 rgb[8];
	u32 pad[2];
};
__global__ __launch_bounds__(256) void k_read_down(Table t, MapGeom g, ReadRec* __restrict__ rec, u32 lo, u32 hi, u32 level, u32* __restrict__ kill,
                                                   u32 kcap, u32 scan_id, ScanCtl* ctl)
{
	const u32 max_probe = (t.mask >> 1) + 1;
	u32 n_created = 0;
	for (u32 r = lo + blockIdx.x * blockDim.x + threadIdx.x; r < hi; r += gridDim.x * blockDim.x) {
		const ReadRec me = rec[r];
		bool created;
		const u32 s = tableEnsure(t, me.lk, scan_id, max_probe, &created, &n_created);
		if (s == NONE) {
			atomicOr(&ctl->err, ERR_TABLE_FULL);
			continue;
		}
		rec[r].slot = s;
		if (created) {
			// createChildren (octree.h:1022-1058): the 8 children are copies of the node
			float v;
			u32 col = 0;
			if (1 == me.lk) {
				t.parent(s) = NONE;
				v = t.root->occ;
				col = t.root->rgb;
			} else {
				const u32 p = rec[me.parent].slot, ci = (u32)(me.lk & 7);
				t.parent(s) = p;
				atomicOr(&t.flags(p), 1u << (16 + ci));
				v = t.occ(p)[ci];
				if (g.color) col = t.rgb[8 * (size_t)p + ci];
			}
			const float4 vv = make_float4(v, v, v, v);
			float4* po = reinterpret_cast<float4*>(t.occ(s));
			po[0] = vv;
			po[1] = vv;
			if (g.color) {
				const uint4 cc = make_uint4(col, col, col, col);
				uint4* pc = reinterpret_cast<uint4*>(t.rgb + 8 * (size_t)s);
				pc[0] = cc;
				pc[1] = cc;
			}
			t.flags(s) = (isFreeV(g, v) ? F_CFREE : 0u) | (isUnknownV(g, v) ? F_CUNK : 0u);
		}
		u32 f = t.flags(s);
		for (u32 i = 0; i < 8; ++i) {
			if (!((me.set_mask >> i) & 1u)) continue;
			if (level >= 2 && (f & (1u << (16 + i)))) {
				// deleteChildren(child) (occupancy_map_base.h:1446): its whole subtree dies
				const u32 cs = tableFind(t, (me.lk << 3) | (u64)i);
				if (cs != NONE) {
					const u32 kp = atomicAdd(&ctl->n_codes, 1u);
					if (kp < kcap) kill[kp] = cs;
					else atomicOr(&ctl->err, ERR_ENTRIES);
				}
				f &= ~(1u << (16 + i));
			}
			const float v = me.val[i];
			t.occ(s)[i] = v;
			if (g.color) t.rgb[8 * (size_t)s + i] = me.rgb[i];
			// the leaf's own updateNode: indicators from its value (OMB:1181-1189)
			f = (f & ~((1u << i) | (1u << (8 + i)))) | (isFreeV(g, v) ? (1u << i) : 0u) | (isUnknownV(g, v) ? (1u << (8 + i)) : 0u);
		}
		// this block's word is private to this thread during the launch: other records' creations set bits in THEIR
		// parents' words, which are blocks of the previous (upper) level
		t.flags(s) = f;
	}
	for (int o = 32; o > 0; o >>= 1) n_created += __shfl_xor(n_created, o);
	if (__lane_id() == 0 && n_created) atomicAdd(&t.root->used, n_created);
}
// updateNode on the way back (occupancy_map_base.h:1438, 1447, 1452): every node of the stream, unconditionally
__global__ __launch_bounds__(256) void k_read_up(Table t, MapGeom g, const ReadRec* __restrict__ rec, u32 lo, u32 hi, u32 level, const ScanCtl* ctl)
{
	if (ctl->err) return;
	for (u32 r = lo + blockIdx.x * blockDim.x + threadIdx.x; r < hi; r += gridDim.x * blockDim.x) {
		const u32 s = rec[r].slot;
		const u64 lk = rec[r].lk;
		const u32 f = t.flags(s);
		const Summ sm = blockSummary(t, g, s, level, f);
		if (sm.collapsible) collapseBlock(t, s, lk);
		(void)writeToParent(t, g, s, lk, sm);
	}
}

// ------------------------------------------------------------------------------------------------
// table growth: re-insert every block into a table of twice/four times the capacity
// ------------------------------------------------------------------------------------------------
// Collapsed blocks (F_DEAD: the node is a leaf again, octree.h:1060-1066 deleteChildren) are not copied: nothing refers
// to them (the parent's INNER bit went with the collapse) and createNode makes a fresh block if the node is ever split
// again -- a re-hash is where their slots are reclaimed. `copied` counts what the new table holds (MapRoot::used).
__global__ __launch_bounds__(256) void k_rehash_copy(Table src, Table dst, u32* fail, u32* copied)
{
	u32 ncap = src.mask + 1;
	u32 mine = 0;
	for (u32 s = blockIdx.x * blockDim.x + threadIdx.x; s < ncap; s += gridDim.x * blockDim.x) {
		u64 lk = src.key(s);
		if (0 == lk) continue;
		if (src.flags(s) & F_DEAD) continue;
		++mine;
		const u32 d = tableInsertNew(dst, lk);
		if (d == NONE) {
			atomicOr(fail, 1u);
			continue;
		}
		const float4* so = reinterpret_cast<const float4*>(src.occ(s));
		float4* dofs = reinterpret_cast<float4*>(dst.occ(d));
		dofs[0] = so[0];
		dofs[1] = so[1];
		if (src.rgb) {
			const uint4* sc = reinterpret_cast<const uint4*>(src.rgb + 8 * (size_t)s);
			uint4* dcl = reinterpret_cast<uint4*>(dst.rgb + 8 * (size_t)d);
			dcl[0] = sc[0];
			dcl[1] = sc[1];
		}
		dst.flags(d) = src.flags(s);
		dst.stamp(d) = src.stamp(s);
	}
	for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
	if (0 == (threadIdx.x & 63u) && mine) atomicAdd(copied, mine);
}
// fill of a table: [0] groups claimed, [1] blocks of the first region (one wave)
__global__ void k_table_counts(Table t, u32* out)
{
	u32 g, u;
	tableCounts(t, threadIdx.x, &g, &u);
	if (0 == threadIdx.x) {
		out[0] = g;
		out[1] = u;
	}
}
__global__ __launch_bounds__(256) void k_rehash_parents(Table dst)
{
	u32 ncap = dst.mask + 1;
	for (u32 s = blockIdx.x * blockDim.x + threadIdx.x; s < ncap; s += gridDim.x * blockDim.x) {
		u64 lk = dst.key(s);
		if (0 == lk) continue;
		dst.parent(s) = (1 == lk) ? NONE : tableFind(dst, lk >> 3);
	}
}
}  // namespace ufo
