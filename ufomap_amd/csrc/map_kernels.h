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

// float exp as the reference's toProb sees it: std::exp(float) (OMB:911 with LogitType=float)
__device__ inline double toProbF(float logit) { return 1.0 / (1.0 + (double)((float)exp((double)(-logit)))); }

// updateNode for a non-leaf node (OMB:1191-1224) + colour average (OMC.cpp:177-222), read-only part
__device__ inline Summ blockSummary(const Table& t, const MapGeom& g, u32 s, u32 level, u32 f)
{
	Summ r;
	const float4* pv = reinterpret_cast<const float4*>(t.occ + 8 * (size_t)s);
	float4 a = pv[0], b = pv[1];
	float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
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
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			u32 c = pc[i];
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
__device__ inline bool writeToParent(const Table& t, const MapGeom& g, u32 s, u64 lk, const Summ& sm)
{
	if (1 == lk) {
		MapRoot* r = t.root;
		bool ch = r->occ != sm.occ || (r->flags & 3u) != sm.fl || (g.color && r->rgb != sm.rgb);
		r->occ = sm.occ;
		r->flags = sm.fl;
		r->rgb = sm.rgb;
		return ch;
	}
	u32 p = t.parent[s];
	u32 ci = (u32)(lk & 7);
	float* po = t.occ + 8 * (size_t)p + ci;
	u32 fp = __hip_atomic_load(&t.flags[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
		if (setm) atomicOr(&t.flags[p], setm);
		if (clrm) atomicAnd(&t.flags[p], ~clrm);
	}
	return ch;
}

// The node became a leaf again (deleteChildren, octree.h:1060-1066): mark the block DEAD and clear
// the parent's "child is inner" bit.
__device__ inline void collapseBlock(const Table& t, u32 s, u64 lk)
{
	atomicOr(&t.flags[s], F_DEAD);
	if (1 != lk) atomicAnd(&t.flags[t.parent[s]], ~(1u << (16 + (u32)(lk & 7))));
}

__device__ inline void markDirty(const Table& t, u32 p, u32 extra, u32* __restrict__ wl, u32* wl_count)
{
	u32 old = atomicOr(&t.flags[p], F_DIRTY | extra);
	if (!(old & F_DIRTY)) wl[atomicAdd(wl_count, 1u)] = p;
}

// ------------------------------------------------------------------------------------------------
// S1 ensure: createNode (octree.h:997-1016) for every entry, batched: the thread that creates or
// revives a block also links it to its parent and continues upward; stops at the first block that
// already existed (its ancestors exist by induction).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ensure(Table t, MapGeom g, const Entry* __restrict__ entries,
                                                const u32* n_entries_p, u32 scan_id, u32* __restrict__ ent_slot,
                                                u32* __restrict__ newlist, u32 newcap, ScanCtl* ctl)
{
	u32 n = *n_entries_p;
	const u32 max_probe = (t.mask >> 1) + 1;
	for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		u64 lk = entries[i].lk;
		bool cr;
		u32 s = tableEnsure(t, lk, scan_id, max_probe, &cr);
		ent_slot[i] = s;
		if (s == NONE) {
			atomicOr(&ctl->err, ERR_TABLE_FULL);
			continue;
		}
		while (cr) {
			u32 pos = atomicAdd(&ctl->n_new, 1u);
			if (pos < newcap) newlist[pos] = s;
			else atomicOr(&ctl->err, ERR_TABLE_FULL);
			if (1 == lk) {
				t.parent[s] = NONE;
				break;
			}
			u64 plk = lk >> 3;
			bool pcr;
			u32 ps = tableEnsure(t, plk, scan_id, max_probe, &pcr);
			if (ps == NONE) {
				atomicOr(&ctl->err, ERR_TABLE_FULL);
				break;
			}
			t.parent[s] = ps;
			atomicOr(&t.flags[ps], 1u << (16 + (u32)(lk & 7)));
			s = ps;
			lk = plk;
			cr = pcr;
		}
	}
}

// S2 init: children of a new block inherit the whole value of the node (createChildren,
// octree.h:1044-1054). The node's value is found in the first ancestor block that is not new.
__global__ __launch_bounds__(256) void k_init_new(Table t, MapGeom g, const u32* __restrict__ newlist, u32 newcap,
                                                  u32 scan_id, const ScanCtl* ctl)
{
	u32 n = min(ctl->n_new, newcap);
	if (ctl->err) return;
	for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		u32 s = newlist[i];
		u32 a = s;
		float v;
		u32 c = 0;
		for (;;) {
			u64 lk = t.keys[a];
			if (1 == lk) {
				v = t.root->occ;
				c = t.root->rgb;
				break;
			}
			u32 p = t.parent[a];
			if (t.stamp[p] != scan_id) {
				u32 ci = (u32)(lk & 7);
				v = t.occ[8 * (size_t)p + ci];
				if (g.color) c = t.rgb[8 * (size_t)p + ci];
				break;
			}
			a = p;
		}
		float4 vv = make_float4(v, v, v, v);
		float4* po = reinterpret_cast<float4*>(t.occ + 8 * (size_t)s);
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
		u32 f = t.flags[s];
		t.flags[s] = (f & F_INNER) | masks;
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
// S3 apply (level-1 blocks): hits with clamp, then misses with clamp (OMB:1351-1365, 1139-1145),
// followed by this block's own updateNode (OMB:1195-1224) and the hand-off to its parent.
// One thread per node block: the 8 children are one 32-byte record.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_apply_leaf(Table t, MapGeom g, const Entry* __restrict__ entries,
                                                    const u32* n_entries_p, const u32* __restrict__ ent_slot, float miss,
                                                    HitHash hh, const uint8_t* __restrict__ rgb_in, u32* __restrict__ wl,
                                                    ScanCtl* ctl)
{
	u32 n = *n_entries_p;
	if (ctl->err) return;
	for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		Entry e = entries[i];
		u32 s = ent_slot[i];
		float4* po = reinterpret_cast<float4*>(t.occ + 8 * (size_t)s);
		float4 a = po[0], b = po[1];
		float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
		const u64 pcode = (e.lk ^ (1ULL << (3 * (g.L - 1)))) << 3;  // depth-0 code of child 0
		float mid_max = 0;
		u32 mid_fl = 0;
		bool have_mid = false;
		if (e.hit) {
#pragma unroll
			for (int c = 0; c < 8; ++c) {
				if ((e.hit >> c) & 1) {
					if (g.color && rgb_in) {
						u32 hs = hitHashFind(hh, pcode | (u64)c);
						if (hs != NONE) {
							u32 pt = hh.minidx[hs];
							u32 upd = (u32)rgb_in[3 * (size_t)pt] | ((u32)rgb_in[3 * (size_t)pt + 1] << 8) |
							          ((u32)rgb_in[3 * (size_t)pt + 2] << 16);
							u32* pc = t.rgb + 8 * (size_t)s + c;
							*pc = blendColor(g, *pc, upd, v[c]);
						}
					}
					v[c] = clampAdd(v[c], g.hit, g.cmin, g.cmax);
				}
			}
			if (e.miss) {
				have_mid = true;
				mid_max = v[0];
#pragma unroll
				for (int c = 0; c < 8; ++c) {
					mid_max = fmaxf(mid_max, v[c]);
					mid_fl |= (isFreeV(g, v[c]) ? 1u : 0u) | (isUnknownV(g, v[c]) ? 2u : 0u);
				}
			}
		}
		if (e.miss) {
#pragma unroll
			for (int c = 0; c < 8; ++c)
				if ((e.miss >> c) & 1) v[c] = clampAdd(v[c], miss, g.cmin, g.cmax);
		}
		po[0] = make_float4(v[0], v[1], v[2], v[3]);
		po[1] = make_float4(v[4], v[5], v[6], v[7]);
		Summ sm = blockSummary(t, g, s, 1, 0);
		// what the parent's slot held before this scan touched the block
		u32 p = t.parent[s];
		u32 ci = (u32)(e.lk & 7);
		float old_occ = t.occ[8 * (size_t)p + ci];
		u32 fp = __hip_atomic_load(&t.flags[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		u32 old_fl = ((fp >> ci) & 1u) | (((fp >> (8 + ci)) & 1u) << 1);
		bool transient = have_mid && ((mid_max != old_occ || mid_fl != old_fl) || (mid_max != sm.occ || mid_fl != sm.fl));
		if (sm.collapsible) collapseBlock(t, s, e.lk);
		bool changed = writeToParent(t, g, s, e.lk, sm);
		if (changed) markDirty(t, p, 0, wl, &ctl->wl_count[0]);
		else if (transient) markDirty(t, p, F_TRANS, wl, &ctl->wl_count[0]);
	}
}

// updateAllChildren (OMB:1085-1120) on the subtree below block `s0` (level `lvl0`), iterative.
// Returns the bool the reference returns: something changed AND the node's summary changed.
__device__ inline bool subtreeApply(const Table& t, const MapGeom& g, u32 s0, u64 lk0, u32 lvl0, float u)
{
	u32 st_s[22];
	u64 st_lk[22];
	u32 st_i[22];
	bool st_ch[22];
	int sp = 0;
	st_s[0] = s0;
	st_lk[0] = lk0;
	st_i[0] = 0;
	st_ch[0] = false;
	u32 lvl = lvl0;
	bool ret = false;
	while (sp >= 0) {
		u32 s = st_s[sp];
		if (st_i[sp] < 8) {
			u32 i = st_i[sp]++;
			u32 f = t.flags[s];
			if (lvl > 1 && ((f >> (16 + i)) & 1u)) {
				u64 clk = (st_lk[sp] << 3) | (u64)i;
				u32 cs = tableFind(t, clk);
				if (cs == NONE) continue;  // cannot happen: inner bit implies a live block
				++sp;
				--lvl;
				st_s[sp] = cs;
				st_lk[sp] = clk;
				st_i[sp] = 0;
				st_ch[sp] = false;
			} else {
				float* pv = t.occ + 8 * (size_t)s + i;
				float v = *pv;
				float nv = clampAdd(v, u, g.cmin, g.cmax);
				if (nv != v) {
					*pv = nv;
					st_ch[sp] = true;
					if (lvl > 1) {
						// updateNode on a leaf inner node: flags from its own value (OMB:1181-1189)
						u32 nf = f & ~((1u << i) | (1u << (8 + i)));
						nf |= (isFreeV(g, nv) ? (1u << i) : 0u) | (isUnknownV(g, nv) ? (1u << (8 + i)) : 0u);
						if (nf != f) {
							// the top block's flags word is private to this thread too (its parent's is not)
							atomicAnd(&t.flags[s], nf | ~(F_CFREE | F_CUNK));
							atomicOr(&t.flags[s], nf & (F_CFREE | F_CUNK));
						}
					}
				}
			}
		} else {
			ret = false;
			if (st_ch[sp]) {
				Summ sm = blockSummary(t, g, s, lvl, t.flags[s]);
				if (sm.collapsible) collapseBlock(t, s, st_lk[sp]);
				ret = writeToParent(t, g, s, st_lk[sp], sm);
			}
			--sp;
			++lvl;
			if (sp >= 0 && ret) st_ch[sp] = true;
		}
	}
	return ret;
}

// S3c apply (blocks above level 1): misses at depth >= 1 (insert_depth > 0). A child that is a leaf
// is updated in place (OMB:1067-1072); a child that has been expanded gets the update on every leaf
// below it (OMB:1073-1079). The block itself is re-evaluated only when the reference would walk up
// to it: a leaf child's *flags* changed (OMB:1126-1133 starts at the leaf itself) or a subtree's
// summary changed.
__global__ __launch_bounds__(256) void k_apply_coarse(Table t, MapGeom g, const Entry* __restrict__ entries,
                                                      const u32* n_entries_p, const u32* __restrict__ ent_slot, float miss,
                                                      u32* __restrict__ wl, ScanCtl* ctl)
{
	u32 n = *n_entries_p;
	if (ctl->err) return;
	for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		Entry e = entries[i];
		u32 s = ent_slot[i];
		u32 level = e.level;
		bool evaluate = false;
		for (u32 c = 0; c < 8; ++c) {
			if (!((e.miss >> c) & 1)) continue;
			u32 f = t.flags[s];
			if ((f >> (16 + c)) & 1u) {
				u64 clk = (e.lk << 3) | (u64)c;
				u32 cs = tableFind(t, clk);
				if (cs != NONE && subtreeApply(t, g, cs, clk, level - 1, miss)) evaluate = true;
			} else {
				float* pv = t.occ + 8 * (size_t)s + c;
				float v = *pv;
				float nv = clampAdd(v, miss, g.cmin, g.cmax);
				*pv = nv;
				u32 nbits = (isFreeV(g, nv) ? (1u << c) : 0u) | (isUnknownV(g, nv) ? (1u << (8 + c)) : 0u);
				u32 obits = f & ((1u << c) | (1u << (8 + c)));
				if (nbits != obits) {
					evaluate = true;
					t.flags[s] = (f & ~((1u << c) | (1u << (8 + c)))) | nbits;
				}
			}
		}
		if (!evaluate) continue;
		Summ sm = blockSummary(t, g, s, level, t.flags[s]);
		if (sm.collapsible) collapseBlock(t, s, e.lk);
		bool changed = writeToParent(t, g, s, e.lk, sm);
		if (changed && 1 != e.lk) markDirty(t, t.parent[s], 0, wl, &ctl->wl_count[0]);
	}
}

// ------------------------------------------------------------------------------------------------
// P propagate: one level of updateParents (OMB:1126-1133). wl_in holds blocks whose children
// changed; a block whose own summary changes queues its parent for the next launch.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_propagate(Table t, MapGeom g, const u32* __restrict__ wl_in, u32* __restrict__ wl_out,
                                                   u32 in_idx, ScanCtl* ctl)
{
	u32 n = ctl->wl_count[in_idx];
	if (ctl->err) return;
	for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		u32 s = wl_in[i];
		u32 old = atomicAnd(&t.flags[s], ~(F_DIRTY | F_TRANS));
		u64 lk = t.keys[s];
		u32 level = levelOf(g, lk);
		Summ sm = blockSummary(t, g, s, level, old);
		if (sm.collapsible) collapseBlock(t, s, lk);
		bool changed = writeToParent(t, g, s, lk, sm);
		if (1 == lk) continue;
		if (changed) markDirty(t, t.parent[s], 0, wl_out, &ctl->wl_count[in_idx ^ 1]);
		else if (old & F_TRANS) markDirty(t, t.parent[s], F_TRANS, wl_out, &ctl->wl_count[in_idx ^ 1]);
	}
}

__global__ void k_reset_wl(ScanCtl* ctl, u32 idx) { ctl->wl_count[idx] = 0; }

// ------------------------------------------------------------------------------------------------
// read-back
// ------------------------------------------------------------------------------------------------
struct DumpCtl {
	unsigned long long n_out;
	unsigned long long n_live;
	unsigned long long n_leaf;
};

__global__ __launch_bounds__(256) void k_export_leaves(Table t, MapGeom g, int include_unknown, u64* __restrict__ codes,
                                                       uint8_t* __restrict__ depths, float* __restrict__ occ,
                                                       u32* __restrict__ rgb, unsigned long long cap, DumpCtl* dc)
{
	u32 ncap = t.mask + 1;
	for (u32 s = blockIdx.x * blockDim.x + threadIdx.x; s < ncap; s += gridDim.x * blockDim.x) {
		u64 lk = t.keys[s];
		if (0 == lk) continue;
		u32 f = t.flags[s];
		if (f & F_DEAD) continue;
		u32 level = levelOf(g, lk);
		u64 p = lk ^ (1ULL << (3 * (g.L - level)));
		atomicAdd(&dc->n_live, 1ULL);
		for (u32 i = 0; i < 8; ++i) {
			if (level > 1 && ((f >> (16 + i)) & 1u)) continue;
			float v = t.occ[8 * (size_t)s + i];
			atomicAdd(&dc->n_leaf, 1ULL);
			if (!include_unknown && isUnknownV(g, v)) continue;
			unsigned long long pos = atomicAdd(&dc->n_out, 1ULL);
			if (pos < cap) {
				codes[pos] = (p << 3) | (u64)i;
				depths[pos] = (uint8_t)(level - 1);
				occ[pos] = v;
				rgb[pos] = t.rgb ? t.rgb[8 * (size_t)s + i] : 0u;
			}
		}
	}
}

__global__ __launch_bounds__(256) void k_export_inner(Table t, MapGeom g, u64* __restrict__ codes, uint8_t* __restrict__ depths,
                                                      float* __restrict__ occ, uint8_t* __restrict__ flags,
                                                      u32* __restrict__ rgb, unsigned long long cap, DumpCtl* dc)
{
	u32 ncap = t.mask + 1;
	for (u32 s = blockIdx.x * blockDim.x + threadIdx.x; s < ncap; s += gridDim.x * blockDim.x) {
		u64 lk = t.keys[s];
		if (0 == lk) continue;
		u32 f = t.flags[s];
		if (f & F_DEAD) continue;
		u32 level = levelOf(g, lk);
		unsigned long long pos = atomicAdd(&dc->n_out, 1ULL);
		if (pos >= cap) continue;
		codes[pos] = lk ^ (1ULL << (3 * (g.L - level)));
		depths[pos] = (uint8_t)level;
		if (1 == lk) {
			occ[pos] = t.root->occ;
			flags[pos] = (uint8_t)(t.root->flags & 3u);
			rgb[pos] = t.root->rgb;
		} else {
			u32 p = t.parent[s];
			u32 ci = (u32)(lk & 7);
			u32 fp = t.flags[p];
			occ[pos] = t.occ[8 * (size_t)p + ci];
			flags[pos] = (uint8_t)(((fp >> ci) & 1u) | (((fp >> (8 + ci)) & 1u) << 1));
			rgb[pos] = t.rgb ? t.rgb[8 * (size_t)p + ci] : 0u;
		}
	}
}

// ------------------------------------------------------------------------------------------------
// table growth: re-insert every block into a table of twice/four times the capacity
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rehash_copy(Table src, Table dst, u32* fail)
{
	u32 ncap = src.mask + 1;
	for (u32 s = blockIdx.x * blockDim.x + threadIdx.x; s < ncap; s += gridDim.x * blockDim.x) {
		u64 lk = src.keys[s];
		if (0 == lk) continue;
		u32 d = hash64(lk) & dst.mask;
		bool ok = false;
		for (u32 probe = 0; probe <= dst.mask; ++probe) {
			u64 prev = atomicCAS((unsigned long long*)&dst.keys[d], 0ULL, (unsigned long long)lk);
			if (prev == 0) {
				ok = true;
				break;
			}
			d = (d + 1) & dst.mask;
		}
		if (!ok) {
			atomicOr(fail, 1u);
			continue;
		}
		const float4* so = reinterpret_cast<const float4*>(src.occ + 8 * (size_t)s);
		float4* dofs = reinterpret_cast<float4*>(dst.occ + 8 * (size_t)d);
		dofs[0] = so[0];
		dofs[1] = so[1];
		if (src.rgb) {
			const uint4* sc = reinterpret_cast<const uint4*>(src.rgb + 8 * (size_t)s);
			uint4* dcl = reinterpret_cast<uint4*>(dst.rgb + 8 * (size_t)d);
			dcl[0] = sc[0];
			dcl[1] = sc[1];
		}
		dst.flags[d] = src.flags[s];
		dst.stamp[d] = src.stamp[s];
	}
}
__global__ __launch_bounds__(256) void k_rehash_parents(Table dst)
{
	u32 ncap = dst.mask + 1;
	for (u32 s = blockIdx.x * blockDim.x + threadIdx.x; s < ncap; s += gridDim.x * blockDim.x) {
		u64 lk = dst.keys[s];
		if (0 == lk) continue;
		dst.parent[s] = (1 == lk) ? NONE : tableFind(dst, lk >> 3);
	}
}
}  // namespace ufo
