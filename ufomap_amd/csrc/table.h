// table.h -- the GPU-resident linear-hashed octree ("node table") and its device accessors.
//
// Replaces the reference's pointer octree (map/octree.h:957-1162: getNodePath/createNode/
// createChildren/deleteChildren with new/delete of 8-child arrays). One table slot = one 8-child
// node block, i.e. the `children` array of one inner node:
//
// (one array per field, see UFO_SLOT_BYTES below)
//   key        u64   location key of the inner node:  (1 << 3*(L-d)) | (code >> 3*d), d = node depth
//                    (sentinel bit encodes the depth; root = 1; 0 = empty slot)
//   occ[i]     f32   log-odds of child i (child index = x | y<<1 | z<<2, map/code.h:245-248)
//   rgb[8*s+i] u32   r | g<<8 | b<<16 of child i (colour maps only)
//   flags      u32   bits 0-7  contains_free of child i      (map/occupancy_map_node.h:171-176)
//                    bits 8-15 contains_unknown of child i
//                    bits 16-23 child i is an inner node with a live block of its own
//                    bit 24 DIRTY (queued for propagation), 25 DEAD (collapsed: the node is a leaf
//                    again, octree.h:1060-1066), 26 SUB (changed during a coarse-miss subtree update)
//   parent     u32   slot of the block that holds this node's own value (NONE for the root)
//   stamp      u32   id of the phase that created / revived the block ("new this phase")
//   tmax[s]    u64   (phase tag << 40) | (time of the last update beneath this node << 3) | child index
//                    that update lies under ("time": point index for hits = cloud order, 0 for misses =
//                    ascending code order; see map_kernels.h "last-update chain")
//   lu_*[s]          last-update record of the node for the current phase: did its last update reach it
//                    and change its summary, and the summary it had just before that update
//
// A node's own value lives in its parent's block; the root's value lives in MapRoot.
// Open addressing, linear probing, ANY capacity (the home slot is hash * capacity >> 32: no power of two needed, so a table
// is as large as its blocks need and not up to twice that) -- since round 4 TILE-MAJOR, see struct Table; blocks are never removed (a collapsed
// block is marked DEAD and revived by inheritance when a later update descends through it, which
// is exactly what the reference does with pruning disabled, octree.h:1064).
#pragma once
#include "geom.h"

namespace ufo
{
enum : u32 {
	F_CFREE = 0x000000FFu,
	F_CUNK = 0x0000FF00u,
	F_INNER = 0x00FF0000u,
	F_DIRTY = 1u << 24,
	F_DEAD = 1u << 25,
	F_SUB = 1u << 26,  // coarse-miss phase: a leaf child or a child's summary changed, re-evaluate bottom-up
	NONE = 0xFFFFFFFFu,
};

struct MapRoot {
	float occ;
	u32 flags;  // bit0 contains_free, bit1 contains_unknown
	u32 rgb;
	u32 used;  // number of occupied table slots
	u32 n_changes;     // change detection (occupancy_map_base.h:779-791): records appended to the change log so far
	u32 chg_overflow;  // a record did not fit the log (the host sizes it for the worst case: must stay 0)
	u32 pad[2];
};

// Change log: while change detection is enabled every leaf update that changes a value appends the code the reference
// inserts into `changes_` (occupancy_map_base.h:1070-1072, 1094-1108) as (1 << 3*(L-depth)) | (code >> 3*depth) -- the code's
// 3*(L-depth) significant bits behind a sentinel bit whose position tells the depth (the form of the table's location
// keys): 64 bits hold it for every depth_levels the map accepts (21: 63 bits + sentinel), where a depth field in the top
// bits would collide with the code from depth_levels = 20 on. The host decodes, sorts and de-duplicates when the set is
// read (ufomap_map_changes). buf == nullptr: disabled.
struct ChangeLog {
	u64* buf;
	u32 cap;
	u32 L;
};

// One table slot = one 8-child node block. Since round 5 the fields of a slot live in ARRAYS OF THEIR OWN (occ 32 B, key 8 B,
// flags / stamp / parent 4 B each: 52 B per slot; round 2-4: one 64-byte record per slot). The tree update of the tile paths
// (k_tile) is the bandwidth-bound kernel of the one bandwidth-bound configuration (a 2 mm RGB-D frame: 6.6e7 blocks per scan),
// and a tile's 73 slots are consecutive: with arrays the wave reads 44 B per block (values, key, flags) and writes back the
// 36 B it changed (values, flags) -- 80 B instead of 128 B per block, every line fully used; the parent link and the stamp
// are touched when a block is created and by the general path only. The general path's random look-ups pay one more line
// per touch (the key, then values + flags in flight together); its cost is launches and atomics, not lines.
#define UFO_SLOT_BYTES 52u
// TILE-MAJOR (round 4). The node blocks beneath one depth-3 node -- its level-3 block, the 8 level-2 and the 64 level-1 blocks:
// a TILE of 8x8x8 voxels, what one wavefront of the tiled tree update works on -- lie in 73 consecutive slots (4 672 bytes of
// records), found through a directory of tile keys: ONE hashed probe per tile instead of 73, and the records of a tile are
// contiguous in HBM instead of 73 random 64-byte lines (a 2 mm RGB-D frame touches 9e5 tiles = 6.6e7 blocks: its tree update
// ran at 1.3 TB/s). Slot numbers stay what every kernel works with:
//   slots [0, capU)                    node blocks of levels >= 4 (a few per cent of a map), open addressing as before
//   slots capU + 73 g + j, g < nG      group g: j < 64 the level-1 block whose 6-bit position inside the tile is j (lk & 63),
//                                      j = 64 + c the level-2 block c (lk & 7), j = 72 the level-3 block
//   gdir[g]                            location key of the level-3 node group g belongs to (0: free) -- double hashing over
//                                      the directory (8 bytes per tile: it lives in the L2s); the number of groups is a prime
// A block "exists" if its slot's key is set (a claimed group alone creates nothing). Maps with fewer than four levels keep
// everything in the first region.
#define UFO_GROUP 73u
// (128 counters in 512 consecutive bytes are four cache lines: 9e5 atomics of a fresh 2 mm frame's walk on them queued at ~12 ns
// each whichever of a line's words they named -- 2.6 of that walk's 4.5 ms. A line per counter.)
#define UFO_GCNT_STRIDE 32u
#define UFO_GCNT_WORDS (128u * UFO_GCNT_STRIDE)
struct Table {
	float* occA;   // [8 * slot + child]: log-odds of the 8 children (32 B per slot, float4-aligned)
	u64* keyA;     // [slot]: location key, 0 = empty slot
	u32* flagsA;   // [slot]
	u32* stampA;   // [slot]
	u32* parentA;  // [slot]
	u32* rgb;  // [8*slot + child], colour maps only (nullptr otherwise)
	u64* tmax;
	float* lu_occ;  // [slot]: the block's last-update record (map_kernels.h: publishLast); level-1 blocks park the old value of their last-updated voxel here first
	u32* lu_fl;     // bit0/1 = contains_free/unknown of the pre-last summary, bit 8 = "reached and changed", 9.. = phase tag
	u32* lu_rgb;    // colour maps only
	MapRoot* root;
	u32 mask;   // slots - 1 (ALL slots of both regions: the kernels that visit every slot)
	u32 capU;   // slots of the first region
	u32 nG;     // groups
	u64* gdir;  // [nG]
	u32* gcnt;  // 128 sharded counters, UFO_GCNT_STRIDE words apart: [0, 64) groups claimed, [64, 128) blocks created in the first region
	u32 L;      // depth levels of the map (a key's level = L - position of its sentinel bit / 3)
	__device__ __forceinline__ u64& key(u32 s) const { return keyA[s]; }
	__device__ __forceinline__ float* occ(u32 s) const { return occA + 8 * (size_t)s; }
	__device__ __forceinline__ u32& flags(u32 s) const { return flagsA[s]; }
	__device__ __forceinline__ u32& parent(u32 s) const { return parentA[s]; }
	__device__ __forceinline__ u32& stamp(u32 s) const { return stampA[s]; }
};

__device__ inline u32 hash64(u64 k)
{
	k ^= k >> 33;
	k *= 0xff51afd7ed558ccdULL;
	k ^= k >> 33;
	k *= 0xc4ceb9fe1a85ec53ULL;
	k ^= k >> 33;
	return (u32)k;
}

// level of the node block a location key names
__device__ __forceinline__ u32 keyLevel(const Table& t, u64 lk) { return t.L - (63u - (u32)__clzll((long long)lk)) / 3u; }
// a block of levels 1..3: its tile's key and its place inside the tile's group
__device__ __forceinline__ void tilePlace(u64 lk, u32 lvl, u64* lk3, u32* j)
{
	*lk3 = lk >> (3u * (3u - lvl));
	*j = (3u == lvl) ? 72u : ((2u == lvl) ? 64u + (u32)(lk & 7) : (u32)(lk & 63));
}
__device__ __forceinline__ bool inGroups(const Table& t, u32 lvl) { return lvl <= 3u && t.L >= 4u; }

// The directory is probed by DOUBLE hashing: it is kept up to 85 % full (a group is 73 slots: every spare directory entry is
// 5.8 KB of table), where linear probing needs 3.8 probes to find a key and 22 to find that it is not there; with a second
// hash as the stride -- the number of groups is a prime, so every stride visits every entry -- 2.2 and 6.7.
__device__ __forceinline__ void groupProbe(const Table& t, u64 lk3, u32* home, u32* stride, u32* h_out)
{
	u64 k = lk3;
	k ^= k >> 33;
	k *= 0xff51afd7ed558ccdULL;
	k ^= k >> 33;
	k *= 0xc4ceb9fe1a85ec53ULL;
	k ^= k >> 33;
	const u32 h = (u32)k, h2 = (u32)(k >> 32) ^ 0x9E3779B9u;
	*home = (u32)(((u64)h * (u64)t.nG) >> 32);
	*stride = 1u + (u32)(((u64)h2 * (u64)(t.nG - 1u)) >> 32);  // in [1, nG - 1]
	*h_out = h;
}
// group of a tile; NONE when the tile has none
__device__ inline u32 groupFind(const Table& t, u64 lk3)
{
	u32 g, st, h;
	groupProbe(t, lk3, &g, &st, &h);
	for (u32 probe = 0; probe < t.nG; ++probe) {
		const u64 k = __hip_atomic_load(&t.gdir[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (k == lk3) return g;
		if (k == 0) return NONE;
		g += st;
		if (g >= t.nG) g -= t.nG;
	}
	return NONE;
}
// ... found or claimed; NONE when the directory is full
__device__ inline u32 groupEnsure(const Table& t, u64 lk3)
{
	u32 g, st, h;
	groupProbe(t, lk3, &g, &st, &h);
	for (u32 probe = 0; probe < t.nG; ++probe) {
		u64 k = __hip_atomic_load(&t.gdir[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (k == 0) {
			const u64 prev = atomicCAS((unsigned long long*)&t.gdir[g], 0ULL, (unsigned long long)lk3);
			if (prev == 0) {
				atomicAdd(&t.gcnt[(h & 63u) * UFO_GCNT_STRIDE], 1u);  // (sharded: every new tile of a scan passes here)
				return g;
			}
			k = prev;
		}
		if (k == lk3) return g;
		g += st;
		if (g >= t.nG) g -= t.nG;
	}
	return NONE;
}
__device__ __forceinline__ u32 groupSlot(const Table& t, u32 g, u32 j) { return t.capU + UFO_GROUP * g + j; }

// Lookup only. Returns NONE when absent (DEAD blocks are returned: callers check flags).
__device__ inline u32 tableFind(const Table& t, u64 lk)
{
	const u32 lvl = keyLevel(t, lk);
	if (inGroups(t, lvl)) {
		u64 lk3;
		u32 j;
		tilePlace(lk, lvl, &lk3, &j);
		const u32 g = groupFind(t, lk3);
		if (g == NONE) return NONE;
		const u32 s = groupSlot(t, g, j);
		return __hip_atomic_load(&t.key(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == lk ? s : NONE;
	}
	u32 s = (u32)(((u64)hash64(lk) * (u64)t.capU) >> 32);
	for (u32 probe = 0; probe < t.capU; ++probe) {
		u64 k = __hip_atomic_load(&t.key(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (k == lk) return s;
		if (k == 0) return NONE;
		s = (s + 1u == t.capU) ? 0u : s + 1u;
	}
	return NONE;
}

// The block of child c of the node block in slot s (key lk): inside a tile group the child's slot follows from the parent's --
// no hashing, one load (the level-synchronous walks down a subtree and the serialiser look children up by the million).
__device__ inline u32 tableFindChild(const Table& t, u32 s, u64 lk, u32 c)
{
	const u64 ck = (lk << 3) | (u64)c;
	if (s >= t.capU && t.L >= 4u) {
		const u32 r = s - t.capU, g = r / UFO_GROUP, j = r - g * UFO_GROUP;
		if (j >= 64u) {  // (a level-1 block has no child blocks)
			const u32 cs = groupSlot(t, g, 72u == j ? 64u + c : (j - 64u) * 8u + c);
			return __hip_atomic_load(&t.key(cs), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ck ? cs : NONE;
		}
	}
	return tableFind(t, ck);
}

// a slot's key is there, or becomes the caller's: the find-or-create step both regions share
__device__ __forceinline__ bool slotClaim(const Table& t, u32 s, u64 lk, u64* seen)
{
	u64 k = __hip_atomic_load(&t.key(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if (k == 0) {
		const u64 prev = atomicCAS((unsigned long long*)&t.key(s), 0ULL, (unsigned long long)lk);
		if (prev == 0) {
			*seen = lk;
			return true;
		}
		k = prev;
	}
	*seen = k;
	return false;
}
__device__ __forceinline__ u32 reviveIfDead(const Table& t, u32 s, u32 scan_id, bool* created)
{
	u32 f = __hip_atomic_load(&t.flags(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if (f & F_DEAD) {
		u32 old = atomicAnd(&t.flags(s), ~F_DEAD);
		if (old & F_DEAD) {
			t.stamp(s) = scan_id;
			*created = true;
		}
	}
	return s;
}

// Find or insert/revive. *created = 1 when this thread created the block or revived a DEAD one.
// Returns NONE when the table is full (caller raises the capacity error).
__device__ inline u32 tableEnsure(const Table& t, u64 lk, u32 scan_id, u32 max_probe, bool* created, u32* n_created)
{
	*created = false;
	const u32 lvl = keyLevel(t, lk);
	u64 seen;
	if (inGroups(t, lvl)) {
		u64 lk3;
		u32 j;
		tilePlace(lk, lvl, &lk3, &j);
		const u32 g = groupEnsure(t, lk3);
		if (g == NONE) return NONE;
		const u32 s = groupSlot(t, g, j);
		if (slotClaim(t, s, lk, &seen)) {
			// empty slots have flags == 0 (table is zero-filled and never shrinks)
			t.stamp(s) = scan_id;
			++*n_created;  // the caller adds these to MapRoot::used once per wave
			*created = true;
			return s;
		}
		return reviveIfDead(t, s, scan_id, created);  // (seen == lk: the slot belongs to this key alone)
	}
	const u32 h = hash64(lk);
	u32 s = (u32)(((u64)h * (u64)t.capU) >> 32);
	const u32 lim = min(max_probe, t.capU);
	for (u32 probe = 0; probe < lim; ++probe) {
		if (slotClaim(t, s, lk, &seen)) {
			t.stamp(s) = scan_id;
			++*n_created;
			*created = true;
			atomicAdd(&t.gcnt[(64u + (h & 63u)) * UFO_GCNT_STRIDE], 1u);
			return s;
		}
		if (seen == lk) return reviveIfDead(t, s, scan_id, created);
		s = (s + 1u == t.capU) ? 0u : s + 1u;
	}
	return NONE;
}
// Re-hash: the block with key lk into a table that does not hold it; its slot, NONE when there is no room.
__device__ inline u32 tableInsertNew(const Table& t, u64 lk)
{
	const u32 lvl = keyLevel(t, lk);
	u64 seen;
	if (inGroups(t, lvl)) {
		u64 lk3;
		u32 j;
		tilePlace(lk, lvl, &lk3, &j);
		const u32 g = groupEnsure(t, lk3);
		if (g == NONE) return NONE;
		const u32 s = groupSlot(t, g, j);
		return slotClaim(t, s, lk, &seen) ? s : NONE;
	}
	const u32 h = hash64(lk);
	u32 s = (u32)(((u64)h * (u64)t.capU) >> 32);
	for (u32 probe = 0; probe < t.capU; ++probe) {
		if (slotClaim(t, s, lk, &seen)) {
			atomicAdd(&t.gcnt[(64u + (h & 63u)) * UFO_GCNT_STRIDE], 1u);
			return s;
		}
		s = (s + 1u == t.capU) ? 0u : s + 1u;
	}
	return NONE;
}
// the sharded counters, by one wave: groups claimed, blocks of the first region
__device__ inline void tableCounts(const Table& t, u32 lane, u32* groups, u32* upper)
{
	u32 a = __hip_atomic_load(&t.gcnt[(lane & 63u) * UFO_GCNT_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
	    b = __hip_atomic_load(&t.gcnt[(64u + (lane & 63u)) * UFO_GCNT_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	for (int o = 32; o > 0; o >>= 1) {
		a += __shfl_xor(a, o);
		b += __shfl_xor(b, o);
	}
	*groups = a;
	*upper = b;
}
// children `mask` of the node whose children have codes base | c at `depth`
__device__ inline void logChanges(const Table& t, const ChangeLog& cl, u64 base, u32 depth, u32 mask)
{
	if (!cl.buf || 0 == mask) return;
	u32 pos = atomicAdd(&t.root->n_changes, (u32)__popc(mask));
	while (mask) {
		const u32 c = (u32)__ffs(mask) - 1u;
		mask &= mask - 1u;
		if (pos < cl.cap) cl.buf[pos] = (base | (u64)c) | (1ULL << (3u * (cl.L - depth)));
		else t.root->chg_overflow = 1u;
		++pos;
	}
}
__device__ inline void logChange(const Table& t, const ChangeLog& cl, u64 code_shifted, u32 depth)
{
	if (!cl.buf) return;
	const u32 pos = atomicAdd(&t.root->n_changes, 1u);
	if (pos < cl.cap) cl.buf[pos] = code_shifted | (1ULL << (3u * (cl.L - depth)));
	else t.root->chg_overflow = 1u;
}
}  // namespace ufo
