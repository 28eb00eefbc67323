// table.h -- the GPU-resident linear-hashed octree ("node table") and its device accessors.
//
// Replaces the reference's pointer octree (map/octree.h:957-1162: getNodePath/createNode/
// createChildren/deleteChildren with new/delete of 8-child arrays). One table slot = one 8-child
// node block, i.e. the `children` array of one inner node:
//
// (struct Block below; rgb and the last-update records are separate arrays)
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
// is as large as its blocks need and not up to twice that); blocks are never removed (a collapsed
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

// One table slot: the record of one 8-child node block, 64 bytes, 64-byte aligned -- what a lookup finds (the key),
// what an update reads and writes (the 8 child values) and what the walk up needs (flags, parent) arrive with ONE
// memory transaction. (Round 1 kept these as separate arrays: one block touch pulled five cache lines.)
struct alignas(64) Block {
	float occ[8];  // log-odds of the 8 children; 16-byte aligned for float4 access
	u64 key;       // location key, 0 = empty slot
	u32 flags;
	u32 parent;
	u32 stamp;
	u32 pad[3];
};
static_assert(sizeof(Block) == 64, "Block must be one 64-byte record");

struct Table {
	Block* blk;
	u32* rgb;  // [8*slot + child], colour maps only (nullptr otherwise)
	u64* tmax;
	float* lu_occ;  // [slot]: the block's last-update record (map_kernels.h: publishLast); level-1 blocks park the old value of their last-updated voxel here first
	u32* lu_fl;     // bit0/1 = contains_free/unknown of the pre-last summary, bit 8 = "reached and changed", 9.. = phase tag
	u32* lu_rgb;    // colour maps only
	MapRoot* root;
	u32 mask;  // capacity - 1 (the capacity need not be a power of two: tableHome / tableNext)
	__device__ __forceinline__ u64& key(u32 s) const { return blk[s].key; }
	__device__ __forceinline__ float* occ(u32 s) const { return blk[s].occ; }
	__device__ __forceinline__ u32& flags(u32 s) const { return blk[s].flags; }
	__device__ __forceinline__ u32& parent(u32 s) const { return blk[s].parent; }
	__device__ __forceinline__ u32& stamp(u32 s) const { return blk[s].stamp; }
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

// home slot of a key and the probe sequence's next slot, for any capacity
__device__ __forceinline__ u32 tableHome(const Table& t, u64 lk) { return (u32)(((u64)hash64(lk) * ((u64)t.mask + 1ull)) >> 32); }
__device__ __forceinline__ u32 tableNext(const Table& t, u32 s) { return s == t.mask ? 0u : s + 1u; }

// Lookup only. Returns NONE when absent (DEAD blocks are returned: callers check flags).
__device__ inline u32 tableFind(const Table& t, u64 lk)
{
	u32 s = tableHome(t, lk);
	for (u32 probe = 0; probe <= t.mask; ++probe) {
		u64 k = __hip_atomic_load(&t.key(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (k == lk) return s;
		if (k == 0) return NONE;
		s = tableNext(t, s);
	}
	return NONE;
}

// Find or insert/revive. *created = 1 when this thread created the block or revived a DEAD one.
// Returns NONE when the table is full (caller raises the capacity error).
__device__ inline u32 tableEnsure(const Table& t, u64 lk, u32 scan_id, u32 max_probe, bool* created, u32* n_created)
{
	u32 s = tableHome(t, lk);
	*created = false;
	for (u32 probe = 0; probe < max_probe; ++probe) {
		u64 k = __hip_atomic_load(&t.key(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (k == 0) {
			u64 prev = atomicCAS((unsigned long long*)&t.key(s), 0ULL, (unsigned long long)lk);
			if (prev == 0) {
				// empty slots have flags == 0 (table is zero-filled and never shrinks)
				t.stamp(s) = scan_id;
				++*n_created;  // the caller adds these to MapRoot::used once per wave
				*created = true;
				return s;
			}
			k = prev;
		}
		if (k == lk) {
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
		s = tableNext(t, s);
	}
	return NONE;
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
