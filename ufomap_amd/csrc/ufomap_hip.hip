// ufomap_hip.hip -- C-ABI implementation (include/ufomap_hip.h) over the HIP kernels of
// scan_kernels.h / map_kernels.h / fast_kernels.h. gfx950 only; build: see ufomap_amd/build.py. ONE translation unit; three
// parts of it live in files of their own, included where they used to stand: host_fast_path.inl (the steady-state path's
// host side), host_multi_gpu.inl (ufomap_comm_*, ufomap_map_insert_batch), host_serialise.inl (write / read of the byte stream)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
//
// Launch sequence of one depth-0 integration in the steady state (doInsert; fast_kernels.h):
//   prep stream: [H2D of a host cloud] -> k_fhits -> k_signal
//   scan stream: k_done_gate (the descriptor and number of the scan half BEFORE become visible to the walks; then the gate
//                of this one: waits for k_signal) -> k_fcast [a grid beyond LDS: memsets -> k_fselect -> k_cast<2>]
//   map stream:  k_claim (waits for the scan half; takes every scan that is ready along) -> k_fmerge -> k_tile (looks at
//                the predecessor's status) -> k_ftail (stores the finished control blocks and the scans' numbers into
//                pinned host memory) [a grid beyond LDS: k_up between k_tile and k_ftail] -- ONE slot for all the scans it
//                takes; no slot of its own for a scan while two are waiting. A synchronous call with nothing in flight:
//                k_fhits -> k_fcast -> k_fmerge -> k_tile -> k_ftail on the map stream alone.
//   join (of integrations that have completed): the host polls those words; no copy, no stream synchronisation
// First scans, insert depth > 0, simple ray casting, > 1022 cells per axis (the general path):
//   scan stream: memset hit hash -> control block H2D -> k_classify -> k_select -> k_reduce_boxes (checks the
//                predicted ray grid) -> k_hitmark -> k_cast<0|2> -> [k_merge_slabs] -> k_extract_bits -> k_extract_hits
//   map stream:  [waits for the scan's event] k_ensure -> k_init_new -> k_apply_leaf -> k_propagate x wide levels ->
//                k_propagate_tail; join: event wait -> one control-block D2H on the read-back stream
// Without a predicted grid the boxes are read back after k_select (scanPhase); grids of > 1023 cells per axis and
// simple ray casting use k_dda, boxes beyond the scratch limit k_dda_set; insert depth > 0 adds the k_coarse_* phase
// and walks the tree twice.
// There is no CPU fallback: every entry point fails with UFOMAP_ERR_DEVICE when HIP is unusable.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <sstream>
#include <chrono>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/ufomap_hip.h"
#include "vol_kernels.h"

using namespace ufo;

namespace
{
thread_local std::string g_err;

int fail(int code, const std::string& msg)
{
	g_err = msg;
	return code;
}

#define HIP_TRY(expr)                                                                              \
	do {                                                                                           \
		hipError_t e__ = (expr);                                                                   \
		if (e__ != hipSuccess) {                                                                   \
			return fail(UFOMAP_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__));    \
		}                                                                                          \
	} while (0)

// Process-wide counts of device allocations (ufomap_alloc_counters): a warm scan must not allocate -- hipMalloc / hipFree
// synchronise the device and take as long as the driver's page tables need (the one leg whose time moved 7 x
// between boxes reports how many it saw inside every timed call).
std::atomic<uint64_t> g_n_malloc{0}, g_n_free{0}, g_bytes_malloc{0}, g_ns_alloc{0}, g_n_rehash{0};

// grow-only device buffer
// device allocation, move-only owner: released when it goes out of scope (every early return of the functions below)
struct DevBuf {
	void* p = nullptr;
	size_t cap = 0;
	DevBuf() = default;
	DevBuf(const DevBuf&) = delete;
	DevBuf& operator=(const DevBuf&) = delete;
	DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap)
	{
		o.p = nullptr;
		o.cap = 0;
	}
	DevBuf& operator=(DevBuf&& o) noexcept
	{
		if (this != &o) {
			release();
			p = o.p;
			cap = o.cap;
			o.p = nullptr;
			o.cap = 0;
		}
		return *this;
	}
	~DevBuf() { release(); }
	hipError_t reserve(size_t bytes)
	{
		if (bytes <= cap) return hipSuccess;
		const auto t0 = std::chrono::steady_clock::now();
		size_t want = std::max(bytes, cap + cap / 2);
		if (p) {
			hipError_t e = hipFree(p);
			g_n_free.fetch_add(1, std::memory_order_relaxed);
			if (e != hipSuccess) return e;
			p = nullptr;
			cap = 0;
		}
		// (large buffers in whole 2 MiB pieces: the driver maps them with 2 MiB fragments, and random 64-byte accesses to a
		// multi-gigabyte node table are then one TLB entry per 2 MiB instead of one per 4 KiB)
		want = want >= (1u << 21) ? ((want + (1u << 21) - 1) & ~(size_t)((1u << 21) - 1)) : ((want + 255) & ~(size_t)255);
		hipError_t e = hipMalloc(&p, want);
		g_n_malloc.fetch_add(1, std::memory_order_relaxed);
		g_bytes_malloc.fetch_add(want, std::memory_order_relaxed);
		g_ns_alloc.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed);
		if (e != hipSuccess) {
			p = nullptr;
			return e;
		}
		cap = want;
		// Test aid (UFOMAP_POISON=<byte>): new device memory is filled with that byte -- the driver hands out zeroed pages to a fresh
		// process, so a buffer that is read before it is written only shows in a process that has freed memory before (round 6:
		// scripts/dev/fuzz_api.py found one that way); with the poison it shows at once, in any test.
		static const int poison = [] {
			const char* v = getenv("UFOMAP_POISON");
			return v && *v ? (int)(strtol(v, nullptr, 0) & 0xFF) : -1;
		}();
		if (poison >= 0 && want <= (1ull << 32)) {
			e = hipMemset(p, poison, want);
			if (e == hipSuccess) e = hipDeviceSynchronize();  // (the fill runs on the null stream, which the handle's non-blocking streams do not wait for)
			if (e != hipSuccess) return e;
		}
		return hipSuccess;
	}
	void release()
	{
		if (p) {
			const auto t0 = std::chrono::steady_clock::now();
			(void)hipFree(p);
			g_n_free.fetch_add(1, std::memory_order_relaxed);
			g_ns_alloc.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed);
		}
		p = nullptr;
		cap = 0;
	}
	template <class T>
	T* as() const { return reinterpret_cast<T*>(p); }
};

struct KernelStat {
	const char* name;
	uint64_t launches = 0;
	double total_ms = 0;
};

struct PendingEvent {
	hipEvent_t a, b;
	int stat;
};

struct TableBufs {
	DevBuf blk, rgb, tmax, luocc, lufl, lurgb, gdir, gcnt;
	void release()
	{
		DevBuf* all[] = {&blk, &rgb, &tmax, &luocc, &lufl, &lurgb, &gdir, &gcnt};
		for (DevBuf* b : all) b->release();
	}
};

inline u32 nextPow2(u64 v)
{
	u64 p = 1;
	while (p < v) p <<= 1;
	return (u32)std::min<u64>(p, 1ull << 31);
}
// What an update can add to the node table, as its two regions count it (table.h): node blocks in all, tile groups (= new
// level-3 blocks), blocks of the first region (levels >= 4). The host keeps the table large enough for the upper bound of
// everything in flight: a walk that finds no free group or slot half-way cannot stand back (the volume path, which
// creates against a reserve, can).
struct Need {
	u64 blocks = 0, groups = 0, upper = 0;
	Need& operator+=(const Need& o)
	{
		blocks += o.blocks;
		groups += o.groups;
		upper += o.upper;
		return *this;
	}
};
inline Need operator+(Need a, const Need& b) { return a += b; }
inline Need needMin(const Need& a, const Need& b) { return Need{std::min(a.blocks, b.blocks), std::min(a.groups, b.groups), std::min(a.upper, b.upper)}; }
}  // namespace

// What the map half of scan i needs while the scan half of scan i+1 already runs on the other stream:
// double-buffered and swapped at the start of every scan (see doInsert).
// The arguments of an integration, kept with its hand-over set: a scan that was enqueued speculatively (on a grid
// predicted from the previous scan) and turns out not to fit is repeated from them when it is joined.
struct ScanArgs {
	bool spec = false;
	double origin[3] = {0, 0, 0};
	const double* d_xyz = nullptr;
	const uint8_t* d_rgb = nullptr;
	size_t n = 0;
	double max_range = -1;
	unsigned depth = 0;
	int discrete = 0, simple = 0;
	Ingest ing{};
};

constexpr int kAlt = 7;  // hand-over sets besides the current one: up to kAlt + 1 integrations in flight

struct HandOver {
	DevBuf b_ctl, b_entries, b_hh_keys, b_in_xyz, b_in_rgb;  // b_hh_keys: hit hash, keys followed by point indices
	// what the tree update of a fast-path scan (fast_kernels.h) reads after the scan half has moved on to the next scan
	DevBuf b_gridM, b_gridH, b_part1, b_hit_code, b_first, b_tilebits, b_slabs;
	uint64_t seq = 0;         // running number of the integration that uses this set
	bool first_dirty = true;  // b_first / b_tilebits are left clean by k_fcast / k_ftail unless the scan stood back
	bool fast = false;        // the integration that uses this set runs on the fast path
	uint64_t fseq = 0;        // ... its running number among the fast-path scans (Pipe: ring entry, slot, status word)
	bool deferred = false;    // ... and no slot on the map stream has been enqueued for it yet (the next slot will take it along)
	bool has_slot = false;    // ... a slot of its own has been enqueued for the scan
	DevBuf b_keep;            // the scan's points as its head loop saw them (written by k_fhits): what a repeat of the scan reads
	DevBuf b_keep_rgb;        // ... and their colours (colour maps: read by the tree update, fast_kernels.h: k_tile)
	// a step of ufomap_map_insert_batch on the fast path: this rank's scan, exchanged as bit grids, one walk for all ranks' scans
	int batch_world = 0;           // 0: not such a step; else the number of ranks
	struct ufomap_comm* comm = nullptr;
	DevBuf b_xsend, b_xrecv, b_bpipe;  // exchange slot of this rank, the gathered slots, the walk's own Pipe (fast_kernels.h)
	uint8_t* h_res_all = nullptr;  // pinned: the finished control blocks of the other ranks' scans (errors, boxes)
	int h_res_all_world = 0;
	hipEvent_t xchg_ev = nullptr;  // end of the all-gather on the scan stream
	bool hit_grid = false;    // the hits of the set's scan are in b_gridH (fast path), not in the hit list (ufomap_map_last_hits)
	FastGeo fgeo{};
	UpperGeo ugeo{};
	ScanCtl* h_ctl = nullptr;  // pinned
	ScanCtl* h_res = nullptr;  // pinned: the finished control block as k_ftail stored it
	unsigned long long* sig_prep = nullptr;  // signal memory: "first kernel done" of the set's scan
	bool done_by_flag = false; // the set's integration announces its end in the word behind h_res, not by done_ev
	bool ctl_clean = false;    // b_ctl holds the fast path's start state
	void* h_stage = nullptr;   // pinned staging of a pageable host cloud (ufomap_map_insert): filled by the host, drained by
	size_t h_stage_cap = 0;    // an asynchronous H2D copy on the scan stream; free again once the set's integration is joined
	u32 hh_mask = 0;
	uint64_t counts[8] = {0};
	ScanArgs args;
	hipEvent_t done_ev = nullptr;  // end of the integration that uses this set
	bool pending = false;          // that integration has been enqueued and not yet been joined
	Need bound;                    // upper bound of what it may add to the node table
	// the volume path (vol_kernels.h): what the tree update of the set's scan reads while the next scan's scan half runs (round 5:
	// an asynchronous call returns with the walk enqueued; the next call casts its rays meanwhile and joins it before its own walk)
	DevBuf b_vM, b_vMm, b_vH, b_vtb, b_vlist, b_vcopies, b_vslots, b_vaux, b_vupbits;  // (b_vM: eight copies, one per XCD)
	DevBuf b_vrec;           // the walk's tile records (a walk that ran out of its reserve is finished from them when it is joined)
	bool vol = false;        // the integration that uses this set runs on the volume path
	bool vol_dirty = true;   // the set's brick grids are not known to be all zero
	bool vol_walk = false;   // ... its walk has been enqueued and not been looked at (volWalkFinish: the reserve may have run out)
	u32 vol_count = 0;       // tiles the scan has listed
	u32 vol_scan_id = 0;     // the walk's number (tile records of a walk that is repeated after a table growth carry it)
	const uint8_t* vol_rgb = nullptr;  // the scan's colours (the caller's device array, or the set's own copy of it)
	VolPlan vplan{};
};

// Round 6: the helper that copies a pageable host cloud into the set's pinned staging buffer while the calling thread enqueues the
// scan (uploadCloud). One per handle, started by the first such call; it touches host memory only -- no HIP call is ever made from it.
struct StageWorker {
	std::thread th;
	std::mutex mu;
	std::condition_variable cv;
	std::atomic<bool> quit{false};
	bool sleeping = false;
	struct Job {
		void* dst[2];
		const void* src[2];
		size_t bytes[2];
		size_t chunk;  // the first range goes in pieces of this size, announced one by one: its H2D copy starts behind the first
	} job{};
	std::atomic<unsigned long long> posted{0}, done{0};
	// pinned, what k_host_gate polls: (job number << 8) | pieces of the job that have been copied -- the second range is the last piece
	volatile unsigned long long* flag = nullptr;
	static u32 pieces(const Job& j) { return (u32)((j.bytes[0] + j.chunk - 1) / j.chunk) + (j.bytes[1] ? 1u : 0u); }
	void run()
	{
		unsigned long long seen = 0;
		for (;;) {
			// a cloud a few tens of microseconds after the last one is the steady state: spin briefly, then sleep
			const auto t0 = std::chrono::steady_clock::now();
			while (posted.load(std::memory_order_acquire) == seen) {
				if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) {
					std::unique_lock<std::mutex> lk(mu);
					sleeping = true;
					cv.wait(lk, [&] { return quit.load() || posted.load(std::memory_order_acquire) != seen; });
					sleeping = false;
				}
			}
			if (quit.load()) return;
			seen = posted.load(std::memory_order_acquire);
			u32 piece = 0;
			for (size_t off = 0; off < job.bytes[0]; off += job.chunk) {
				memcpy(static_cast<char*>(job.dst[0]) + off, static_cast<const char*>(job.src[0]) + off, std::min(job.chunk, job.bytes[0] - off));
				std::atomic_thread_fence(std::memory_order_release);
				*flag = (seen << 8) | (unsigned long long)++piece;
			}
			if (job.bytes[1]) {
				memcpy(job.dst[1], job.src[1], job.bytes[1]);
				std::atomic_thread_fence(std::memory_order_release);
				*flag = (seen << 8) | (unsigned long long)++piece;
			}
			done.store(seen, std::memory_order_release);
		}
	}
	void post(const Job& j)
	{
		job = j;
		posted.fetch_add(1, std::memory_order_release);
		std::lock_guard<std::mutex> lk(mu);
		if (sleeping) cv.notify_one();
	}
	void wait()
	{
		const unsigned long long want = posted.load(std::memory_order_relaxed);
		while (done.load(std::memory_order_acquire) < want) {
		}
	}
	void stop()
	{
		if (!th.joinable()) return;
		{
			std::lock_guard<std::mutex> lk(mu);
			quit.store(true);
			cv.notify_one();
		}
		posted.fetch_add(1, std::memory_order_release);  // (a worker that is spinning leaves its loop and looks at `quit`)
		th.join();
	}
};

struct ufomap_map {
	int device = 0;
	hipStream_t stream = nullptr;   // map stream: everything that touches the node table
	hipStream_t sstream = nullptr;  // scan stream: classify .. extract of the NEXT scan overlaps the previous map phase
	hipStream_t pstream = nullptr;  // prep stream: H2D copy of a host cloud and the fast path's first kernel (k_fhits) -- neither depends
	                                // on the scan before, so they overlap its ray kernel instead of queueing behind it
	hipStream_t cs = nullptr;       // stream the helpers currently launch on
	hipEvent_t done_ev = nullptr, scan_ev = nullptr, prep_ev = nullptr;
	hipStream_t xstream = nullptr;  // read-back of control blocks whose producers are known to be complete
	hipStream_t gstream = nullptr;  // the all-gather of a batch step (host_multi_gpu.inl), created with the first such step
	hipEvent_t pack_ev = nullptr;   // ... its exchange slot has been packed on the scan stream
	bool prev_flagged = false;      // the integration joined last had flagged an error (finishPending)
	int opt_early = 1;              // enqueue the map half before the previous integration has been joined (doInsert)
	HandOver alt[kAlt];             // the other hand-over sets, in no particular order: integrations are told apart by `seq`
	Need bound;                     // (of the current set, see HandOver::bound)
	                  // the other set of hand-over buffers
	int async_status = UFOMAP_OK;   // first error of an integration that was joined by a later call
	MapGeom g{};
	double model_log[6] = {0, 0, 0, 0, 0, 0};  // toLogit of occupied_thres, free_thres, prob_hit, prob_miss, clamp min, clamp max (OMB:1536-1542)
	bool chg_enabled = false;                 // enableChangeDetection (OMB:783): leaf updates append to the change log
	DevBuf b_changes;                         // the log: u64 records (table.h: ChangeLog)
	u32 chg_cap = 0;
	bool minmax_enabled = true;               // enableMinMaxChangeDetection (OMB:791-797); on by default, as the server sets it
	// node table
	Table t{};
	TableBufs tb;
	DevBuf b_root;
	u32 scan_id = 0;
	u64 miss_set_slots = 0;  // Grid::layout 2: capacity of the sparse set of ray cells (grows by doubling, remembered)
	u32 miss_set_count = 0;  // ... node blocks in it after the last walk
	bool poisoned = false;  // an update overran the node table half-way (ctlError): only clear / destroy are accepted
	u32 phase_limit = (1u << 22) - (1u << 12);  // phaseGuard: tags are cleared and the numbering restarts here
	u64 n_phase_resets = 0;
	u64 used_est = 0;  // host-side view of MapRoot::used (refreshed at every control-block read)
	u64 used_g = 0, used_u = 0;  // ... of the groups claimed and the blocks of the first region (table.h)
	u64 hit_tiles = ~0ull;       // depth-3 nodes that hold a hit voxel of the current scan (k_select), if it has been read back
	// per-scan buffers
	DevBuf b_ctl, b_pt_end, b_pt_flag, b_pt_slot, b_ray_end, b_hit_code, b_hit_pt, b_hh_keys;
	DevBuf b_part0, b_part1, b_slabs, b_hb_keys, b_hb_mask, b_hb_time;  // (b_slabs: per hand-over set on the fast path, HandOver)
	DevBuf b_first, b_tilebits, b_gridH;  // fast path, per hand-over set (HandOver)
	UpperGeo ugeo{};
	DevBuf b_tilerec;             // fast path, map stream only
	DevBuf b_ser[6];              // scratch of the map byte stream (serialiseNodes), kept between calls
	uint8_t* h_ser = nullptr;     // ... pinned: per-level counts, the root, the stream's length
	DevBuf b_upbits;              // grids beyond LDS: which level-4 blocks the walk's k_up has evaluated (bitmap; left clean by k_ftail)
	DevBuf b_upguess;             // k_ftail: where the block of every cell above the tiles was when a walk last looked (guesses, checked by key)
	DevBuf b_ts;                  // developer aid (option "tstamps"): device clock at the pipeline's hand-overs (fast_kernels.h: Pipe::ts)
	DevBuf b_pipe;                // fast path: which walk applies which scan (fast_kernels.h: Pipe), device side
	uint64_t n_fseq = 0;          // fast-path scans enqueued so far
	uint64_t last_slot_fseq = 0;  // ... the newest of them whose slot on the map stream has been enqueued (enqueueSlot)
	u32 geo_id = 0;               // scans with the same geo id may share a walk: same ray grid, no other update of the map between them
	bool chain_ok = false;        // the update enqueued last on the map stream was a fast-path slot (with ray grid chain_geo)
	FastGeo chain_geo{};
	DevBuf b_blk_range;           // k_select: where each of its workgroups' rays lie in the ray list (k_cast<2>)
	bool first_dirty = true, fast = false, hit_grid = false, deferred = false, has_slot = false;  // (HandOver)
	int batch_world = 0;          // (HandOver)
	struct ufomap_comm* comm = nullptr;
	DevBuf b_xsend, b_xrecv, b_bpipe;
	uint8_t* h_res_all = nullptr;
	int h_res_all_world = 0;
	hipEvent_t xchg_ev = nullptr;
	DevBuf b_sig_xchg;           // ... or, with gates, the word a kernel behind the all-gather stores the step's number in (fastBatchStep)
	uint64_t fseq = 0;            // (HandOver)
	DevBuf b_keep, b_keep_rgb;    // (HandOver)
	unsigned long long* h_prep = nullptr;  // pinned: integration number of the newest scan whose k_fhits has finished (k_signal)
	uint64_t n_walks = 0, n_walk_scans = 0, n_gate_timeouts = 0;  // fast-path walks that applied scans, scans in them; stream hand-overs that timed out
	int opt_batch_max = 8;        // scans a walk may take when scans have queued up behind the map stream (1 = one walk per scan)
	u32 ser_tail_blocks = 0xFFFFFFFFu, ser_tail_first = 0;  // blocks in the serialiser's narrow levels as the last serialisation found them
	int n_cus = 256;              // compute units of the device (the ray kernel's LDS allows one workgroup on each)
	int opt_vol_fused = 1;        // setValueVolume of a small volume at min_depth 0: one launch for all levels
	int opt_ser_short = 1;        // serialisation without host round trips in the middle (maps up to 32 MiB of stream)
	uint8_t* h_out = nullptr;     // ... its pinned output buffer
	size_t h_out_cap = 0;
	int opt_big = 1;              // ray grids beyond LDS on the fast path (k_fselect / k_cast<2> / k_up); 0: the general path
	int opt_fast_color = 1;       // colour maps on the fast path (0: the general path, as before round 3)
	int opt_solo = 1;             // synchronous calls with nothing in flight run on the map stream alone
	bool solo = false;            // ... the current integration does
	int opt_cast_threads = 512;
	int opt_cast_batch = 256, opt_cast_qcap = 1024, opt_cast_prio = 0;  // k_fcast: rays per round, segment queue entries (LDS), wave priority
	int opt_lazy_done = 1;        // asynchronous fast-path calls: the end of scan half i is published by the gate kernel of scan i+1 (k_done_gate)
	ScanDesc sd_saved{};          // ... the descriptor kept back,
	bool sd_pending = false;      // ... if any
	// a batch step's own scan is merged into its exchange slot, which also gets the walk's descriptors (fastBatchStep -> fastScanPhase -> k_fmerge_batch)
	const DescPack* batch_pack = nullptr;
	u32 batch_B = 0;
	uint8_t* batch_send = nullptr;
	int opt_stage_pieces = 8;     // a pageable cloud reaches its pinned staging buffer -- and from there HBM -- in this many pieces (uploadCloud)
	int opt_batch_depth = 3;      // batch steps in flight, the one being enqueued included, before the oldest is joined (insert_batch: the same on
	                              // every rank). 2 until round 6: the host then enqueued step i only after the walk of step i - 2 -- i.e. when the scan
	                              // half of step i - 1 was ending -- and the scan stream idled for the time of the enqueue (0.078 -> 0.060 ms per step)
	int opt_hold = 0;             // test aid: a slot is enqueued for every hold-th scan only (walks over several scans whatever the timing)
	int opt_gate_us = 20000;      // a stream hand-over gives up after this long (and the handle stops using gates)
	uint64_t seq = 0, latest_seq = 0;  // seq: of the integration that uses the current set; latest_seq: of the newest one enqueued
	FastGeo fgeo{};
	int opt_fast = 1;  // 0 = never take the fast path (fast_kernels.h)
	// the volume path (vol_kernels.h, host_vol.inl): depth-0 scans whose ray grid is beyond the steady-state path's
	int opt_vol = 1;        // 0 = never; 2 = also for the ray grids the steady-state path would take (tests)
	int opt_vol_pregrow = 1;  // 0: no growth of the node table before a walk (tests: the walk runs out of its reserve)
	int opt_gather_stream = 0;  // batch steps: the all-gather on a stream of its own (host_multi_gpu.inl)
	int opt_fast_simple = 1; // simple (fixed-step) ray casting on the fast path (0: the general path)
	int opt_fail_scan = 0;  // test aid: the scan half of the next batch steps 'fails' on this rank (host_multi_gpu.inl)
	int opt_vol_mode = 0;   // measuring aid (k_vdda / k_vwalk): bit 1 blocks in launch order, bit 2 rays in the cloud's order, bit 3 no write-combining table, bit 4 one lane per ray (k_vdda)
	int opt_vol_seg = 192;  // cells per segment of a ray on the volume path (k_vcutA / k_vwalk)
	int opt_vol_walk_blocks = 1536;  // workgroups of k_vwalk per eighth of the scan
	int opt_vol_walk_lds = 0;  // extra LDS per workgroup of k_vwalk, bytes: caps its workgroups per CU (what is left takes the tree update of the scan before)
	int opt_fmerge_rows = 2;  // rows of workgroups of k_fmerge (a row takes every rows-th scan of the walk)
	int opt_wait_flush_first = 1;  // ufomap_map_wait enqueues the walk of the scans that wait for company before it synchronises anything
	int opt_vol_color = 1;  // colour maps on the volume path (0: the general path, as in round 4)
	int opt_vol_async = 1;  // an asynchronous call returns with the volume path's walk enqueued (0: every call returns a finished integration)
	int opt_vol_keep = 1;   // k_tile leaves the merged ray cells of its tiles behind (ufomap_map_last_misses)
	// (the current hand-over set's share of the volume path's state: HandOver)
	bool vol = false, vol_dirty = true, vol_walk = false;
	u64 n_fill_zero = 0;     // joined fast-path scans whose result block reported an empty table while the host knew better (must stay 0)
	bool keys_mode = false;  // scanPhase is run for an update list (ufomap_map_scan_keys, the list form of a batch step)
	u32 vol_count = 0, vol_scan_id = 0;
	const uint8_t* vol_rgb = nullptr;
	DevBuf b_vM, b_vMm, b_vH, b_vtb, b_vlist, b_vcopies, b_vslots, b_vaux, b_vupbits, b_vrec;
	VolPlan vplan{};
	// ... and what all sets share: the scan half's own scratch (scan halves run one after the other on the scan stream)
	DevBuf b_vbin, b_vrays, b_vsegs, b_vsegcnt;
	uint64_t n_vol = 0, n_vol_grow = 0, n_vol_fallback = 0;
	int opt_tile_waves = 4;   // k_tile: tiles per workgroup
	int opt_gates = 1;        // 0 = events instead of gate kernels between the streams of the steady-state path
	bool gates = false;       // ... as decided for the scan being enqueued
	int opt_sparse_set = 0;   // 1 = every scan's ray cells through the sparse set (Grid::layout 2), whatever its box needs
	int opt_cast_global = 1;  // 0 = grids beyond LDS go through k_dda_seg / k_dda (byte-per-block grid) instead of k_cast<true>
	u64 host_ns[4] = {0, 0, 0, 0};  // diagnostics: host time inside doInsert -- scan half enqueue, map half enqueue, join, total
	uint64_t n_fast = 0;
	u32 hb_cap_mask = 0;
	Ingest ing{};      // ufomap_map_insert_pointcloud2: raw PointCloud2 records, converted inside k_classify
	u32 hb_clean = 0;  // slots [0, hb_clean) of the hit-block hash are known to be empty
	DevBuf b_crec, b_dlist, b_rays;
	u64 es_set_slots = 0;  // early stopping, sparse form: slots of the set of first rays the last such scan needed
	DevBuf b_ray_pt, b_es_first, b_es_stop;  // early stopping (scan_kernels.h: k_es_*): the rays' ranks, who visits a cell first, the rays' stops
	u32 es_rounds = 0;                        // ... rounds the last such scan took to settle
	DevBuf b_gridM, b_entries, b_ent_slot, b_newlist, b_wl0, b_wl1, b_in_xyz, b_in_rgb, b_codes, b_dump;
	ScanCtl* h_ctl = nullptr;  // pinned
	ScanCtl* h_res = nullptr;  // pinned: k_ftail stores the finished control block here itself (no read-back copy, no stream sync)
	unsigned long long* sig_prep = nullptr;  // (HandOver)
	bool done_by_flag = false;
	bool res_direct = false;      // ... by a kernel of the general path that stored the result block itself (setValueVolume: k_vol_all)
	bool ctl_clean = false;    // the device control block holds the fast path's start state (k_ftail left it so): no upload
	DevBuf b_ctl_init;         // that start state, uploaded once
	bool ctl_init_done = false;
	void* h_stage = nullptr;   // pinned staging buffer of the current hand-over set (HandOver::h_stage)
	size_t h_stage_cap = 0;
	hipEvent_t copy_ev = nullptr;  // end of the H2D copy of a cloud that lies in caller-owned pinned memory
	bool copy_wait = false;        // ... which the call that enqueued it has not awaited yet (copyDone)
	StageWorker* stager = nullptr; // copies pageable clouds into pinned staging beside the calling thread (uploadCloud)
	unsigned long long* h_stage_flag = nullptr;  // pinned: StageWorker::flag
	bool stage_wait = false;       // the helper is still reading the caller's cloud: joined when the call ends (copyDone)
	int opt_stage_thread = 1;      // 0: the calling thread copies the cloud itself before it enqueues anything (as until round 5)
	MapRoot* h_root = nullptr;  // pinned
	size_t scratch_limit = 16ull << 30;
	// state of the last integration
	bool pending = false;
	int pending_status = UFOMAP_OK;
	ScanArgs args;                     // of the integration that uses the current hand-over set
	Grid spec_grid{};                  // ray grid predicted for the next depth-0 scan (from the last one's box + margin)
	bool spec_valid = false;
	int opt_spec = 1;                  // 0 = always read the boxes back before sizing the grid
	uint64_t n_spec = 0, n_spec_redo = 0;
	int opt_async_apply = 0;           // ufomap_map_apply_keys_batch returns after enqueueing (join at the next call / wait)
	Grid gridH{}, gridM{};
	bool haveH = false, haveM = false;
	u32 last_depth = 0;
	u32 hh_mask = 0;  // hit-hash mask of the current scan
	const uint8_t* last_rgb = nullptr;
	// diagnostic overrides (ufomap_map_set_option); -1 / 0 = automatic
	Need scan_new_bound;  // upper bound of what both phases of the current scan can create
	int opt_dda_mode = -1;
	int opt_dda_seg = 1;  // 0 = force the lane-per-ray kernel
	int opt_dda_block = 0, opt_dda_lanes = 0;  // 0 = automatic
	int opt_cast_oct = 0;        // 1: the steady-state ray kernel on octant sub-boxes (k_fcast4: 5 MB of slabs per scan instead of 20 -- and, measured, a
	                             // SLOWER scan: 0.0476 against 0.0415 ms); 0: a copy of the whole grid per workgroup (k_fcast3 / k_fcast2)
	int opt_cast_oct_lds = 80;   // ... as long as a workgroup's LDS (sub-box, queue, lists) stays below this many KiB
	int opt_es_sparse = 0;       // tests: early stopping keeps "who visits a cell first" in the sparse set whatever the ray box needs (2: starting from 1 Ki slots)
	int opt_ctl_dbg = 0;     // k_ftail also reports the control block's diagnostics (clock stamps: scripts/dev/dev_*.py) to the host -- 512 bytes more across PCIe per scan
	int opt_cast_fused = 2;  // the steady-state ray kernel: 2 = k_fcast3 (round 6: rays packed before set-up, cuts by estimate + check), else k_fcast2 (round 5, the cross-check)
	int opt_cast2_k = 64;    // ... its cells per segment (a cut costs ~1.5 us of a lane's chain: measured 32 -> 44.0, 48 -> 43.5, 64 -> 41.4, 96 -> 43.0 us per pipelined scan)
	int opt_cast = 1, opt_cast_wgs = 0, opt_cast_k = 32;  // fused ray kernel: on/off, workgroups (0 = 256), steps per segment
	int opt_bits = 1;     // 0 = never use the bit-per-cell grid / k_walk
	int opt_merge = 1;    // 0 = hits and misses as two separate passes over the tree also at insert depth 0
	u64 opt_entry_guess = 0;
	uint64_t counts[8] = {0};
	double min_change[3], max_change[3];
	// profiling
	bool profiling = false;
	std::vector<KernelStat> stats;
	std::vector<PendingEvent> pend_ev;
	std::vector<hipEvent_t> ev_pool;
};

namespace
{
int statIndex(ufomap_map* m, const char* name)
{
	for (size_t i = 0; i < m->stats.size(); ++i)
		if (m->stats[i].name == name || 0 == strcmp(m->stats[i].name, name)) return (int)i;
	KernelStat s;
	s.name = name;
	m->stats.push_back(s);
	return (int)m->stats.size() - 1;
}

hipEvent_t getEvent(ufomap_map* m)
{
	if (!m->ev_pool.empty()) {
		hipEvent_t e = m->ev_pool.back();
		m->ev_pool.pop_back();
		return e;
	}
	hipEvent_t e = nullptr;
	(void)hipEventCreate(&e);
	return e;
}

struct ProfScope {
	ufomap_map* m;
	PendingEvent pe;
	bool on;
	ProfScope(ufomap_map* mm, const char* name) : m(mm), on(mm->profiling)
	{
		if (on) {
			pe.a = getEvent(m);
			pe.b = getEvent(m);
			pe.stat = statIndex(m, name);
			(void)hipEventRecord(pe.a, m->cs);
		}
	}
	~ProfScope()
	{
		if (on) {
			(void)hipEventRecord(pe.b, m->cs);
			m->pend_ev.push_back(pe);
		}
	}
};

void drainEvents(ufomap_map* m)
{
	std::vector<PendingEvent> later;
	for (PendingEvent& pe : m->pend_ev) {
		if (hipEventQuery(pe.b) != hipSuccess) {
			later.push_back(pe);  // e.g. the next scan's kernels on the scan stream: not finished yet
			continue;
		}
		float ms = 0;
		if (hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess) {
			m->stats[pe.stat].launches += 1;
			m->stats[pe.stat].total_ms += ms;
		}
		m->ev_pool.push_back(pe.a);
		m->ev_pool.push_back(pe.b);
	}
	m->pend_ev.swap(later);
}

inline dim3 gridFor(u64 n, u32 block = 256, u32 maxBlocks = 4096)
{
	u64 b = (n + block - 1) / block;
	if (b < 1) b = 1;
	if (b > maxBlocks) b = maxBlocks;
	return dim3((u32)b);
}

// the first region for `capU` blocks of levels >= 4, `nG` tile groups (table.h); everything zeroed on the map stream
int allocTable(ufomap_map* m, u32 nG, u32 capU, Table* out, TableBufs* tb)
{
	capU = std::max<u32>(capU, 1024u);
	nG = std::max<u32>(nG, 64u);
	// (a prime: the directory is probed with a second hash as the stride, table.h, and every stride has to visit every entry)
	for (;; ++nG) {
		bool prime = 0 != (nG & 1u);
		for (u32 d = 3; prime && (u64)d * d <= nG; d += 2) prime = 0 != nG % d;
		if (prime) break;
	}
	const u64 cap64 = (u64)capU + (u64)UFO_GROUP * nG;
	if (cap64 > (1ull << 31)) return fail(UFOMAP_ERR_CAPACITY, "node table would exceed 2^31 blocks");
	const u32 cap = (u32)cap64;
	HIP_TRY(tb->blk.reserve((size_t)cap * UFO_SLOT_BYTES));
	HIP_TRY(tb->tmax.reserve((size_t)cap * 8));
	HIP_TRY(tb->luocc.reserve((size_t)cap * 4));
	HIP_TRY(tb->lufl.reserve((size_t)cap * 4));
	HIP_TRY(tb->gdir.reserve((size_t)nG * 8));
	HIP_TRY(tb->gcnt.reserve(UFO_GCNT_WORDS * 4));
	if (m->g.color) {
		HIP_TRY(tb->rgb.reserve((size_t)cap * 32));
		HIP_TRY(tb->lurgb.reserve((size_t)cap * 4));
	}
	// an empty slot is key 0, flags 0, stamp 0 (values and parent are written when a block is created: not zeroed -- 16 of a
	// slot's 52 bytes; the fresh scan of a 2 mm frame allocates 7e7 slots)
	HIP_TRY(hipMemsetAsync((char*)tb->blk.p + (size_t)cap * 32, 0, (size_t)cap * 16, m->stream));
	HIP_TRY(hipMemsetAsync(tb->tmax.p, 0, (size_t)cap * 8, m->stream));
	HIP_TRY(hipMemsetAsync(tb->lufl.p, 0, (size_t)cap * 4, m->stream));
	HIP_TRY(hipMemsetAsync(tb->gdir.p, 0, (size_t)nG * 8, m->stream));
	HIP_TRY(hipMemsetAsync(tb->gcnt.p, 0, UFO_GCNT_WORDS * 4, m->stream));
	// [occ 32 B][key 8 B][flags 4 B][stamp 4 B][parent 4 B] x cap
	out->occA = tb->blk.as<float>();
	out->keyA = reinterpret_cast<u64*>((char*)tb->blk.p + (size_t)cap * 32);
	out->flagsA = reinterpret_cast<u32*>((char*)tb->blk.p + (size_t)cap * 40);
	out->stampA = reinterpret_cast<u32*>((char*)tb->blk.p + (size_t)cap * 44);
	out->parentA = reinterpret_cast<u32*>((char*)tb->blk.p + (size_t)cap * 48);
	out->rgb = m->g.color ? tb->rgb.as<u32>() : nullptr;
	out->tmax = tb->tmax.as<u64>();
	out->lu_occ = tb->luocc.as<float>();
	out->lu_fl = tb->lufl.as<u32>();
	out->lu_rgb = m->g.color ? tb->lurgb.as<u32>() : nullptr;
	out->root = m->b_root.as<MapRoot>();
	out->mask = cap - 1;
	out->capU = capU;
	out->nG = nG;
	out->gdir = tb->gdir.as<u64>();
	out->gcnt = tb->gcnt.as<u32>();
	out->L = m->g.L;
	return UFOMAP_OK;
}

int growTable(ufomap_map* m, u32 new_nG, u32 new_capU)
{
	Table nt{};
	TableBufs nb;
	g_n_rehash.fetch_add(1, std::memory_order_relaxed);
	int rc = allocTable(m, new_nG, new_capU, &nt, &nb);
	if (rc) return rc;
	if (m->cs != m->stream) HIP_TRY(hipStreamSynchronize(m->stream));  // (the new arrays are zeroed on the map stream)
	u32* d_fail = m->b_ctl.as<u32>() + (sizeof(ScanCtl) + 3) / 4;  // four spare words after the control block: failures, blocks copied, groups, first-region blocks
	HIP_TRY(hipMemsetAsync(d_fail, 0, 16, m->cs));
	{
		ProfScope ps(m, "k_rehash_copy");
		hipLaunchKernelGGL(k_rehash_copy, gridFor((u64)m->t.mask + 1), dim3(256), 0, m->cs, m->t, nt, d_fail, d_fail + 1);
	}
	{
		ProfScope ps(m, "k_rehash_parents");
		hipLaunchKernelGGL(k_rehash_parents, gridFor((u64)nt.mask + 1), dim3(256), 0, m->cs, nt);
	}
	hipLaunchKernelGGL(k_table_counts, dim3(1), dim3(64), 0, m->cs, nt, d_fail + 2);
	HIP_TRY(hipStreamSynchronize(m->cs));
	u32 res[4] = {0, 0, 0, 0};
	HIP_TRY(hipMemcpy(res, d_fail, 16, hipMemcpyDeviceToHost));
	if (res[0])  // (cannot happen: the new table holds what the old one does and more; the old table stays, `nb` is released on return)
		return fail(UFOMAP_ERR_CAPACITY, "re-hash into the larger node table failed; map unchanged");
	// collapsed blocks were left behind: the fill count is what was copied
	HIP_TRY(hipMemcpy(&m->b_root.as<MapRoot>()->used, &res[1], 4, hipMemcpyHostToDevice));
	m->used_est = res[1];
	m->used_g = res[2];
	m->used_u = res[3];
	m->tb = std::move(nb);  // (releases the old arrays)
	m->t = nt;
	return UFOMAP_OK;
}

// Does the table take what is in flight plus `need`? Groups up to 85 % of the directory, the first region up to load 0.6.
bool tableTakes(const ufomap_map* m, const Need& need)
{
	return (m->used_g + need.groups) * 20 <= (u64)m->t.nG * 17 && (m->used_u + need.upper) * 5 <= (u64)m->t.capU * 3;
}
// ... a table that does (each region that is short: at least 1.5 x what it has, so that a map that keeps growing is
// re-hashed a logarithmic number of times; 80 % / 57 % full with everything `need` names). Joins nothing: the caller has.
int growFor(ufomap_map* m, const Need& need)
{
	u64 nG = m->t.nG, capU = m->t.capU;
	if ((m->used_g + need.groups) * 20 > nG * 17) nG = std::max<u64>(nG + nG / 2, (m->used_g + need.groups) * 5 / 4 + 64);
	if ((m->used_u + need.upper) * 5 > capU * 3) capU = std::max<u64>(capU + capU / 2, (m->used_u + need.upper) * 7 / 4 + 1024);
	nG = (nG + 63) & ~63ull;
	capU = (capU + 4095) & ~4095ull;
	if (capU + UFO_GROUP * nG > (1ull << 31)) return fail(UFOMAP_ERR_CAPACITY, "node table would exceed 2^31 blocks");
	return growTable(m, (u32)nG, (u32)capU);
}
// the table's fill from the device (after updates that have no propagation tail to report it: volumes, streams read)
int refreshFill(ufomap_map* m)
{
	u32* d_cnt = m->b_ctl.as<u32>() + (sizeof(ScanCtl) + 3) / 4;
	hipLaunchKernelGGL(k_table_counts, dim3(1), dim3(64), 0, m->stream, m->t, d_cnt + 2);
	u32 res[2] = {0, 0};
	HIP_TRY(hipMemcpyAsync(res, d_cnt + 2, 8, hipMemcpyDeviceToHost, m->stream));
	HIP_TRY(hipMemcpyAsync(m->h_root, m->b_root.p, sizeof(MapRoot), hipMemcpyDeviceToHost, m->stream));
	HIP_TRY(hipStreamSynchronize(m->stream));
	m->used_est = m->h_root->used;
	m->used_g = res[0];
	m->used_u = res[1];
	return UFOMAP_OK;
}

int resetRoot(ufomap_map* m)
{
	// a fresh map's root is one unknown leaf (occupancy_map_base.h:871)
	m->h_root->occ = 0.0f;
	m->h_root->flags = (isFreeV(m->g, 0.0f) ? 1u : 0u) | (isUnknownV(m->g, 0.0f) ? 2u : 0u);
	m->h_root->rgb = 0;
	m->h_root->used = 0;
	// (the change log's counters behind these four words stay: Octree::clear does not touch changes_)
	HIP_TRY(hipMemcpyAsync(m->b_root.p, m->h_root, offsetof(MapRoot, n_changes), hipMemcpyHostToDevice, m->stream));
	HIP_TRY(hipStreamSynchronize(m->stream));
	m->used_est = 0;
	m->used_g = m->used_u = 0;
	return UFOMAP_OK;
}

// sensor model from its six logits (occupancy_map_base.h:864-869, 909): thresholds stay double (OMB:926-940), the
// update and clamping values are narrowed to LogitType = float where they are used (OMB:296, 311, 1142-1143)
void applyModel(ufomap_map* m)
{
	MapGeom& g = m->g;
	g.occ_thr = m->model_log[0];
	g.free_thr = m->model_log[1];
	g.hit = (float)m->model_log[2];
	g.miss_log = m->model_log[3];
	g.cmin = (float)m->model_log[4];
	g.cmax = (float)m->model_log[5];
	// toProb(update) with LogitType=float: std::exp(float) (occupancy_map_base.h:911)
	g.prob_hit_f = 1.0 / (1.0 + std::exp(-g.hit));
}
void setSensorModel(ufomap_map* m, double occupied_thres, double free_thres, double prob_hit, double prob_miss, double cmin, double cmax)
{
	auto logit = [](double p) { return std::log(p / (1.0 - p)); };  // occupancy_map_base.h:909
	const double v[6] = {occupied_thres, free_thres, prob_hit, prob_miss, cmin, cmax};
	for (int k = 0; k < 6; ++k) m->model_log[k] = logit(v[k]);
	applyModel(m);
}

// upper bound on what a list of n entries at `level` inside a grid of nb[] blocks can add to the node table: per level, no
// more distinct ancestors than entries, nor than fit in the bounding box. Groups: the level-3 blocks among them.
Need needBound(const ufomap_map* m, u64 n, const i32 nb[3], u32 level)
{
	Need r;
	for (u32 l = level; l <= m->g.L; ++l) {
		u32 sh = l - level;
		long double vol = 1;
		for (int a = 0; a < 3; ++a) vol *= (long double)(((u64)nb[a] >> std::min(sh, 62u)) + 2);
		u64 lim = vol > 1e18L ? (u64)1e18 : (u64)vol;
		const u64 c = std::min<u64>(n, lim);
		r.blocks += c;
		if (m->g.L < 4 || l >= 4) r.upper += c;
		else if (3 == l) r.groups += c;
	}
	r.blocks += 8;
	r.upper += 8;
	if (level <= 3 && m->g.L >= 4) r.groups += 1;
	return r;
}
u64 blockBound(const ufomap_map* m, u64 n, const i32 nb[3], u32 level) { return needBound(m, n, nb, level).blocks; }

// exchange the current hand-over set (members of the map object) with another one
void swapWith(ufomap_map* m, HandOver& o)
{
	std::swap(m->b_ctl, o.b_ctl);
	std::swap(m->b_entries, o.b_entries);
	std::swap(m->b_hh_keys, o.b_hh_keys);
	std::swap(m->b_in_xyz, o.b_in_xyz);
	std::swap(m->b_in_rgb, o.b_in_rgb);
	std::swap(m->b_gridM, o.b_gridM);
	std::swap(m->b_gridH, o.b_gridH);
	std::swap(m->fseq, o.fseq);
	std::swap(m->b_keep, o.b_keep);
	std::swap(m->b_keep_rgb, o.b_keep_rgb);
	std::swap(m->deferred, o.deferred);
	std::swap(m->has_slot, o.has_slot);
	std::swap(m->batch_world, o.batch_world);
	std::swap(m->comm, o.comm);
	std::swap(m->b_xsend, o.b_xsend);
	std::swap(m->b_xrecv, o.b_xrecv);
	std::swap(m->b_bpipe, o.b_bpipe);
	std::swap(m->h_res_all, o.h_res_all);
	std::swap(m->h_res_all_world, o.h_res_all_world);
	std::swap(m->xchg_ev, o.xchg_ev);
	std::swap(m->b_slabs, o.b_slabs);
	std::swap(m->hit_grid, o.hit_grid);
	std::swap(m->b_part1, o.b_part1);
	std::swap(m->b_hit_code, o.b_hit_code);
	std::swap(m->b_first, o.b_first);
	std::swap(m->b_tilebits, o.b_tilebits);
	std::swap(m->ugeo, o.ugeo);
	std::swap(m->first_dirty, o.first_dirty);
	std::swap(m->seq, o.seq);
	std::swap(m->fast, o.fast);
	std::swap(m->fgeo, o.fgeo);
	std::swap(m->h_ctl, o.h_ctl);
	std::swap(m->h_res, o.h_res);
	std::swap(m->sig_prep, o.sig_prep);
	std::swap(m->done_by_flag, o.done_by_flag);
	std::swap(m->ctl_clean, o.ctl_clean);
	std::swap(m->h_stage, o.h_stage);
	std::swap(m->h_stage_cap, o.h_stage_cap);
	std::swap(m->hh_mask, o.hh_mask);
	for (int k = 0; k < 8; ++k) std::swap(m->counts[k], o.counts[k]);
	std::swap(m->args, o.args);
	std::swap(m->done_ev, o.done_ev);
	std::swap(m->pending, o.pending);
	std::swap(m->bound, o.bound);
	std::swap(m->b_vM, o.b_vM);
	std::swap(m->b_vMm, o.b_vMm);
	std::swap(m->b_vH, o.b_vH);
	std::swap(m->b_vtb, o.b_vtb);
	std::swap(m->b_vlist, o.b_vlist);
	std::swap(m->b_vcopies, o.b_vcopies);
	std::swap(m->b_vslots, o.b_vslots);
	std::swap(m->b_vaux, o.b_vaux);
	std::swap(m->b_vupbits, o.b_vupbits);
	std::swap(m->b_vrec, o.b_vrec);
	std::swap(m->vol, o.vol);
	std::swap(m->vol_dirty, o.vol_dirty);
	std::swap(m->vol_walk, o.vol_walk);
	std::swap(m->vol_count, o.vol_count);
	std::swap(m->vol_scan_id, o.vol_scan_id);
	std::swap(m->vol_rgb, o.vol_rgb);
	std::swap(m->vplan, o.vplan);
}

int finishSet(ufomap_map* m, int k);
int finishPending(ufomap_map* m);

// Stream-to-stream hand-overs of the steady-state path are spinning one-wave kernels (k_gate) unless something is known
// to serialise kernels across streams -- then a gate would wait for a producer that is not allowed to run: counter
// collection of rocprofv3 (--pmc), rocprof v1/v2 (HSA_TOOLS_LIB), AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING. Events then.
bool useGates(const ufomap_map* m)
{
	static const bool env_ok = [] {
		auto on = [](const char* k) {
			const char* v = getenv(k);
			return v && *v && 0 != strcmp(v, "0");
		};
		return !(on("ROCPROF_COUNTER_COLLECTION") || getenv("ROCPROF_COUNTERS") || getenv("HSA_TOOLS_LIB") || on("AMD_SERIALIZE_KERNEL") ||
		         on("HIP_LAUNCH_BLOCKING") || on("UFOMAP_NO_GATES"));
	}();
	return env_ok && 0 != m->opt_gates;
}
extern "C" int ufomap_map_wait(ufomap_map* m);

// Phase numbers (ufomap_map::scan_id) tag "created / reached / timed in this phase" in the table: 24 bits in tmax, 22 in
// lu_fl, 32 in the stamps and the fast path's tile records. Before the shortest tag can wrap, everything in flight is
// joined, every tag in the table is cleared and the numbering starts over -- a full pass over the table every ~4 million
// phases (about 2 million scans). `phase_limit` is an option so that tests can make it happen every few scans.
int phaseGuard(ufomap_map* m)
{
	if (m->scan_id < m->phase_limit) return UFOMAP_OK;
	const int rc = ufomap_map_wait(m);
	if (rc) return rc;
	hipLaunchKernelGGL(k_reset_tags, gridFor((u64)m->t.mask + 1), dim3(256), 0, m->stream, m->t);
	if (m->b_tilerec.p) HIP_TRY(hipMemsetAsync(m->b_tilerec.p, 0, m->b_tilerec.cap, m->stream));
	if (m->b_vrec.p) HIP_TRY(hipMemsetAsync(m->b_vrec.p, 0, m->b_vrec.cap, m->stream));
	for (int i = 0; i < kAlt; ++i)
		if (m->alt[i].b_vrec.p) HIP_TRY(hipMemsetAsync(m->alt[i].b_vrec.p, 0, m->alt[i].b_vrec.cap, m->stream));
	HIP_TRY(hipStreamSynchronize(m->stream));
	m->scan_id = 0;
	++m->n_phase_resets;
	return UFOMAP_OK;
}

// Wait for the integration that uses hand-over set s. General path: the set's event. Fast path: k_ftail's last action is a
// store of the integration's number into pinned memory behind the result block -- the host reads that word instead of
// having the stream process an event record after every scan; if the word does not show up within a few milliseconds
// (a stream error?) the map stream is synchronised, which reports whatever went wrong.
hipError_t waitSetDone(ufomap_map* m, HandOver& s)
{
	if (!s.done_by_flag) return hipEventSynchronize(s.done_ev);
	volatile unsigned long long* done = reinterpret_cast<volatile unsigned long long*>(s.h_res + 1);
	const auto t0 = std::chrono::steady_clock::now();
	for (u32 spins = 0; *done != (unsigned long long)s.seq; ++spins) {
		if (0 == (spins & 1023u) && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) {
			const hipError_t e = hipStreamSynchronize(m->stream);
			if (e != hipSuccess) return e;
			if (*done != (unsigned long long)s.seq) return hipErrorUnknown;  // (the kernel ran and did not store: cannot happen)
			break;
		}
	}
	std::atomic_thread_fence(std::memory_order_acquire);
	return hipSuccess;
}

// ---- the hand-over sets in flight, told apart by their integrations' running numbers (HandOver::seq) ----
int oldestPendingAlt(const ufomap_map* m)
{
	int k = -1;
	for (int i = 0; i < kAlt; ++i)
		if (m->alt[i].pending && (k < 0 || m->alt[i].seq < m->alt[k].seq)) k = i;
	return k;
}
int countPendingAlts(const ufomap_map* m)
{
	int n = 0;
	for (int i = 0; i < kAlt; ++i) n += m->alt[i].pending ? 1 : 0;
	return n;
}
// has the integration of set s finished? (never blocks)
bool setDoneNow(const HandOver& s)
{
	bool done;
	if (s.done_by_flag) done = *reinterpret_cast<const volatile unsigned long long*>(s.h_res + 1) == (unsigned long long)s.seq;
	else done = hipEventQuery(s.done_ev) == hipSuccess;
	if (done) std::atomic_thread_fence(std::memory_order_acquire);
	return done;
}
int flushDeferred(ufomap_map* m, bool publish = true);

// Join the oldest integration among the other sets; if its tree update has not been enqueued yet (it was waiting for
// company, doInsert), that happens first.
int joinOldestAlt(ufomap_map* m)
{
	const int k = oldestPendingAlt(m);
	if (k < 0) return UFOMAP_OK;
	if (m->alt[k].deferred) {
		const int frc = flushDeferred(m);
		if (frc) return frc;
	}
	HIP_TRY(waitSetDone(m, m->alt[k]));
	const int rc = finishSet(m, k);
	if (rc && UFOMAP_OK == m->async_status) m->async_status = rc;
	return rc;
}

// Every integration whose tree update has been enqueued is joined, oldest first; what still waits for company stays.
// (After a flagged integration has been repeated everything enqueued behind it has stood back and is repeated here.)
int joinEnqueued(ufomap_map* m)
{
	HIP_TRY(hipStreamSynchronize(m->stream));
	int rc = UFOMAP_OK;
	for (;;) {
		const int k = oldestPendingAlt(m);
		if (k < 0 || m->alt[k].deferred) break;
		// (a scan that goes with the slot of a newer scan -- no slot of its own -- while that slot is still being enqueued: this very
		// function is called from there when the table has to grow. Nothing of it is on the map stream yet: it waits for company
		// like a deferred one. Taking its unwritten result block for a finished update released its set under the walk to come.)
		if (m->alt[k].fast && m->alt[k].done_by_flag && !m->alt[k].has_slot && !m->alt[k].batch_world && !m->alt[k].vol_walk && m->alt[k].fseq > m->last_slot_fseq) break;
		const int r = finishSet(m, k);
		if (!rc) rc = r;
	}
	if (m->pending && !m->deferred) {
		HIP_TRY(hipStreamSynchronize(m->stream));  // (a repeated scan above has enqueued more)
		const int r = finishPending(m);
		if (!rc) rc = r;
	}
	if (rc && UFOMAP_OK == m->async_status) m->async_status = rc;
	return rc;
}

// Integrations that have completed are taken in (grid prediction, table fill, flagged scans repeated), oldest first,
// without waiting for anything. Stops at the first one that flagged itself (ufomap_map::prev_flagged: the caller drains).
int joinCompleted(ufomap_map* m)
{
	int rc = UFOMAP_OK;
	m->prev_flagged = false;
	for (;;) {
		const int k = oldestPendingAlt(m);
		if (k < 0 || m->alt[k].deferred || !setDoneNow(m->alt[k])) break;
		const int r = finishSet(m, k);
		if (!rc) rc = r;
		if (m->prev_flagged) break;
	}
	if (rc && UFOMAP_OK == m->async_status) m->async_status = rc;
	return rc;
}

// A new integration begins: it gets a hand-over set of its own. The current set is reused if its integration has been
// joined; else the current set changes places with an idle one, or -- all sets in use -- with the oldest integration's,
// which is joined first.
int rotateSets(ufomap_map* m)
{
	int rc = phaseGuard(m);
	if (rc) return rc;
	if (!m->pending) return UFOMAP_OK;
	int k = -1;
	for (int i = 0; i < kAlt && k < 0; ++i)
		if (!m->alt[i].pending) k = i;
	if (k < 0) {
		k = oldestPendingAlt(m);
		rc = joinOldestAlt(m);
		if (m->alt[k].pending) return rc ? rc : fail(UFOMAP_ERR_DEVICE, "no hand-over set could be freed");
	}
	swapWith(m, m->alt[k]);
	return rc;
}

int readCtl(ufomap_map* m)
{
	HIP_TRY(hipMemcpyAsync(m->h_ctl, m->b_ctl.p, sizeof(ScanCtl), hipMemcpyDeviceToHost, m->cs));
	HIP_TRY(hipMemcpyAsync(m->h_root, m->b_root.p, sizeof(MapRoot), hipMemcpyDeviceToHost, m->cs));
	HIP_TRY(hipStreamSynchronize(m->cs));
	m->used_est = m->h_root->used;  // (the regions' counts are not needed where this is called: counting passes before a growth decision)
	return UFOMAP_OK;
}

// the same for an integration that is known to be complete (stream or event synchronised by the caller): on the
// read-back stream, so that it does not queue behind a later update already enqueued on the map stream
int readCtlDone(ufomap_map* m)
{
	HIP_TRY(hipMemcpyAsync(m->h_ctl, m->b_ctl.p, sizeof(ScanCtl), hipMemcpyDeviceToHost, m->xstream));
	HIP_TRY(hipStreamSynchronize(m->xstream));
	if (m->h_ctl->used_now) {
		m->used_est = m->h_ctl->used_now;  // MapRoot::used as the propagation tail saw it (after all creations of the update)
		m->used_g = m->h_ctl->used_g_now;
		m->used_u = m->h_ctl->used_u_now;
	} else {
		// an update without a propagation tail (nothing to apply, or it stood back): read the root and the regions' counters
		u32* d_cnt = m->b_ctl.as<u32>() + (sizeof(ScanCtl) + 3) / 4;
		hipLaunchKernelGGL(k_table_counts, dim3(1), dim3(64), 0, m->xstream, m->t, d_cnt + 2);
		u32 res[2] = {0, 0};
		HIP_TRY(hipMemcpyAsync(res, d_cnt + 2, 8, hipMemcpyDeviceToHost, m->xstream));
		HIP_TRY(hipMemcpyAsync(m->h_root, m->b_root.p, sizeof(MapRoot), hipMemcpyDeviceToHost, m->xstream));
		HIP_TRY(hipStreamSynchronize(m->xstream));
		m->used_est = m->h_root->used;
		m->used_g = res[0];
		m->used_u = res[1];
	}
	return UFOMAP_OK;
}

int ctlError(ufomap_map* m)
{
	u32 e = m->h_ctl->err;
	if (!e) return UFOMAP_OK;
	if (e & ERR_RUNAWAY)
		return fail(UFOMAP_ERR_RUNAWAY,
		            "a clipped ray left the map cube (end point outside after moveLineInside); the reference walks ~2^31 "
		            "cells on this input. Map unchanged.");
	if (e & ERR_GATE)
		return fail(UFOMAP_ERR_DEVICE, "a stream hand-over timed out (are kernels being serialised across streams by a tool? set UFOMAP_NO_GATES=1); map unchanged");
	if (e & ERR_TABLE_FULL) {
		// blocks created before the table ran full are linked into the tree with unset contents: the handle refuses
		// further work until ufomap_map_clear
		m->poisoned = true;
		return fail(UFOMAP_ERR_CAPACITY, "node table full (internal bound violated): the map is inconsistent, ufomap_map_clear it");
	}
	if (e & ERR_ENTRIES) return fail(UFOMAP_ERR_CAPACITY, "update list larger than its buffer (internal bound violated); map unchanged");
	if (e & ERR_GRID_OOB) return fail(UFOMAP_ERR_CAPACITY, "a ray cell fell outside the scan grid (internal bound violated)");
	return fail(UFOMAP_ERR_CAPACITY, "hit hash full (internal bound violated)");
}

int makeGrid(const i32 mn[3], const i32 mx[3], u32 depth, Grid* gr)
{
	u64 bytes = 1;
	for (int a = 0; a < 3; ++a) {
		long long lo = ((long long)mn[a] - 2) & ~1LL;  // pad by one block, keep the base even
		long long hi = (long long)mx[a] + 2;
		long long nb = (hi - lo) / 2 + 1;
		if (nb <= 0 || nb > (1LL << 30)) return UFOMAP_ERR_CAPACITY;
		gr->base[a] = (i32)lo;
		gr->nb[a] = (i32)nb;
		if (bytes > (1ull << 62) / (u64)nb) return UFOMAP_ERR_CAPACITY;
		bytes *= (u64)nb;
	}
	gr->depth = depth;
	gr->layout = 0;
	gr->bytes = (bytes + 15) & ~15ull;  // whole uint4 words (k_dda's LDS copy is read 16 B at a time)
	return UFOMAP_OK;
}

// upper bound on the blocks of `level` that can lie on the paths of entries inside a grid of nb[] blocks
u64 levelBound(const i32 nb[3], u32 shift)
{
	long double vol = 1;
	for (int a = 0; a < 3; ++a) vol *= (long double)(((u64)nb[a] >> std::min(shift, 62u)) + 2);
	return vol > 1e18L ? (u64)1e18 : (u64)vol;
}


// ---- change log (change detection) ------------------------------------------------------------------------------
// the log the update kernels append to; empty (disabled) unless enableChangeDetection is on
ChangeLog changeLog(const ufomap_map* m) { return m->chg_enabled ? ChangeLog{m->b_changes.as<u64>(), m->chg_cap, m->g.L} : ChangeLog{nullptr, 0u, m->g.L}; }

// room for `extra` more records: the log grows by copy. Synchronises the map stream (change detection runs every
// update synchronously, see doInsert: it is a diagnostic mode, not the fast path).
int ensureChangeCap(ufomap_map* m, u64 extra)
{
	if (!m->chg_enabled) return UFOMAP_OK;
	HIP_TRY(hipStreamSynchronize(m->stream));
	HIP_TRY(hipMemcpy(m->h_root, m->b_root.p, sizeof(MapRoot), hipMemcpyDeviceToHost));
	const u64 have = m->h_root->n_changes;
	if (have + extra <= m->chg_cap) return UFOMAP_OK;
	const u64 want = std::max<u64>(have + extra, (u64)m->chg_cap * 2);
	if (want > 0xFFFFFFF0ull) return fail(UFOMAP_ERR_CAPACITY, "change log exceeds 2^32 records (read or reset the change set)");
	void* np = nullptr;
	HIP_TRY(hipMalloc(&np, (size_t)want * 8));
	if (have) {
		hipError_t e = hipMemcpy(np, m->b_changes.p, (size_t)std::min<u64>(have, m->chg_cap) * 8, hipMemcpyDeviceToDevice);
		if (e != hipSuccess) {
			(void)hipFree(np);
			return fail(UFOMAP_ERR_DEVICE, hipGetErrorString(e));
		}
	}
	m->b_changes.release();
	m->b_changes.p = np;
	m->b_changes.cap = (size_t)want * 8;
	m->chg_cap = (u32)want;
	return UFOMAP_OK;
}

// updateParents (OMB:1126-1133) from level `first` upward: wide levels one launch each, the narrow rest in one
// launch. bound(l) = upper bound of the queued blocks of level l; the level-l worklist is b_wl[l & 1].
template <typename F>
void propagateLevels(ufomap_map* m, u32 first, F bound, ScanCtl::PhaseCtr* pc, u32 which)
{
	ScanCtl* ctl = m->b_ctl.as<ScanCtl>();
	u32* wl[2] = {m->b_wl0.as<u32>(), m->b_wl1.as<u32>()};
	u32 l = first;
	for (; l <= m->g.L; ++l) {
		const u64 b = bound(l);
		if (b <= 2048) break;
		ProfScope ps(m, "k_propagate");
		hipLaunchKernelGGL(k_propagate, gridFor(b, 256, 1024), dim3(256), 0, m->cs, m->t, m->g, wl[l & 1], wl[(l + 1) & 1], l, m->scan_id, pc,
		                   ctl);
	}
	if (l <= m->g.L) {
		ProfScope ps(m, "k_propagate_tail");
		hipLaunchKernelGGL(k_propagate_tail, dim3(1), dim3(1024), 0, m->cs, m->t, m->g, wl[0], wl[1], l, m->scan_id, pc, ctl, which * 32u);
	}
}

// One phase of the map update: entries of one level -> ensure, init, apply, propagate.
// `cap` is the capacity of the entry buffer (the device-side count may be smaller; if it is larger
// k_ensure raises ERR_ENTRIES and nothing is applied).
// Merged mode (nbB != nullptr, level 1): the list holds every touched block of the scan once -- up to cap - capB
// blocks inside grid nb[] (hits) and capB blocks inside grid nbB[] (misses); `which` is 0.
int applyEntries(ufomap_map* m, const Entry* d_entries, u32 cap, u32 which, u32 level, const i32 nb[3], float upd,
                 const uint8_t* d_rgb, bool zero_ctr, u32 cap_h, u32 cap_m, u32 capB = 0, const i32* nbB = nullptr,
                 float upd_miss = 0.f, const ScanCtl* prev = nullptr)
{
	if (0 == cap) return UFOMAP_OK;
	const bool merged = nullptr != nbB;
	const u32 capA = merged ? cap - capB : cap;
	// blocks of level `level + sh` that the entries' paths can touch
	auto lvlBound = [&](u32 sh) {
		u64 b = capA ? std::min<u64>(capA, levelBound(nb, sh)) : 0;
		if (merged && capB) b += std::min<u64>(capB, levelBound(nbB, sh));
		return b;
	};
	if (m->chg_enabled) {
		const int crc = ensureChangeCap(m, (u64)cap * 8 + ((1 == level) ? 0 : (u64)m->t.mask + 1));
		if (crc) return crc;
	}
	const ChangeLog cl = changeLog(m);
	m->scan_id += 1;  // "new this phase" stamp: blocks made by an earlier phase of the same scan are old
	u64 newcap = (capA ? blockBound(m, capA, nb, level) : 0) + ((merged && capB) ? blockBound(m, capB, nbB, level) : 0);
	HIP_TRY(m->b_ent_slot.reserve((size_t)cap * 4));
	HIP_TRY(m->b_newlist.reserve((size_t)newcap * 4));
	size_t wlcap = (size_t)lvlBound(1) + 8;
	HIP_TRY(m->b_wl0.reserve(wlcap * 4));
	HIP_TRY(m->b_wl1.reserve(wlcap * 4));
	ScanCtl* ctl = m->b_ctl.as<ScanCtl>();
	const u32* d_n = &ctl->n_entries[which];
	ScanCtl::PhaseCtr* pc = &ctl->ph[which];
	if (zero_ctr) HIP_TRY(hipMemsetAsync(pc, 0, sizeof(ScanCtl::PhaseCtr), m->cs));
	HitHash hh{m->b_hh_keys.as<u64>(), reinterpret_cast<u32*>(m->b_hh_keys.as<u64>() + ((size_t)m->hh_mask + 1)), m->hh_mask};
	u32* wl[2] = {m->b_wl0.as<u32>(), m->b_wl1.as<u32>()};
	dim3 ge = gridFor(cap);
	{
		ProfScope ps(m, "k_ensure");
		hipLaunchKernelGGL(k_ensure, ge, dim3(256), 0, m->cs, m->t, m->g, d_entries, d_n, cap_h, cap_m, m->scan_id, m->b_ent_slot.as<u32>(),
		                   m->b_newlist.as<u32>(), (u32)std::min<u64>(newcap, 0xFFFFFFFFull), pc, ctl, prev);
	}
	{
		ProfScope ps(m, "k_init_new");
		hipLaunchKernelGGL(k_init_new, gridFor(std::min<u64>(newcap, cap)), dim3(256), 0, m->cs, m->t, m->g, m->b_newlist.as<u32>(),
		                   (u32)std::min<u64>(newcap, 0xFFFFFFFFull), m->scan_id, pc, ctl);
	}
	if (1 == level) {
		ProfScope ps(m, "k_apply_leaf");
		const u32 mode = merged ? 2u : (which == 0 ? 1u : 0u);
		hipLaunchKernelGGL(k_apply_leaf, ge, dim3(256), 0, m->cs, m->t, m->g, d_entries, d_n, m->b_ent_slot.as<u32>(),
		                   merged ? upd : (which == 0 ? upd : 0.f), merged ? upd_miss : (which == 0 ? 0.f : upd), mode, m->scan_id, hh,
		                   d_rgb, wl[0], pc, ctl, cl);
	} else {
		// coarse misses: level-synchronous walk of the subtrees below the masked children (map_kernels.h S3c)
		// every live block can be visited: blocks known at the last control-block read + everything this scan may add
		const u64 dcap64 = std::min<u64>((u64)m->t.mask + 1, m->used_est + m->scan_new_bound.blocks + 8);
		const u32 dcap = (u32)std::min<u64>(dcap64, 0xFFFFFFF0ull);
		HIP_TRY(m->b_crec.reserve((size_t)cap * sizeof(CoarseRec)));
		HIP_TRY(m->b_dlist.reserve((size_t)dcap * 4));
		if (zero_ctr) HIP_TRY(hipMemsetAsync(&ctl->dl_total, 0, 4 * 26, m->cs));  // dl_total + dl_start[25]
		CoarseRec* rec = m->b_crec.as<CoarseRec>();
		u32* dl = m->b_dlist.as<u32>();
		dim3 gd = gridFor(std::max<u64>(dcap, 256), 256, 4096);
		{
			ProfScope ps(m, "k_coarse_begin");
			hipLaunchKernelGGL(k_coarse_begin, ge, dim3(256), 0, m->cs, m->t, m->g, d_entries, d_n, m->b_ent_slot.as<u32>(), upd, rec,
			                   dl, dcap, ctl, cl);
			hipLaunchKernelGGL(k_coarse_mark, dim3(1), dim3(1), 0, m->cs, ctl, level - 1);
		}
		for (u32 l = level - 1; l >= 1; --l) {
			ProfScope ps(m, "k_coarse_down");
			hipLaunchKernelGGL(k_coarse_down, gd, dim3(256), 0, m->cs, m->t, m->g, l, upd, dl, dcap, ctl, cl, level - 1);
			if (l > 1) hipLaunchKernelGGL(k_coarse_mark, dim3(1), dim3(1), 0, m->cs, ctl, l - 1);
		}
		for (u32 l = 1; l + 1 <= level; ++l) {
			ProfScope ps(m, "k_coarse_up");
			hipLaunchKernelGGL(k_coarse_up, gd, dim3(256), 0, m->cs, m->t, m->g, l, dl, dcap, ctl);
		}
		{
			ProfScope ps(m, "k_coarse_end");
			hipLaunchKernelGGL(k_coarse_end, ge, dim3(256), 0, m->cs, m->t, m->g, d_entries, d_n, m->b_ent_slot.as<u32>(), rec,
			                   m->scan_id, wl[(level + 1) & 1], pc, ctl);
		}
	}
	propagateLevels(m, level + 1, [&](u32 l) { return lvlBound(l - level); }, pc, which);
	HIP_TRY(hipGetLastError());
	return UFOMAP_OK;
}

// Make sure the node table can take what both phases may create. The a-priori bound (blockBound with every
// entry new) is a true upper bound but far too pessimistic on a warm map; when it asks for growth, count
// the entries whose block is really missing (one extra kernel + host read on this rare path) and bound again.
// extra_used / no_grow: the caller has an update in flight that may add up to extra_used blocks and cannot wait for
// it here; if the table might not take both, 1 is returned (nothing done) and the caller joins first.
int sizeTable(ufomap_map* m, const Entry* ent_h, u64 capH, const i32 nbH[3], const Entry* ent_m, u64 capM, const i32 nbM[3],
              unsigned depth, bool merged = false, Need extra_used = Need{}, bool no_grow = false, u32 headroom_scans = 0)
{
	// merged list (depth 0) whose hit box lies inside the ray box: all entries are blocks of the miss grid
	bool h_in_m = merged && capH && capM && m->haveH && m->haveM && nbH == m->gridH.nb && nbM == m->gridM.nb;
	for (int a = 0; a < 3 && h_in_m; ++a)
		h_in_m = m->gridH.base[a] >= m->gridM.base[a] &&
		         (long long)m->gridH.base[a] + 2ll * m->gridH.nb[a] <= (long long)m->gridM.base[a] + 2ll * m->gridM.nb[a];
	auto bound = [&](u64 nh, u64 nm) {
		if (h_in_m) return needBound(m, nh + nm, nbM, 1);
		Need b;
		if (nh) {
			Need hb = needBound(m, nh, nbH, 1);
			if (m->hit_tiles != ~0ull && m->g.L >= 4) {
				// (the hits' tiles have been counted, k_select: no more groups than that, no more blocks above than tiles per level)
				Need tb = needBound(m, m->hit_tiles, nbH, 1);
				hb.groups = std::min(hb.groups, m->hit_tiles + 1);
				hb.upper = std::min(hb.upper, tb.upper);
			}
			b += hb;
		}
		if (nm) b += needBound(m, nm, nbM, (u32)depth + 1);
		return b;
	};
	m->scan_new_bound = bound(capH, capM);
	// headroom_scans: a stream of pipelined scans keeps up to two more updates of this size in flight; size the
	// table for that now, so that the following updates can be enqueued without joining their predecessors
	for (u32 k = 0; k < headroom_scans; ++k) extra_used += m->scan_new_bound;
	if (tableTakes(m, extra_used + m->scan_new_bound)) return UFOMAP_OK;
	if (no_grow) return 1;
	// small tables simply grow to the pessimistic size once (cheap, and the fast check passes from then on);
	// the exact count is worth a kernel and a host round trip only when growing would cost hundreds of megabytes
	// (slots, not blocks: a tile group is 73 of them, whatever it holds)
	auto slotsFor = [&](const Need& nd) { return (m->used_g + nd.groups) * (u64)UFO_GROUP + m->used_u + nd.upper; };
	const bool cheap = slotsFor(extra_used + m->scan_new_bound) * 2 <= (1ull << 22);
	if (!cheap && (ent_h || ent_m)) {
		ScanCtl* ctl = m->b_ctl.as<ScanCtl>();
		u32* d_cnt = reinterpret_cast<u32*>(&ctl->dbg[60]);  // two spare words of the control block
		HIP_TRY(hipMemsetAsync(d_cnt, 0, 8, m->cs));
		if (merged) {
			// one list (ent_h, capacity capH + capM): a missing block lies in the hit grid or in the miss grid
			hipLaunchKernelGGL(k_count_missing, gridFor(capH + capM), dim3(256), 0, m->cs, m->t, ent_h, &ctl->n_entries[0],
			                   (u32)(capH + capM), d_cnt);
		} else {
			if (capH)
				hipLaunchKernelGGL(k_count_missing, gridFor(capH), dim3(256), 0, m->cs, m->t, ent_h, &ctl->n_entries[0], (u32)capH, d_cnt);
			if (capM)
				hipLaunchKernelGGL(k_count_missing, gridFor(capM), dim3(256), 0, m->cs, m->t, ent_m, &ctl->n_entries[1], (u32)capM, d_cnt + 1);
		}
		int rc = readCtl(m);
		if (rc) return rc;
		u32 cnt[2];
		memcpy(cnt, &m->h_ctl->dbg[60], 8);
		bool counted = false;
		if (merged) {
			if (m->h_ctl->n_entries[0] <= capH + capM) {
				m->scan_new_bound = h_in_m ? needBound(m, cnt[0], nbM, 1) : bound(capH ? cnt[0] : 0, capM ? cnt[0] : 0);
				counted = true;
			}
		} else if (m->h_ctl->n_entries[0] <= capH && m->h_ctl->n_entries[1] <= capM) {
			m->scan_new_bound = bound(cnt[0], cnt[1]);
			counted = true;
		}
		if (tableTakes(m, extra_used + m->scan_new_bound)) return UFOMAP_OK;
		// Still growing, and by a lot: count the distinct missing parents and grandparents of the level-1 entries exactly
		// (k_mark_parents) instead of bounding them by min(entries, cells of the box) -- the merged list inside one box, or
		// the hit list of an update at insert depth > 0 (3e5 hit voxels of a 2 mm frame lie in 1e4 tiles, not in 3e5)
		const bool one_box = merged && h_in_m;
		const bool hit_list = !merged && capH && m->haveH && nbH == m->gridH.nb;
		if (counted && (one_box || hit_list) && cnt[0] > (1u << 16)) {
			const Grid& gb = one_box ? m->gridM : m->gridH;
			const u64 n_list = one_box ? capH + capM : capH;
			ParentBox pb;
			u64 bits2 = 1, bits3 = 1;
			for (int a = 0; a < 3; ++a) {
				const long long b0 = (long long)(gb.base[a] >> 1), b1 = b0 + gb.nb[a] - 1;  // level-1 block coordinates of the box
				pb.lo2[a] = (i32)(b0 >> 1);
				pb.n2[a] = (u32)((b1 >> 1) - (b0 >> 1) + 1);
				pb.lo3[a] = (i32)(b0 >> 2);
				pb.n3[a] = (u32)((b1 >> 2) - (b0 >> 2) + 1);
				bits2 *= pb.n2[a];
				bits3 *= pb.n3[a];
			}
			if (bits2 <= (1ull << 32)) {  // (<= 512 MB of bitmap)
				const u64 w2 = (bits2 + 31) / 32, w3 = (bits3 + 31) / 32;
				DevBuf bm;
				HIP_TRY(bm.reserve((w2 + w3) * 4 + 64));
				HIP_TRY(hipMemsetAsync(bm.p, 0, (w2 + w3) * 4 + 64, m->cs));
				u32* d2 = bm.as<u32>();
				u32* d3 = d2 + w2;
				unsigned long long* d_out = reinterpret_cast<unsigned long long*>(d3 + w3 + ((w2 + w3) & 1));
				hipLaunchKernelGGL(k_mark_parents, gridFor(n_list), dim3(256), 0, m->cs, m->t, m->g, ent_h, &ctl->n_entries[0], (u32)n_list, pb, d2, d3);
				hipLaunchKernelGGL(k_popcount, gridFor(w2, 256, 4096), dim3(256), 0, m->cs, d2, w2, d_out);
				hipLaunchKernelGGL(k_popcount, gridFor(w3, 256, 4096), dim3(256), 0, m->cs, d3, w3, d_out + 1);
				unsigned long long h_out[2] = {0, 0};
				HIP_TRY(hipMemcpyAsync(h_out, d_out, 16, hipMemcpyDeviceToHost, m->cs));
				HIP_TRY(hipStreamSynchronize(m->cs));
				// level 1: the missing entries; level 2, 3: counted; above: no more than level 3 has, nor than fit in the box
				Need b;
				b.blocks = (u64)cnt[0] + h_out[0] + h_out[1] + 8;
				b.groups = h_out[1] + 1;
				b.upper = 8;
				for (u32 l = 4; l <= m->g.L; ++l) {
					const u64 c = std::min<u64>(h_out[1], levelBound(gb.nb, l - 1));
					b.blocks += c;
					b.upper += c;
				}
				if (hit_list && capM) b += needBound(m, cnt[1], nbM, (u32)depth + 1);  // (the coarse misses' own entries)
				m->scan_new_bound = needMin(m->scan_new_bound, b);
				if (tableTakes(m, extra_used + m->scan_new_bound)) return UFOMAP_OK;
			}
		}
	}
	return growFor(m, extra_used + m->scan_new_bound);
}

// Update lists from the two grids of the current scan into b_entries: hit entries first, then miss entries.
// capH/capM: capacities (upper bounds or guesses; the device-side counts land in ctl->n_entries[]).
// merged (insert depth 0): ONE list in which a block with hits and misses appears once (misses first, then
// the blocks that only received hits); its count is n_entries[0], its capacity capH + capM.
int extractLists(ufomap_map* m, u64 capH, u64 capM, bool zero_counts, bool merged = false)
{
	ScanCtl* ctl = m->b_ctl.as<ScanCtl>();
	if (capH > 0x7FFFFFFFull || capM > 0x7FFFFFFFull || capH + capM > 0x7FFFFFFFull)
		return fail(UFOMAP_ERR_CAPACITY, "update list exceeds 2^31 entries");
	// (+ 32 bytes per record for the colour section of a colour map's update list: ufomap_map_scan_keys_rgb)
	HIP_TRY(m->b_entries.reserve(((size_t)capH + capM + 1) * (sizeof(Entry) + (m->g.color ? 32u : 0u))));
	Entry* ent_h = m->b_entries.as<Entry>();
	Entry* ent_m = ent_h + capH;
	if (zero_counts) HIP_TRY(hipMemsetAsync(&ctl->n_entries[0], 0, 8, m->cs));  // otherwise zero from the control-block upload
	if (merged) {
		HitBlocks hb{capH ? m->b_hb_keys.as<u64>() : nullptr, m->b_hb_mask.as<u32>(), m->b_hb_time.as<u32>(), m->hb_cap_mask};
		if (capM && 2 == m->gridM.layout) {
			ProfScope ps(m, "k_extract_set");
			const u64 slots = m->miss_set_slots;
			const MissSet ms{m->b_gridM.as<u64>(), reinterpret_cast<u32*>(m->b_gridM.as<u64>() + slots), (u32)(slots - 1), nullptr};
			hipLaunchKernelGGL(k_extract_set<true>, gridFor(slots, 256, 2048), dim3(256), 0, m->cs, m->g, m->gridM, ms, 0u, ent_h, (u32)(capH + capM),
			                   ctl, hb);
		} else if (capM) {
			ProfScope ps(m, "k_extract");
			if (1 == m->gridM.layout)
				hipLaunchKernelGGL(k_extract_bits<true>, gridFor(m->gridM.bytes, 256, 2048), dim3(256), 0, m->cs, m->g, m->gridM,
				                   m->b_gridM.as<u32>(), 0u, ent_h, (u32)(capH + capM), ctl, hb);
			else
				hipLaunchKernelGGL(k_extract<true>, gridFor(m->gridM.bytes >> 2, 256, 2048), dim3(256), 0, m->cs, m->g, m->gridM,
				                   m->b_gridM.as<u32>(), 0u, ent_h, (u32)(capH + capM), ctl, hb);
		}
		if (capH) {
			ProfScope ps(m, "k_extract_hits");
			hipLaunchKernelGGL(k_extract_hits, dim3((u32)(((u64)m->hb_cap_mask + 1 + 2047) / 2048)), dim3(256), 0, m->cs, m->g, hb, ent_h,
			                   (u32)(capH + capM), ctl);
			m->hb_clean = m->hb_cap_mask + 1;
		}
		return UFOMAP_OK;
	}
	if (capH) {
		ProfScope ps(m, "k_extract_hits");
		HitBlocks hb{m->b_hb_keys.as<u64>(), m->b_hb_mask.as<u32>(), m->b_hb_time.as<u32>(), m->hb_cap_mask};
		hipLaunchKernelGGL(k_extract_hits, dim3((u32)(((u64)m->hb_cap_mask + 1 + 2047) / 2048)), dim3(256), 0, m->cs, m->g, hb, ent_h,
		                   (u32)capH, ctl);
		m->hb_clean = m->hb_cap_mask + 1;
	}
	if (capM && 2 == m->gridM.layout) {
		ProfScope ps(m, "k_extract_set");
		const u64 slots = m->miss_set_slots;
		const MissSet ms{m->b_gridM.as<u64>(), reinterpret_cast<u32*>(m->b_gridM.as<u64>() + slots), (u32)(slots - 1), nullptr};
		hipLaunchKernelGGL(k_extract_set<false>, gridFor(slots, 256, 2048), dim3(256), 0, m->cs, m->g, m->gridM, ms, 1u, ent_m, (u32)capM, ctl,
		                   HitBlocks{nullptr, nullptr, nullptr, 0});
	} else if (capM) {
		ProfScope ps(m, "k_extract");
		if (1 == m->gridM.layout)
			hipLaunchKernelGGL(k_extract_bits<false>, gridFor(m->gridM.bytes, 256, 2048), dim3(256), 0, m->cs, m->g, m->gridM,
			                   m->b_gridM.as<u32>(), 1u, ent_m, (u32)capM, ctl, HitBlocks{nullptr, nullptr, nullptr, 0});
		else
			hipLaunchKernelGGL(k_extract<false>, gridFor(m->gridM.bytes >> 2, 256, 2048), dim3(256), 0, m->cs, m->g, m->gridM,
			                   m->b_gridM.as<u32>(), 1u, ent_m, (u32)capM, ctl, HitBlocks{nullptr, nullptr, nullptr, 0});
	}
	return UFOMAP_OK;
}

// The map half of an integration (map stream): size the table, hits phase, then misses phase (OMB:1351-1365).
// The two update lists are already in b_entries (hit entries first); capH/capM are their capacities
// (true upper bounds or exact counts, so the device-side counts always fit).
// prev != nullptr: control block of the integration enqueued just before, not yet checked by the host (doInsert);
// returns 1 (nothing enqueued) if the table might have to grow while that one is in flight.
int mapPhase(ufomap_map* m, unsigned depth, const uint8_t* d_rgb, u64 capH, u64 capM, bool merged, const ScanCtl* prev = nullptr,
             Need extra_used = Need{}, u32 headroom_scans = 0)
{
	const float miss = (float)(m->g.miss_log / double((2.0 * depth) + 1));  // OMB:311
	m->hit_grid = false;
	Entry* ent_h = m->b_entries.as<Entry>();
	Entry* ent_m = ent_h + capH;
	int rc = sizeTable(m, ent_h, capH, m->gridH.nb, ent_m, capM, m->gridM.nb, depth, merged, extra_used, nullptr != prev, headroom_scans);
	if (rc) return rc;
	if (merged)
		return applyEntries(m, ent_h, (u32)(capH + capM), 0, 1, m->gridH.nb, m->g.hit, d_rgb, false, (u32)(capH + capM), 0u, (u32)capM,
		                    m->gridM.nb, miss, prev);
	rc = applyEntries(m, ent_h, (u32)capH, 0, 1, m->gridH.nb, m->g.hit, d_rgb, false, (u32)capH, (u32)capM);
	if (rc) return rc;
	return applyEntries(m, ent_m, (u32)capM, 1, (u32)depth + 1, m->gridM.nb, miss, nullptr, false, (u32)capH, (u32)capM);
}

int redoScan(ufomap_map* m);

#include "host_fast_path.inl"
#include "host_vol.inl"
int volWalkFinish(ufomap_map* m);
int finishPending(ufomap_map* m)
{
	if (!m->pending) return UFOMAP_OK;
	m->cs = m->stream;
	if (m->vol_walk) {
		// a walk of the volume path that an asynchronous call left enqueued: awaited here; if its reserve ran out the table is
		// exchanged and the tiles that stood back are run (nothing else is on the map stream: walks are enqueued one at a time)
		const int vrc = volWalkFinish(m);
		if (vrc) {
			m->pending = false;
			return vrc;
		}
	}
	m->pending = false;
	int rc = UFOMAP_OK;
	if (m->fast && 0 == m->h_res->err) {
		// k_ftail stored the finished control block in pinned memory itself and left the device copy in its start state
		memcpy(m->h_ctl, m->h_res, sizeof(ScanCtl));
		// (every scan of a walk carries the table's fill after the walk -- round 4 patched one of three words into the blocks of the
		// scans that were not the walk's last, and the host sized the next update as if the table were empty: ADVICE r4; counted here
		// so that a test can see it)
		if (0 == m->h_ctl->used_g_now && m->used_g > 0) ++m->n_fill_zero;
		m->used_est = m->h_ctl->used_now;
		m->used_g = m->h_ctl->used_g_now;
		m->used_u = m->h_ctl->used_u_now;
		m->ctl_clean = true;
		if (m->h_ctl->walk_scans) {  // (the last scan of a walk carries the number of scans the walk applied)
			++m->n_walks;
			m->n_walk_scans += m->h_ctl->walk_scans;
		}

	} else if (m->res_direct && m->done_by_flag && 0 == (m->h_res->err & ERR_NOT_STORED)) {
		// (setValueVolume's one-workgroup walk has stored the block, the table's fill included: k_vol_all)
		memcpy(m->h_ctl, m->h_res, offsetof(ScanCtl, dbg));
		m->used_est = m->h_ctl->used_now;
		m->used_g = m->h_ctl->used_g_now;
		m->used_u = m->h_ctl->used_u_now;
	} else {
		rc = readCtlDone(m);
	}
	m->res_direct = false;
	if (rc) return rc;
	drainEvents(m);
	if (m->h_ctl->err) m->prev_flagged = true;  // (sticky: doInsert resets it before a join)
	if (m->h_ctl->err && m->fast) m->first_dirty = true;  // the tree update stood back: it did not clean the set's scratch arrays
	m->fast = false;
	// Flagged and repeatable: a speculative scan that did not fit its predicted grid (or exceeded a bound derived from
	// it), or an update that stood back because the control block it looked at for its predecessor was flagged
	// (ERR_PREV). Nothing of it has reached the map; repeat it now, i.e. before any later update.
	if (m->h_ctl->err & ERR_GATE) {
		// A stream hand-over of the scan timed out: something keeps kernels of different streams from running side by side
		// (a profiler serialising them, fewer hardware queues than streams). Nothing of the scan has reached the map; it
		// is repeated below, and this handle hands over with events from now on.
		++m->n_gate_timeouts;
		m->opt_gates = 0;
		// (the scan half may still be running when a gate gives up: nothing of the set is reused or repeated before it has drained)
		HIP_TRY(hipStreamSynchronize(m->pstream));
		HIP_TRY(hipStreamSynchronize(m->sstream));
	}
	if (m->batch_world) {
		// a step of ufomap_map_insert_batch: a flagged scan of ANY rank made the walk stand back on every rank (all see the
		// same gathered control blocks), and every rank repeats the step here, at the same point of its sequence of calls
		if (m->h_ctl->err & (ERR_SPEC | ERR_PREV | ERR_GATE | ERR_RUNAWAY)) return redoBatchStep(m);
		if (0 == m->h_ctl->err) predictCommonGrid(m);
	}
	// The step's communicator has been looked at for the last time: the set does not carry it into its next integration. (Until round 6
	// it did: a set that had held a batch step and then took a scan of ufomap_map_insert predicted the RANKS' grid from it when it
	// was joined -- through a pointer that ufomap_comm_destroy may have freed by then -- and skipped the map's own prediction.
	// bench.py's batch_step_n1 leg followed by its host legs: one run in eight died of the corrupted heap.)
	const bool was_batch_step = 0 != m->batch_world;
	m->batch_world = 0;
	m->comm = nullptr;
	if (m->h_ctl->err && m->args.n && (m->args.spec || (m->h_ctl->err & (ERR_PREV | ERR_GATE)))) return redoScan(m);
	rc = ctlError(m);
	if (rc) return rc;
	m->counts[1] = m->h_ctl->n_rays;
	m->counts[3] = m->h_ctl->n_hits;
	if (m->args.n && !was_batch_step) predictGrid(m);  // (non-scan updates leave the prediction as it is)
	m->counts[5] = (u64)m->h_ctl->n_entries[0] + m->h_ctl->n_entries[1];
	m->counts[2] = m->h_ctl->n_steps;
	m->counts[6] = (u64)m->h_ctl->ph[0].n_new + m->h_ctl->ph[1].n_new;
	m->counts[7] = m->h_ctl->n_oob;
	for (int a = 0; a < 3; ++a) {
		double lo = decD(m->h_ctl->aabb_min[a]), hi = decD(m->h_ctl->aabb_max[a]);
		if (m->minmax_enabled && m->h_ctl->aabb_min[a] != ~0ull) {  // OMB:1367: only while enabled
			m->min_change[a] = std::min(m->min_change[a], lo);
			m->max_change[a] = std::max(m->max_change[a], hi);
		}
	}
	return UFOMAP_OK;
}

// The scan half of an integration (never touches the map): classify, de-duplicate, cast the rays into
// the dedup grids. On return *n_hits_out / *n_rays_out hold the unique hits / rays cast.
int scanPhase(ufomap_map* m, const double origin[3], const double* d_xyz, const uint8_t* d_rgb, size_t n, double max_range,
              unsigned depth, int discrete, int simple, unsigned early_stopping, u32* n_hits_out, u32* n_rays_out, bool spec = false)
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	HIP_TRY(hipSetDevice(m->device));
	if (depth >= m->g.L) return fail(UFOMAP_ERR_INVALID, "depth must be < depth_levels");
	if (d_rgb && !m->g.color) return fail(UFOMAP_ERR_INVALID, "coloured cloud into a non-colour map");
	if (d_rgb && !discrete)
		return fail(UFOMAP_ERR_UNSUPPORTED,
		            "OccupancyMapColor::insertPointCloud<PointCloudColor> does not compile in the reference (SURVEY.md 4)");
	if (n > 0x7FFFFFFFull) return fail(UFOMAP_ERR_INVALID, "more than 2^31 points");
	for (int k = 0; k < 8; ++k) m->counts[k] = 0;
	m->haveH = m->haveM = false;
	m->last_depth = depth;
	m->counts[0] = n;
	m->vol = false;
	if (0 == n) return UFOMAP_OK;
	m->scan_id += 1;
	const u32 N = (u32)n;
	const D3 sensor{origin[0], origin[1], origin[2]};

	// ---- scan phase ----------------------------------------------------------------------------
	HIP_TRY(m->b_pt_end.reserve(n * sizeof(D3)));
	HIP_TRY(m->b_pt_flag.reserve(n));
	HIP_TRY(m->b_pt_slot.reserve(n * 4));
	HIP_TRY(m->b_ray_end.reserve(n * sizeof(D3)));
	HIP_TRY(m->b_hit_code.reserve(n * 8));
	HIP_TRY(m->b_hit_pt.reserve(n * 4));
	// (at most n hit codes + n tile keys, + n ray cells when depth > 0, in >= 3 n (5 n) slots: load <= 0.67 (0.6) before the rounding up to
	// a power of two, 0.33-0.67 after; linear probing at 0.67 finds a key in 2 probes on average, hitHashInsert* report ERR_HASH_FULL
	// only when the table is FULL -- it cannot be: fewer keys than slots)
	u32 hcap = nextPow2(std::max<u64>(1024, (u64)n * 3 + (depth ? (u64)n * 2 : 0)));
	// keys and point indices in ONE buffer (keys first): one memset per scan instead of two
	HIP_TRY(m->b_hh_keys.reserve((size_t)hcap * 12));
	HIP_TRY(hipMemsetAsync(m->b_hh_keys.p, 0xFF, (size_t)hcap * 12, m->cs));
	HitHash hh{m->b_hh_keys.as<u64>(), reinterpret_cast<u32*>(m->b_hh_keys.as<u64>() + hcap), hcap - 1};
	m->hh_mask = hcap - 1;
	// control block
	ScanCtl init;
	memset(&init, 0, sizeof(init));
	for (int a = 0; a < 3; ++a) {
		init.mb_min[a] = init.hb_min[a] = INT32_MAX;
		init.mb_max[a] = init.hb_max[a] = INT32_MIN;
		init.aabb_min[a] = ~0ull;
		init.aabb_max[a] = 0ull;
	}
	*m->h_ctl = init;
	ScanCtl* ctl = m->b_ctl.as<ScanCtl>();
	HIP_TRY(hipMemcpyAsync(ctl, m->h_ctl, sizeof(ScanCtl), hipMemcpyHostToDevice, m->cs));
	m->ctl_clean = false;  // (the set's device control block no longer holds the fast path's start state)
	dim3 gp((N + 255) / 256);
	HIP_TRY(m->b_part0.reserve((size_t)gp.x * sizeof(BoxPartial)));
	HIP_TRY(m->b_part1.reserve((size_t)gp.x * sizeof(BoxPartial)));
	{
		ProfScope ps(m, "k_classify");
		if (discrete)
			hipLaunchKernelGGL(k_classify<true>, gp, dim3(256), 0, m->cs, m->g, sensor, d_xyz, N, max_range, (u32)depth,
			                   (u32)(d_rgb ? 1 : 0), hh, m->b_pt_end.as<D3>(), m->b_pt_flag.as<u8>(), m->b_pt_slot.as<u32>(),
			                   m->b_part0.as<BoxPartial>(), ctl, m->ing);
		else
			hipLaunchKernelGGL(k_classify<false>, gp, dim3(256), 0, m->cs, m->g, sensor, d_xyz, N, max_range, (u32)depth, 0u, hh,
			                   m->b_pt_end.as<D3>(), m->b_pt_flag.as<u8>(), m->b_pt_slot.as<u32>(), m->b_part0.as<BoxPartial>(), ctl,
			                   m->ing);
	}
	HIP_TRY(m->b_blk_range.reserve((size_t)gp.x * 8));
	u32* ray_pt = nullptr;  // early stopping: the rays' ranks in the cloud's order
	const u32 es_cells = (early_stopping && discrete && depth > 0) ? 1u : 0u;
	if (early_stopping) {
		HIP_TRY(m->b_ray_pt.reserve(n * 4));
		ray_pt = m->b_ray_pt.as<u32>();
		if (es_cells)  // (insert depth > 0: the ray of a cell is its first point's)
			hipLaunchKernelGGL(k_es_raycells<true>, gp, dim3(256), 0, m->cs, m->g, sensor, N, (u32)depth, hh, m->b_pt_end.as<D3>(), m->b_pt_flag.as<u8>(),
			                   m->b_pt_slot.as<u32>(), ctl);
	}
	{
		ProfScope ps(m, "k_select");
		if (discrete)
			hipLaunchKernelGGL(k_select<true>, gp, dim3(256), 0, m->cs, m->g, sensor, N, (u32)depth, hh, m->b_pt_end.as<D3>(),
			                   m->b_pt_flag.as<u8>(), m->b_pt_slot.as<u32>(), m->b_ray_end.as<D3>(), m->b_hit_code.as<u64>(),
			                   m->b_hit_pt.as<u32>(), m->b_part1.as<BoxPartial>(), ctl, m->b_blk_range.as<u32>(), ray_pt, es_cells);
		else
			hipLaunchKernelGGL(k_select<false>, gp, dim3(256), 0, m->cs, m->g, sensor, N, (u32)depth, hh, m->b_pt_end.as<D3>(),
			                   m->b_pt_flag.as<u8>(), m->b_pt_slot.as<u32>(), m->b_ray_end.as<D3>(), m->b_hit_code.as<u64>(),
			                   m->b_hit_pt.as<u32>(), m->b_part1.as<BoxPartial>(), ctl, m->b_blk_range.as<u32>(), ray_pt, es_cells);
	}
	{
		ProfScope ps(m, "k_reduce_boxes");
		hipLaunchKernelGGL(k_reduce_boxes, dim3(1), dim3(256), 0, m->cs, m->b_part1.as<BoxPartial>(), gp.x, (u32)(discrete ? 0 : 1),
		                   m->b_part0.as<BoxPartial>(), ctl, m->spec_grid, (u32)(spec ? 1 : 0));
	}
	HIP_TRY(hipGetLastError());
	u32 n_rays, n_hits;
	m->hit_tiles = ~0ull;
	if (spec) {
		// No read-back: the rest of the scan is enqueued on the grid predicted from the previous scan, with the
		// point count as the bound of rays and hits; k_reduce_boxes has flagged the scan (ERR_SPEC) if its box
		// does not fit, every later kernel of a flagged scan leaves the map alone, and the join repeats it.
		n_rays = n_hits = N;
		m->haveH = m->haveM = true;
		m->gridM = m->spec_grid;
		m->gridH = m->spec_grid;  // the hits are ray ends: same box (only used for bounds)
	} else {
	HIP_TRY(hipMemcpyAsync(m->h_ctl, m->b_ctl.p, sizeof(ScanCtl), hipMemcpyDeviceToHost, m->cs));
	HIP_TRY(hipStreamSynchronize(m->cs));
	int rc = ctlError(m);
	if (rc) return rc;
	n_rays = m->h_ctl->n_rays;
	n_hits = m->h_ctl->n_hits;
	m->counts[1] = n_rays;
	m->counts[3] = n_hits;
	m->hit_tiles = m->g.L <= 19 ? (u64)m->h_ctl->n_hit_tiles : ~0ull;

	// ---- dedup grids -----------------------------------------------------------------------------
	i32 hmn[3], hmx[3], mmn[3], mmx[3];
	for (int a = 0; a < 3; ++a) {
		hmn[a] = m->h_ctl->hb_min[a];
		hmx[a] = m->h_ctl->hb_max[a];
		mmn[a] = m->h_ctl->mb_min[a];
		mmx[a] = m->h_ctl->mb_max[a];
	}
	m->haveH = n_hits > 0;
	m->haveM = n_rays > 0;
	if (m->haveH && makeGrid(hmn, hmx, 0, &m->gridH))
		return fail(UFOMAP_ERR_CAPACITY, "hit bounding box too large for the scan grid");
	if (m->haveM && makeGrid(mmn, mmx, (u32)depth, &m->gridM))
		return fail(UFOMAP_ERR_CAPACITY, "ray bounding box too large for the scan grid (runaway ray?)");
	}
	if (!spec && m->haveM && n_rays) {
		// a ray box far beyond the steady-state path's grids (a 2 mm RGB-D frame): the volume path -- brick grids, the tiled tree update
		VolPlan vp;
		if (volPlan(m, m->gridM, depth, simple, early_stopping, d_rgb, &vp)) {
			m->vol_rgb = d_rgb;
			const int vrc = volScan(m, sensor, vp, n_hits, n_rays);
			if (vrc < 0) return vrc;
			if (0 == vrc) {
				*n_hits_out = n_hits;
				*n_rays_out = n_rays;
				return UFOMAP_OK;
			}
		}
	}
	if (!spec && m->haveM && !simple && 0 == early_stopping && m->opt_dda_seg != 0 && m->opt_bits != 0 && m->opt_dda_mode <= 0) {
		// one bit per cell (rows padded to 32 cells) when that fits in LDS: the fast walk kernel (k_walk)
		Grid& gr = m->gridM;
		const bool packed = 2 * gr.nb[0] < 1023 && 2 * gr.nb[1] < 1023 && 2 * gr.nb[2] < 1023;
		const u64 bytes1 = (u64)(gridRowBits(gr) >> 3) * (2ull * (u64)gr.nb[1]) * (2ull * (u64)gr.nb[2]);
		// ... and, with the marks going to the global grid through an LDS filter (k_cast<true>), for every grid the packed
		// coordinates can address (< 1023 cells per axis, e.g. 8 cm / 20 m)
		if (packed && (bytes1 <= UFO_DDA_LDSGRID_MAX || (m->opt_cast != 0 && m->opt_cast_global != 0 && bytes1 < (1ull << 28)))) {
			gr.layout = 1;
			gr.bytes = (bytes1 + 15) & ~15ull;
		}
	}
	u64 gbytes = m->haveM ? m->gridM.bytes : 0;  // hits are grouped through a hash, only grid M is dense
	if (early_stopping && m->haveM) {
		// (a dense array over the cells of the ray box: who visits a cell first)
		// (round 6: a box whose dense array does not fit takes the sparse form -- a hash of the cells the rays visit, scan_kernels.h: EsArgs;
		// what still has to fit is grid M itself, one byte per node block of the box)
		if (gbytes > m->scratch_limit)
			return fail(UFOMAP_ERR_UNSUPPORTED, "early_stopping > 0 on a ray box whose dedup grid needs " + std::to_string(gbytes) +
			                                        " bytes > scratch limit (ufomap_map_set_scratch_limit)");
	} else
	if (gbytes > m->scratch_limit || (m->opt_sparse_set && m->haveM && !simple)) {  // (option sparse_set: tests)
		// the box is too large for a dense grid: the ray cells go through a hash set of node blocks instead (Grid::layout 2,
		// scan_kernels.h: MissSet) -- bounded by the cells the rays touch, as the reference's CodeMap is
		if (simple)
			return fail(UFOMAP_ERR_CAPACITY, "scan dedup grid needs " + std::to_string(gbytes) + " bytes > scratch limit " +
			                                     std::to_string(m->scratch_limit) + " (ufomap_map_set_scratch_limit; simple ray casting has no sparse fallback)");
		m->gridM.layout = 2;
	}
	if (m->haveH) {
		if (n > (1u << 29)) return fail(UFOMAP_ERR_INVALID, "more than 2^29 points in one scan");
		const u32 hbcap = nextPow2(std::max<u64>(256, (u64)n_hits * 2));
		m->hb_cap_mask = hbcap - 1;
		const size_t p0 = m->b_hb_keys.cap, p1 = m->b_hb_mask.cap, p2 = m->b_hb_time.cap;  // (a re-allocation may return the old address: compare sizes)
		HIP_TRY(m->b_hb_keys.reserve((size_t)hbcap * 8));
		HIP_TRY(m->b_hb_mask.reserve((size_t)hbcap * 4));
		HIP_TRY(m->b_hb_time.reserve((size_t)hbcap * 4));
		if (p0 != m->b_hb_keys.cap || p1 != m->b_hb_mask.cap || p2 != m->b_hb_time.cap) m->hb_clean = 0;
		// k_extract_hits leaves the slots it read empty: only slots never used before need the memsets
		if (hbcap > m->hb_clean) {
			const size_t from = m->hb_clean, cnt = hbcap - m->hb_clean;
			HIP_TRY(hipMemsetAsync(m->b_hb_keys.as<u64>() + from, 0xFF, cnt * 8, m->cs));
			HIP_TRY(hipMemsetAsync(m->b_hb_mask.as<u32>() + from, 0, cnt * 4, m->cs));
			HIP_TRY(hipMemsetAsync(m->b_hb_time.as<u32>() + from, 0, cnt * 4, m->cs));
		}
		m->hb_clean = 0;  // dirty until this scan's k_extract_hits has been enqueued (extractLists)
		ProfScope ps(m, "k_hitmark");
		HitBlocks hb{m->b_hb_keys.as<u64>(), m->b_hb_mask.as<u32>(), m->b_hb_time.as<u32>(), m->hb_cap_mask};
		hipLaunchKernelGGL(k_hitmark, gridFor(n_hits), dim3(256), 0, m->cs, hb, m->b_hit_code.as<u64>(), m->b_hit_pt.as<u32>(), ctl, ctl);
	}
	if (m->haveM && early_stopping) {
		// ---- early stopping (scan_kernels.h: k_es_*): the rays' stops as the fixed point of "who visits a cell first", then the
		// visited cells into grid M (one byte per node block) ----
		const u64 cells = 8ull * (u64)m->gridM.nb[0] * (u64)m->gridM.nb[1] * (u64)m->gridM.nb[2];
		HIP_TRY(m->b_gridM.reserve(m->gridM.bytes));
		// who visits a cell first: a dense array over the box's cells, or -- when that does not fit (4 bytes per cell) -- a hash of the
		// cells the rays visit, doubled and the round repeated when it fills up (its size is remembered for later scans)
		const bool es_sparse = cells >= (1ull << 32) || cells * 4 + m->gridM.bytes > m->scratch_limit || 0 != m->opt_es_sparse;
		u64 es_slots = 0;
		if (es_sparse) {
			es_slots = std::max<u64>(m->es_set_slots, std::max<u64>(1ull << 16, nextPow2((u64)n_rays * 32)));
			if (2 == m->opt_es_sparse) es_slots = 1ull << 10;  // (tests: the set has to grow inside the round)
			if (es_slots * 12 + m->gridM.bytes > m->scratch_limit && es_slots > (1ull << 16))
				return fail(UFOMAP_ERR_UNSUPPORTED, "early_stopping > 0: the set of first rays does not fit the scratch limit (ufomap_map_set_scratch_limit)");
			HIP_TRY(m->b_es_first.reserve(es_slots * 12));
		} else HIP_TRY(m->b_es_first.reserve(cells * 4));
		HIP_TRY(m->b_es_stop.reserve((size_t)n_rays * 4));
		HIP_TRY(hipMemsetAsync(m->b_gridM.p, 0, m->gridM.bytes, m->cs));
		HIP_TRY(hipMemsetAsync(m->b_es_stop.p, 0xFF, (size_t)n_rays * 4, m->cs));
		u32* d_changed = m->b_ctl.as<u32>() + (sizeof(ScanCtl) + 3) / 4;  // (a spare word behind the control block)
		EsArgs ea{es_sparse ? nullptr : m->b_es_first.as<u32>(), es_sparse ? m->b_es_first.as<u64>() : nullptr,
		          es_sparse ? reinterpret_cast<u32*>(m->b_es_first.as<u64>() + es_slots) : nullptr, es_sparse ? (u32)(es_slots - 1) : 0u, m->b_es_stop.as<u32>(),
		          m->b_ray_pt.as<u32>(), (u32)early_stopping, simple ? 1u : 0u};
		const dim3 gr_((n_rays + 255) / 256);
		u32 rounds = 0;
		for (;; ++rounds) {
			if (rounds > n_rays + 2) return fail(UFOMAP_ERR_DEVICE, "early stopping: the rays' stops did not settle (internal error)");
			for (;;) {  // (the marking pass; sparse form: again with a set twice the size when it filled up)
				if (es_sparse) HIP_TRY(hipMemsetAsync(m->b_es_first.p, 0xFF, es_slots * 12, m->cs));
				else HIP_TRY(hipMemsetAsync(m->b_es_first.p, 0xFF, cells * 4, m->cs));
				HIP_TRY(hipMemsetAsync(d_changed, 0, 4, m->cs));
				{
					ProfScope ps(m, "k_es_mark");
					hipLaunchKernelGGL(k_es_mark, gr_, dim3(256), 0, m->cs, m->g, sensor, (u32)depth, m->gridM, ea, m->b_ray_end.as<D3>(), ctl, ctl, (u32*)nullptr);
				}
				if (!es_sparse) break;
				u32 eflags = 0;
				HIP_TRY(hipMemcpyAsync(&eflags, &ctl->err, 4, hipMemcpyDeviceToHost, m->cs));
				HIP_TRY(hipStreamSynchronize(m->cs));
				if (!(eflags & ERR_ENTRIES)) break;
				hipLaunchKernelGGL(k_ctl_clear, dim3(1), dim3(1), 0, m->cs, ctl, (u32)ERR_ENTRIES);
				es_slots *= 2;
				if (es_slots * 12 + m->gridM.bytes > m->scratch_limit || es_slots > (1ull << 31))
					return fail(UFOMAP_ERR_UNSUPPORTED, "early_stopping > 0: the set of first rays does not fit the scratch limit (ufomap_map_set_scratch_limit)");
				HIP_TRY(m->b_es_first.reserve(es_slots * 12));
				ea.hkeys = m->b_es_first.as<u64>();
				ea.hvals = reinterpret_cast<u32*>(ea.hkeys + es_slots);
				ea.hmask = (u32)(es_slots - 1);
			}
			m->es_set_slots = es_sparse ? es_slots : m->es_set_slots;
			{
				ProfScope ps(m, "k_es_stops");
				hipLaunchKernelGGL(k_es_stops, gr_, dim3(256), 0, m->cs, m->g, sensor, (u32)depth, m->gridM, ea, m->b_ray_end.as<D3>(), ctl, ctl, d_changed);
			}
			u32 changed = 0;
			HIP_TRY(hipMemcpyAsync(&changed, d_changed, 4, hipMemcpyDeviceToHost, m->cs));
			HIP_TRY(hipStreamSynchronize(m->cs));
			if (0 == changed) break;
		}
		m->es_rounds = rounds + 1;
		{
			ProfScope ps(m, "k_es_mark");
			hipLaunchKernelGGL(k_es_mark, gr_, dim3(256), 0, m->cs, m->g, sensor, (u32)depth, m->gridM, ea, m->b_ray_end.as<D3>(), ctl, ctl, m->b_gridM.as<u32>());
		}
	} else if (m->haveM && 2 == m->gridM.layout) {
		// sparse: walk into the set; if it fills up, double it and walk again (the size is remembered for later scans)
		for (;;) {
			const u64 slots = std::max<u64>(m->miss_set_slots, 1ull << 16);
			if (slots * 12 > m->scratch_limit && slots > (1ull << 16))
				return fail(UFOMAP_ERR_CAPACITY, "the scan's ray cells do not fit the scratch limit even as a sparse set (ufomap_map_set_scratch_limit)");
			m->miss_set_slots = slots;
			HIP_TRY(m->b_gridM.reserve(slots * 12 + 16));
			u64* keys = m->b_gridM.as<u64>();
			u32* mask = reinterpret_cast<u32*>(keys + slots);
			u32* count = mask + slots;
			HIP_TRY(hipMemsetAsync(keys, 0xFF, slots * 8, m->cs));
			HIP_TRY(hipMemsetAsync(mask, 0, slots * 4 + 16, m->cs));
			m->gridM.bytes = slots * 12;
			const MissSet ms{keys, mask, (u32)(slots - 1), count};
			{
				ProfScope ps(m, "k_dda_set");
				hipLaunchKernelGGL(k_dda_set, dim3((n_rays + 255) / 256), dim3(256), 0, m->cs, m->g, sensor, (u32)depth, m->gridM, ms,
				                   m->b_ray_end.as<D3>(), ctl, ctl);
			}
			HIP_TRY(hipMemcpyAsync(m->h_ctl, m->b_ctl.p, sizeof(ScanCtl), hipMemcpyDeviceToHost, m->cs));
			HIP_TRY(hipMemcpyAsync(&m->miss_set_count, count, 4, hipMemcpyDeviceToHost, m->cs));
			HIP_TRY(hipStreamSynchronize(m->cs));
			if (!(m->h_ctl->err & ERR_ENTRIES)) break;
			if (slots >= (1ull << 31)) return fail(UFOMAP_ERR_CAPACITY, "sparse ray-cell set would exceed 2^31 node blocks");
			m->miss_set_slots = slots * 2;
			// the flag and the step count of the abandoned walk go; what the head kernels left in the control block stays
			hipLaunchKernelGGL(k_ctl_clear, dim3(1), dim3(1), 0, m->cs, ctl, (u32)ERR_ENTRIES);
			HIP_TRY(hipMemsetAsync(&ctl->n_steps, 0, 8, m->cs));
			HIP_TRY(hipMemsetAsync(&ctl->n_oob, 0, 4, m->cs));
		}
		const int erc = ctlError(m);
		if (erc) return erc;
	} else if (m->haveM) {
		HIP_TRY(m->b_gridM.reserve(m->gridM.bytes));
		int mode = m->gridM.bytes <= UFO_DDA_LDSGRID_MAX ? DDA_LDSGRID : (m->gridM.bytes < (1ull << 29) ? DDA_FILTER : DDA_DIRECT);
		if (m->opt_dda_mode >= 0 && m->opt_dda_mode >= mode) mode = m->opt_dda_mode;  // tests may force a more general mode
		// bit grid beyond LDS (or option cast_global = 2, for tests): k_cast<true> marks the global grid directly
		const bool castg = 1 == m->gridM.layout && m->opt_cast != 0 && (m->gridM.bytes > UFO_DDA_LDSGRID_MAX || m->opt_cast_global >= 2);
		if (castg) mode = DDA_FILTER;  // (the grid is cleared below, nothing is merged afterwards)
		const size_t lds = mode == DDA_LDSGRID ? (size_t)m->gridM.bytes : (mode == DDA_FILTER ? (size_t)UFO_DDA_FILT * 4 : 0);
		// Launch shape (measured on C2, 36.6 k rays, LDS-grid mode; lanes per ray x threads per workgroup ->
		// walk kernel): 1x256 33 us, 2x512 34, 2x256 39, 1x512 42, 2x1024 43, 4x1024 47, 4x512 51, 1x128 52,
		// 1x1024 63. A ray (segment) is one dependent instruction chain, so few waves per SIMD hide little;
		// beyond one workgroup per CU the per-workgroup cost of the LDS-grid copy (zero, store, merge) takes
		// over; and a segment lane replays its start state (k0 additions), which makes 4 lanes per ray a loss
		// unless the scan is tiny. Hence: about one workgroup per CU, of 256..1024 threads, 1-2 lanes per ray.
		const bool packed = 2 * m->gridM.nb[0] < 1023 && 2 * m->gridM.nb[1] < 1023 && 2 * m->gridM.nb[2] < 1023;
		const bool seg = !simple && mode != DDA_DIRECT && packed && m->opt_dda_seg != 0;
		u32 seg_shift = 0;
		if (seg) seg_shift = n_rays <= 8192u ? 2u : (n_rays <= 65536u ? 1u : 0u);
		if (seg && m->opt_dda_lanes > 0) seg_shift = m->opt_dda_lanes == 1 ? 0 : (m->opt_dda_lanes == 2 ? 1 : 2);
		const u64 lanes = (u64)n_rays << seg_shift;
		u32 blk = (u32)std::min<u64>(UFO_DDA_BLOCK, std::max<u64>(256, (((lanes + 255) / 256) + 63) & ~63ull));
		if (lanes < 256u * 32u) blk = 128;  // tiny scans: spread over more CUs
		if (m->opt_dda_block >= 64 && m->opt_dda_block <= UFO_DDA_BLOCK) blk = (u32)m->opt_dda_block;
		dim3 gr((u32)((lanes + blk - 1) / blk));
		u32* dda_out = m->b_gridM.as<u32>();
		if (mode == DDA_LDSGRID) {
			HIP_TRY(m->b_slabs.reserve((size_t)gr.x * m->gridM.bytes + (size_t)gr.x * 8));
			dda_out = m->b_slabs.as<u32>();
		} else {
			HIP_TRY(hipMemsetAsync(m->b_gridM.p, 0, m->gridM.bytes, m->cs));
		}
		// fused set-up + segment queue + walk (k_cast) when the bit grid and its queue fit in LDS
		const bool cast = !castg && 1 == m->gridM.layout && m->opt_cast != 0 && m->gridM.bytes + UFO_CAST_LDS_EXTRA <= (160u << 10) - 512u;
		if (castg) {
			const u32 n_blk = (u32)((n + 255) / 256);  // workgroups of k_select
			u32 nwg = m->opt_cast_wgs > 0 ? (u32)m->opt_cast_wgs : std::min<u32>(n_blk, 1024u);
			nwg = std::max<u32>(1u, std::min<u32>(nwg, (n_rays + 63u) / 64u));
			HIP_TRY(m->b_slabs.reserve((size_t)std::max(nwg, n_blk) * 8));
			ProfScope ps(m, "k_cast_global");
			// LDS: a workgroup's box of the grid (at least the fallback's filter), beside the rounds' queue and constants
			const u32 grid_lds = ((160u << 10) - 1024u - (u32)UFO_CAST2_LDS_EXTRA) & ~15u;
			if (3 == m->opt_cast_global) {  // (tests: marks one by one for every workgroup)
				hipLaunchKernelGGL(k_cast<1>, dim3(nwg), dim3(512), ((size_t)4u << UFO_CAST_FILT_LOG2) + UFO_CAST_LDS_EXTRA, m->cs, m->g, sensor,
				                   (u32)depth, m->gridM, m->b_gridM.as<u32>(), m->b_ray_end.as<D3>(), (u32)std::max(8, m->opt_cast_k), ctl, ctl,
				                   m->b_slabs.as<unsigned long long>(), 0u, (const u32*)nullptr, 0u, 0u);
			} else {
				nwg = std::max<u32>(nwg, (n_blk + UFO_CAST_STRETCHES - 1u) / UFO_CAST_STRETCHES);
				hipLaunchKernelGGL(k_cast<2>, dim3(nwg), dim3(512), (size_t)grid_lds + UFO_CAST2_LDS_EXTRA, m->cs, m->g, sensor, (u32)depth, m->gridM,
				                   m->b_gridM.as<u32>(), m->b_ray_end.as<D3>(), (u32)std::max(8, m->opt_cast_k), ctl, ctl,
				                   m->b_slabs.as<unsigned long long>(), grid_lds, m->b_blk_range.as<u32>(), n_blk, 4 == m->opt_cast_global ? 4096u : grid_lds);
			}
		} else if (cast) {
			const u32 cblk = (m->opt_dda_block >= 256 && m->opt_dda_block <= 512) ? (u32)m->opt_dda_block : 512u;
			u32 nwg = m->opt_cast_wgs > 0 ? (u32)m->opt_cast_wgs : 256u;
			nwg = std::max<u32>(1u, std::min<u32>(nwg, (n_rays + 63u) / 64u));
			HIP_TRY(m->b_slabs.reserve((size_t)nwg * m->gridM.bytes + (size_t)nwg * 8));
			unsigned long long* sp = reinterpret_cast<unsigned long long*>(m->b_slabs.as<char>() + (size_t)nwg * m->gridM.bytes);
			{
				ProfScope ps(m, "k_cast");
				hipLaunchKernelGGL(k_cast<0>, dim3(nwg), dim3(cblk), (size_t)m->gridM.bytes + UFO_CAST_LDS_EXTRA, m->cs, m->g, sensor, (u32)depth,
				                   m->gridM, m->b_slabs.as<u32>(), m->b_ray_end.as<D3>(), (u32)std::max(8, m->opt_cast_k), ctl, ctl, sp, 0u, (const u32*)nullptr, 0u, 0u);
			}
			ProfScope ps(m, "k_merge_slabs");
			const u32 n4 = (u32)(m->gridM.bytes >> 4);
			hipLaunchKernelGGL(k_merge_slabs, dim3(std::min<u32>((n4 + 63) / 64, 1024)), dim3(1024), 0, m->cs, m->b_slabs.as<uint4>(), nwg, n4,
			                   m->b_gridM.as<uint4>(), sp, ctl);
		} else {
		if (seg) {
			HIP_TRY(m->b_rays.reserve((size_t)n_rays * sizeof(RayState)));
			ProfScope ps(m, "k_ray_setup");
			hipLaunchKernelGGL(k_ray_setup, dim3((n_rays + 255) / 256), dim3(256), 0, m->cs, m->g, sensor, (u32)depth, m->gridM,
			                   m->b_ray_end.as<D3>(), m->b_rays.as<RayState>(), ctl);
		}
		{
		ProfScope ps(m, 1 == m->gridM.layout ? "k_walk" : (seg ? "k_dda_seg" : "k_dda"));
#define UFO_LAUNCH_DDA(SIMPLE, MODE)                                                                                         \
	hipLaunchKernelGGL((k_dda<SIMPLE, MODE>), gr, dim3(blk), lds, m->cs, m->g, sensor, (u32)depth, m->gridM, \
	                   dda_out, m->b_ray_end.as<D3>(), ctl, ctl)
#define UFO_LAUNCH_SEG(MODE)                                                                                    \
	hipLaunchKernelGGL((k_dda_seg<MODE>), gr, dim3(blk), lds, m->cs, m->g, (u32)depth, m->gridM, dda_out, \
	                   m->b_rays.as<RayState>(), seg_shift, ctl, ctl)
		if (1 == m->gridM.layout) {
			// (layout 1 is only chosen when seg is possible and the bit grid fits in LDS)
			hipLaunchKernelGGL(k_walk, gr, dim3(blk), lds, m->cs, m->g, (u32)depth, m->gridM, dda_out, m->b_rays.as<RayState>(), seg_shift,
			                   ctl, ctl,
			                   reinterpret_cast<unsigned long long*>(m->b_slabs.as<char>() + (size_t)gr.x * m->gridM.bytes));
		} else if (seg) {
			if (mode == DDA_LDSGRID) UFO_LAUNCH_SEG(DDA_LDSGRID);
			else UFO_LAUNCH_SEG(DDA_FILTER);
		} else if (simple) {
			if (mode == DDA_LDSGRID) UFO_LAUNCH_DDA(true, DDA_LDSGRID);
			else if (mode == DDA_FILTER) UFO_LAUNCH_DDA(true, DDA_FILTER);
			else UFO_LAUNCH_DDA(true, DDA_DIRECT);
		} else {
			if (mode == DDA_LDSGRID) UFO_LAUNCH_DDA(false, DDA_LDSGRID);
			else if (mode == DDA_FILTER) UFO_LAUNCH_DDA(false, DDA_FILTER);
			else UFO_LAUNCH_DDA(false, DDA_DIRECT);
		}
#undef UFO_LAUNCH_DDA
#undef UFO_LAUNCH_SEG
		}
		if (mode == DDA_LDSGRID) {
			ProfScope ps(m, "k_merge_slabs");
			const u32 n4 = (u32)(m->gridM.bytes >> 4);
			hipLaunchKernelGGL(k_merge_slabs, dim3(std::min<u32>((n4 + 63) / 64, 1024)), dim3(1024), 0, m->cs, m->b_slabs.as<uint4>(), gr.x,
			                   n4, m->b_gridM.as<uint4>(),
			                   1 == m->gridM.layout ? reinterpret_cast<const unsigned long long*>(m->b_slabs.as<char>() + (size_t)gr.x * m->gridM.bytes)
			                                        : (const unsigned long long*)nullptr,
			                   ctl);
		}
		}  // !cast
	}

	*n_hits_out = n_hits;
	*n_rays_out = n_rays;
	HIP_TRY(hipGetLastError());
	return UFOMAP_OK;
}

// Size the update-list buffers and extract both lists on the scan stream. The capacities are true upper
// bounds (hit blocks <= unique hits; miss blocks <= blocks of grid M) or, for huge sparse grids, the exact
// count from a counting pass -- the device-side counts always fit.
int extractPhase(ufomap_map* m, u32 n_hits, u32 n_rays, u64* capH_out, u64* capM_out, bool merged)
{
	ScanCtl* ctl = m->b_ctl.as<ScanCtl>();
	u64 capH = m->haveH ? (u64)n_hits : 0;
	u64 capM = m->haveM ? (2 == m->gridM.layout ? (u64)m->miss_set_count : m->gridM.bytes) : 0;
	const u64 guess = m->opt_entry_guess ? m->opt_entry_guess : std::max<u64>(1u << 20, (u64)n_rays * 64);
	if (capM > guess && 2 != m->gridM.layout) {
		// counting pass (cap = 0 writes nothing), then the exact size
		ProfScope ps(m, "k_extract_count");
		if (1 == m->gridM.layout)
			hipLaunchKernelGGL(k_extract_bits<false>, gridFor(m->gridM.bytes, 256, 2048), dim3(256), 0, m->cs, m->g, m->gridM,
			                   m->b_gridM.as<u32>(), 1u, (Entry*)nullptr, 0u, ctl, HitBlocks{nullptr, nullptr, nullptr, 0});
		else
			hipLaunchKernelGGL(k_extract<false>, gridFor(m->gridM.bytes >> 2, 256, 2048), dim3(256), 0, m->cs, m->g, m->gridM,
			                   m->b_gridM.as<u32>(), 1u, (Entry*)nullptr, 0u, ctl, HitBlocks{nullptr, nullptr, nullptr, 0});
		HIP_TRY(hipMemcpyAsync(m->h_ctl, m->b_ctl.p, sizeof(ScanCtl), hipMemcpyDeviceToHost, m->cs));
		HIP_TRY(hipStreamSynchronize(m->cs));
		capM = m->h_ctl->n_entries[1];
		HIP_TRY(hipMemsetAsync(&ctl->n_entries[1], 0, 4, m->cs));
	}
	int rc = extractLists(m, capH, capM, false, merged);
	if (rc) return rc;
	*capH_out = capH;
	*capM_out = capM;
	return UFOMAP_OK;
}

// finish (read back, check, repeat if flagged) the integration of hand-over set alt[k]; its kernels must be complete
int finishSet(ufomap_map* m, int k)
{
	if (!m->alt[k].pending) return UFOMAP_OK;
	swapWith(m, m->alt[k]);
	int rc = finishPending(m);
	swapWith(m, m->alt[k]);
	return rc;
}

// join the integrations enqueued by earlier calls, oldest first (the current set's own is not touched)
int joinOlder(ufomap_map* m)
{
	if (oldestPendingAlt(m) < 0) return UFOMAP_OK;
	int rc = flushDeferred(m);  // (what waited for company is enqueued now)
	if (rc) return rc;
	HIP_TRY(hipStreamSynchronize(m->stream));
	for (int k; (k = oldestPendingAlt(m)) >= 0;) {
		const int r = finishSet(m, k);
		if (!rc) rc = r;
	}
	if (rc && UFOMAP_OK == m->async_status) m->async_status = rc;
	return rc;
}

int doInsert(ufomap_map* m, const double origin[3], const double* d_xyz, const uint8_t* d_rgb, size_t n, double max_range,
             unsigned depth, int discrete, int simple, unsigned early_stopping, int async, bool swapped, int spec_mode = 0, int caller_dev = -1)
{
	// caller_dev: the cloud lies in device memory that belongs to the caller (default: whenever the set was not rotated in by
	// an upload, i.e. ufomap_map_insert_device). Such a cloud is consumed before this call returns, asynchronous or not.
	const bool caller_owned = caller_dev < 0 ? !swapped : 0 != caller_dev;
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	HIP_TRY(hipSetDevice(m->device));
	// The scan half never reads the map (freeSpace is const, OMB:1230-1232): it runs on the scan stream
	// while the map half of the previous integration may still be updating the tree on the map stream --
	// the reference overlaps its head loop with the previous integration the same way (the join sits
	// after the head loop, OMB:315). Hand-over buffers are double-buffered and swapped here.
	const auto t_begin = std::chrono::steady_clock::now();
	struct Total {
		ufomap_map* m;
		std::chrono::steady_clock::time_point t0;
		~Total() { m->host_ns[3] += (u64)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
	} total_timer{m, t_begin};
	auto lap = [&](int k, std::chrono::steady_clock::time_point since) {
		m->host_ns[k] += (u64)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - since).count();
	};
	if (m->poisoned) return fail(UFOMAP_ERR_CAPACITY, "the map is inconsistent after a node table overflow: ufomap_map_clear it");
	if (!swapped) (void)rotateSets(m);
	m->seq = ++m->latest_seq;
	if (m->chg_enabled) async = 0;  // the change log is sized between updates: one update at a time
	m->cs = m->sstream;
	// insert depth 0: hits and misses of the scan as ONE pass over the tree (map_kernels.h, k_apply_leaf mode 2)
	const bool merged = 0 == depth && 0 != m->opt_merge;
	// Speculation: enqueue the whole scan on the grid predicted from the previous one instead of reading the
	// bounding boxes back in the middle of the scan half (a host round trip of ~25 us on a 120 us chain).
	bool spec = (0 == spec_mode ? (m->opt_spec && m->spec_valid) : false) && merged && 0 == early_stopping && n > 0 &&
	            n <= (1u << 29) && !(d_rgb && !m->g.color);
	// the fast path (fast_kernels.h): the whole scan on the predicted grid in five launches, the tree update tiled
	// (PointCloud2 records with colours: the first kernel that loads the points -- k_fhits here -- leaves the colours in the
	// set's own array, Ingest::rgb_out, which is what the tree update reads)
	const bool fast = spec && fastEligible(m, m->spec_grid, depth, simple, d_rgb, n, discrete);
	if (spec && !fast && (simple || !gridFitsLds(m->spec_grid))) spec = false;  // (grids beyond LDS and fixed-step casting: predicted for the fast path only)
	{
		ScanArgs& a = m->args;
		a.spec = spec;
		for (int k = 0; k < 3; ++k) a.origin[k] = origin[k];
		a.d_xyz = d_xyz;
		a.d_rgb = d_rgb;
		a.n = n;
		a.max_range = max_range;
		a.depth = depth;
		a.discrete = discrete;
		a.simple = simple;
		a.ing = m->ing;
	}
	if (spec) ++m->n_spec;
	u32 n_hits = 0, n_rays = 0;
	u64 capH = 0, capM = 0;
	int rc;
	auto volWalkPending = [&]() {
		for (int i = 0; i < kAlt; ++i)
			if (m->alt[i].pending && m->alt[i].vol_walk) return true;
		return false;
	};
	int vprc = UFOMAP_OK;  // result of joining a volume walk an earlier asynchronous call left enqueued
	if (fast) {
		// (a walk of the volume path that is still enqueued may have to be run again for the tiles that stood back: joined before
		// anything else is enqueued on the map stream)
		if (volWalkPending()) vprc = joinOlder(m);  // (its result is this call's too: a table growth inside that join may have failed)
		// a synchronous call with nothing in flight: the whole integration on the map stream (no hand-overs between streams)
		const bool solo = !async && m->opt_solo && oldestPendingAlt(m) < 0 && !m->sd_pending;
		rc = fastScanPhase(m, origin, d_xyz, n, max_range, discrete, false, async && !m->profiling && m->opt_early && m->opt_lazy_done, solo, swapped, d_rgb, simple);
		if (!rc && !m->gates && !solo) rc = (hipEventRecord(m->scan_ev, m->sstream) == hipSuccess) ? UFOMAP_OK : fail(UFOMAP_ERR_DEVICE, "hipEventRecord");
		lap(0, t_begin);
		if (rc) {
			(void)hipStreamSynchronize(m->sstream);
			return rc;
		}
		// ---- the tree update: the scan's slot on the map stream (enqueueSlot). Whichever walk gets there first takes the
		// scan: its own slot's, or the slot of an earlier scan that found this one's scan half already complete.
		m->pending = true;
		m->deferred = true;
		m->bound = Need{};
		m->last_rgb = nullptr;
		const auto t_map = std::chrono::steady_clock::now();
		// No slot of its own for a scan while two slots are still waiting on the map stream: the second of them takes every
		// scan along whose scan half is ready when it starts, and what is not ready then goes with the next slot that is
		// enqueued -- instead of a slot per scan, most of which would find their scan taken and cost four empty launches.
		// When the map stream keeps up (fewer than two slots waiting) every scan gets its slot at once.
		bool hold = false;
		if (async && !m->profiling && m->opt_early) {
			int nheld = m->sd_pending ? 0 : 1, nidle = 0, nslots = 0;  // (a scan whose descriptor is kept back cannot get a slot yet)
			for (int i = 0; i < kAlt; ++i) {
				const HandOver& o = m->alt[i];
				if (!o.pending) ++nidle;
				else if (o.deferred) ++nheld;
				else if (o.done_by_flag && o.has_slot && !setDoneNow(o)) ++nslots;
				else if (!o.done_by_flag) ++nslots;
			}
			const int bmax = std::max(1, std::min<int>(m->opt_batch_max, (int)UFO_BATCH_MAX));
			hold = nidle > 0 && nheld < std::min(bmax, 4) && nslots >= 2;
			if (m->opt_hold > 1) hold = nidle > 0 && nheld < std::min(m->opt_hold, bmax);  // (test aid: a slot for every hold-th scan)
		}
		if (!hold) {
			rc = flushDeferred(m, false);
			if (rc) return rc;
		}
		lap(1, t_map);
		if (async && caller_owned) {
			const int wrc = awaitCloudConsumed(m);
			if (wrc) return wrc;
		}
		int prc = UFOMAP_OK;
		if (!async) {
			prc = joinOlder(m);  // (occupancy_map_base.h:315: the previous integrations are joined first)
			{
				// the word k_ftail stores behind the result block in pinned memory (as the joins of asynchronous scans do): the
				// host sees it a few microseconds before a stream synchronisation would return
				volatile unsigned long long* done = reinterpret_cast<volatile unsigned long long*>(m->h_res + 1);
				const auto t0 = std::chrono::steady_clock::now();
				for (u32 spins = 0; *done != (unsigned long long)m->seq; ++spins) {
					if (0 == (spins & 1023u) && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) {
						HIP_TRY(hipStreamSynchronize(m->stream));
						break;
					}
				}
				std::atomic_thread_fence(std::memory_order_acquire);
			}
			rc = finishPending(m);
			if (3 == m->opt_fast) {
				// debugging aid: the scratch arrays must have been left clean
				std::vector<u32> h(m->b_first.cap / 4), tbw(m->b_tilebits.cap / 4);
				(void)hipMemcpy(h.data(), m->b_first.p, h.size() * 4, hipMemcpyDeviceToHost);
				(void)hipMemcpy(tbw.data(), m->b_tilebits.p, tbw.size() * 4, hipMemcpyDeviceToHost);
				size_t bad = 0, badt = 0;
				for (u32 v : h) bad += v != 0xFFFFFFFFu;
				for (u32 v : tbw) badt += v != 0u;
				fprintf(stderr, "[fast dbg] first[]: %zu of %zu entries not clean, tile bitmap: %zu words not clean, dirty flag %d\n", bad, h.size(), badt,
				        (int)m->first_dirty);
			}
			return rc ? rc : (prc ? prc : vprc);
		}
		const auto t_join = std::chrono::steady_clock::now();
		prc = joinCompleted(m);
		if (!prc) prc = vprc;
		if (m->prev_flagged) {
			// an integration that was joined had flagged itself (and has been repeated, or has failed): every walk enqueued
			// behind it stood back. Drain in order -- each is repeated by its own join -- before anything new is enqueued.
			const int drc = joinEnqueued(m);
			if (!prc) prc = drc;
			m->prev_flagged = false;
		}
		lap(2, t_join);
		return prc;
	}
	// ---- the general path ----
	m->fast = false;
	m->deferred = false;
	m->chain_ok = false;  // (no walk may take scans from either side of this update together)
	{
		// its tree update is enqueued by this call: what waits for company goes first; at most two older integrations stay in flight
		const int frc = flushDeferred(m);
		if (frc) return frc;
		while (countPendingAlts(m) > 2) (void)joinOldestAlt(m);
	}
	bool staged = false;
	if (async && caller_owned && n) {
		// The kernels of this path read the cloud (and, colour maps, the colours: in the map half) after the call has
		// returned: they read the set's own copy, made on the prep stream -- a device-to-device copy, awaited below.
		const size_t xyz_bytes = m->ing.data ? n * (size_t)m->ing.step : n * 24;
		HIP_TRY(m->b_in_xyz.reserve(xyz_bytes));
		HIP_TRY(hipMemcpyAsync(m->b_in_xyz.p, m->ing.data ? static_cast<const void*>(m->ing.data) : static_cast<const void*>(d_xyz), xyz_bytes,
		                       hipMemcpyDeviceToDevice, m->pstream));
		if (m->ing.data) {
			m->ing.data = m->b_in_xyz.as<uint8_t>();
			d_xyz = reinterpret_cast<const double*>(m->ing.data);
		} else {
			d_xyz = m->b_in_xyz.as<double>();
		}
		if (d_rgb && d_rgb != m->b_in_rgb.as<uint8_t>()) {
			HIP_TRY(m->b_in_rgb.reserve(n * 3));
			HIP_TRY(hipMemcpyAsync(m->b_in_rgb.p, d_rgb, n * 3, hipMemcpyDeviceToDevice, m->pstream));
			d_rgb = m->b_in_rgb.as<uint8_t>();
		}
		HIP_TRY(hipEventRecord(m->copy_ev, m->pstream));
		m->args.d_xyz = d_xyz;
		m->args.d_rgb = d_rgb;
		m->args.ing = m->ing;
		staged = true;
	}
	// (a host cloud is copied on the prep stream)
	HIP_TRY(hipEventRecord(m->prep_ev, m->pstream));
	HIP_TRY(hipStreamWaitEvent(m->sstream, m->prep_ev, 0));
	rc = scanPhase(m, origin, d_xyz, d_rgb, n, max_range, depth, discrete, simple, early_stopping, &n_hits, &n_rays, spec);
	if (staged) HIP_TRY(hipEventSynchronize(m->copy_ev));
	if (!rc && n && m->vol) {
		// the volume path: its scan half has been awaited (the list of tiles was read back); the tree update runs here, synchronously
		// (an asynchronous call returns a finished integration)
		lap(0, t_begin);
		const int prc = joinOlder(m);  // (occupancy_map_base.h:315: the previous integrations are joined first -- a volume walk among them)
		const auto t_map = std::chrono::steady_clock::now();
		rc = volMapPhase(m, 0 != async && 0 != m->opt_vol_async);
		lap(1, t_map);
		if (!rc && !m->vol_walk) rc = finishPending(m);
		return rc ? rc : prc;
	}
	if (!rc && n && volWalkPending()) {  // (see above; its result is this call's too, below)
		vprc = joinOlder(m);
		m->cs = m->sstream;
	}
	if (!rc && n) rc = extractPhase(m, n_hits, n_rays, &capH, &capM, merged);
	lap(0, t_begin);
	auto mapHalf = [&](const ScanCtl* prev, Need extra_used, u32 headroom) { return mapPhase(m, depth, d_rgb, capH, capM, merged, prev, extra_used, headroom); };
	if (!rc && n) rc = (hipEventRecord(m->scan_ev, m->sstream) == hipSuccess) ? UFOMAP_OK : fail(UFOMAP_ERR_DEVICE, "hipEventRecord");
	// the update enqueued just before this one, if it has not been joined
	int pk = -1;
	Need in_flight;
	for (int i = 0; i < kAlt; ++i) {
		if (!m->alt[i].pending) continue;
		in_flight += m->alt[i].bound;
		if (pk < 0 || m->alt[i].seq > m->alt[pk].seq) pk = i;
	}
	// ---- early map half: enqueue THIS scan's tree update behind the previous one BEFORE that one has been joined,
	// so that the updates run back to back on the map stream and the next call can start its scan half while two
	// updates are still in flight. The first kernel looks at the predecessor's error flags: if it flagged itself, this
	// update stands back too (ERR_PREV, which cascades) and everything flagged is re-run in order when it is joined.
	bool early = false;
	// (only behind an update of the same kind: one that reports through its control block. Behind a walk of the steady-state path -- which
	// reports through the pipe's status words and hands its control block back in the start state -- the predecessor is joined first;
	// round 6: scripts/dev/fuzz_api.py found a map that differed after an early-stopping scan enqueued behind a steady-state scan.)
	if (!rc && n && async && merged && m->opt_early && pk >= 0 && !m->alt[pk].done_by_flag && !m->profiling) {
		m->cs = m->stream;
		HIP_TRY(hipStreamWaitEvent(m->stream, m->scan_ev, 0));
		m->last_rgb = d_rgb;
		const auto t_map = std::chrono::steady_clock::now();
		const int erc = mapHalf(m->alt[pk].b_ctl.as<ScanCtl>(), in_flight, 0u);
		if (erc < 0) return erc;
		early = 0 == erc;  // 1: the table might have to grow: join first (below)
		if (early) {
			m->done_by_flag = false;
			HIP_TRY(hipEventRecord(m->done_ev, m->stream));
		}
		lap(1, t_map);
	}
	int prc = UFOMAP_OK;
	if (early) {
		// join all but the previous integration (long finished as a rule): the previous one and this one keep running
		m->prev_flagged = false;
		const auto t_join = std::chrono::steady_clock::now();
		while (countPendingAlts(m) > 1 && !m->prev_flagged) {
			const int r = joinOldestAlt(m);
			if (!prc) prc = r;
		}
		lap(2, t_join);
		if (m->prev_flagged) {
			// it had flagged itself (and has been repeated, or has failed): the previous update and this one stood back.
			// Drain in order: the previous ones are repeated by their own joins, then this one's tree update is enqueued again.
			HIP_TRY(hipStreamSynchronize(m->stream));
			for (int k; (k = oldestPendingAlt(m)) >= 0;) {
				const int r = finishSet(m, k);
				if (!prc) prc = r;
				if (r && UFOMAP_OK == m->async_status) m->async_status = r;
			}
			HIP_TRY(hipStreamSynchronize(m->stream));
			hipLaunchKernelGGL(k_ctl_clear, dim3(1), dim3(1), 0, m->stream, m->b_ctl.as<ScanCtl>(), (u32)ERR_PREV);
			m->cs = m->stream;
			rc = mapHalf(nullptr, Need{}, 0u);
			if (rc) return rc;
			m->done_by_flag = false;
			HIP_TRY(hipEventRecord(m->done_ev, m->stream));
			m->prev_flagged = false;
		}
		m->pending = true;
		m->deferred = false;
		m->bound = m->scan_new_bound;
		return prc ? prc : vprc;
	}
	// join the previous integrations (occupancy_map_base.h:315): their status is reported by wait()/this call
	prc = joinOlder(m);
	if (!prc) prc = vprc;
	if (rc || 0 == n) {
		if (rc) (void)hipStreamSynchronize(m->sstream);
		return rc ? rc : prc;
	}
	// ---- map half on the map stream, after the scan half of THIS scan ----
	m->cs = m->stream;
	HIP_TRY(hipStreamWaitEvent(m->stream, m->scan_ev, 0));
	m->last_rgb = d_rgb;
	rc = mapHalf(nullptr, Need{}, (async && merged && m->opt_early) ? 2u : 0u);
	if (rc) return rc;
	HIP_TRY(hipGetLastError());
	m->pending = true;
	m->deferred = false;
	m->bound = m->scan_new_bound;
	if (!async) {
		HIP_TRY(hipStreamSynchronize(m->stream));
		rc = finishPending(m);
		return rc ? rc : prc;
	}
	m->done_by_flag = false;
	HIP_TRY(hipEventRecord(m->done_ev, m->stream));
	return prc;
}
// Repeat, synchronously and with the boxes read back, the integration whose hand-over set is current: it had been
// enqueued on a predicted grid and flagged itself (ERR_SPEC, or a bound derived from the prediction was exceeded),
// so nothing of it has reached the map. Called from finishPending, i.e. before any later update of the map.
int redoScan(ufomap_map* m)
{
	const ScanArgs a = m->args;
	m->args.spec = false;
	m->spec_valid = false;
	m->chain_ok = false;
	++m->n_spec_redo;
	// an update enqueued behind this one looks at this control block when it starts: let it do so (and stand back)
	// before the block is rewritten
	HIP_TRY(hipStreamSynchronize(m->stream));
	// the scan half of the NEXT integration may already have run: what it left in the map object is restored
	const Grid sgM = m->gridM, sgH = m->gridH;
	const bool shH = m->haveH, shM = m->haveM;
	const u32 shb = m->hb_cap_mask, sld = m->last_depth;
	const uint8_t* slr = m->last_rgb;
	const Ingest sing = m->ing;
	m->ing = a.ing;
	m->cs = m->sstream;
	u32 n_hits = 0, n_rays = 0;
	int rc = scanPhase(m, a.origin, a.d_xyz, a.d_rgb, a.n, a.max_range, a.depth, a.discrete, a.simple, 0, &n_hits, &n_rays, false);
	u64 capH = 0, capM = 0;
	const bool merged = 0 == a.depth && 0 != m->opt_merge;
	const bool vol = !rc && m->vol;
	if (!rc && !vol) rc = extractPhase(m, n_hits, n_rays, &capH, &capM, merged);
	if (!rc) rc = (hipStreamSynchronize(m->sstream) == hipSuccess) ? UFOMAP_OK : fail(UFOMAP_ERR_DEVICE, "hipStreamSynchronize");
	if (!rc) {
		m->cs = m->stream;
		m->last_rgb = a.d_rgb;
		rc = vol ? volMapPhase(m) : mapPhase(m, a.depth, a.d_rgb, capH, capM, merged);
	}
	if (!rc) rc = (hipStreamSynchronize(m->stream) == hipSuccess) ? UFOMAP_OK : fail(UFOMAP_ERR_DEVICE, "hipStreamSynchronize");
	if (!rc) {
		m->pending = true;
		rc = finishPending(m);
	}
	if (m->seq != m->latest_seq) {
		// a later integration's scan half has already run: the map object describes ITS grids
		m->gridM = sgM;
		m->gridH = sgH;
		m->haveH = shH;
		m->haveM = shM;
		m->hb_cap_mask = shb;
		m->last_depth = sld;
		m->last_rgb = slr;
	}
	m->ing = sing;
	return rc;
}
}  // namespace

extern "C" {

const char* ufomap_last_error(void) { return g_err.c_str(); }

int ufomap_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

const char* ufomap_version(void) { return "ufomap_amd 0.1 (gfx950)"; }

ufomap_map* ufomap_map_create(double resolution, unsigned depth_levels, int automatic_pruning, double occupied_thres,
                              double free_thres, double prob_hit, double prob_miss, double clamping_thres_min,
                              double clamping_thres_max, int has_color, int device)
{
	if (depth_levels < 2 || depth_levels > 21) {  // octree.h:931-935
		fail(UFOMAP_ERR_INVALID, "depth_levels has to be [2, 21]");
		return nullptr;
	}
	if (!(resolution > 0)) {
		fail(UFOMAP_ERR_INVALID, "resolution must be positive");
		return nullptr;
	}
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
		fail(UFOMAP_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
		return nullptr;
	}
	if (device < 0 || device >= ndev) {
		fail(UFOMAP_ERR_INVALID, "device ordinal out of range");
		return nullptr;
	}
	if (hipSetDevice(device) != hipSuccess) {
		fail(UFOMAP_ERR_DEVICE, "hipSetDevice failed");
		return nullptr;
	}
	ufomap_map* m = new ufomap_map;
	if (const char* e = getenv("UFOMAP_MERGE_PHASES")) m->opt_merge = atoi(e) != 0;  // test override, see ufomap_map_set_option
	m->device = device;
	MapGeom& g = m->g;
	g.res = resolution;
	g.rf = 1.0 / resolution;
	g.L = depth_levels;
	g.M = (u32)std::pow(2, depth_levels - 1);
	g.hs[0] = resolution / 2.0;
	g.hs[1] = resolution;
	for (unsigned i = 2; i < 23; ++i) g.hs[i] = g.hs[i - 1] * 2.0;
	g.color = has_color ? 1 : 0;
	g.pruning = automatic_pruning ? 1 : 0;
	setSensorModel(m, occupied_thres, free_thres, prob_hit, prob_miss, clamping_thres_min, clamping_thres_max);
	// The three streams of the pipeline get three different PRIORITIES -- not to order their work (it is ordered by gates),
	// but because the runtime multiplexes streams onto a few hardware queues (4 by default) and streams of different
	// priority never share one: two pipeline streams on one hardware queue run strictly one after the other, a gate
	// spinning in front of the work it waits for. (Measured with several handles created one after the other in one
	// process: 0.066 ms per scan when the streams happened to get queues of their own, 0.106 when two shared one.)
	{
		int cus = 0;
		if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) m->n_cus = cus;
	}
	int prio_lo = 0, prio_hi = 0;
	(void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);  // (least, greatest: numerically lower = higher priority)
	const int prio_mid = (prio_lo + prio_hi) / 2;
	bool ok = hipStreamCreateWithPriority(&m->stream, hipStreamNonBlocking, prio_hi) == hipSuccess &&
	          hipStreamCreateWithPriority(&m->sstream, hipStreamNonBlocking, prio_mid) == hipSuccess &&
	          hipStreamCreateWithFlags(&m->xstream, hipStreamNonBlocking) == hipSuccess &&
	          hipStreamCreateWithPriority(&m->pstream, hipStreamNonBlocking, prio_lo) == hipSuccess &&
	          hipEventCreateWithFlags(&m->prep_ev, hipEventDisableTiming) == hipSuccess &&
	          hipEventCreateWithFlags(&m->done_ev, hipEventDisableTiming) == hipSuccess &&
	          hipEventCreateWithFlags(&m->scan_ev, hipEventDisableTiming) == hipSuccess &&
	          hipEventCreateWithFlags(&m->copy_ev, hipEventDisableTiming) == hipSuccess &&
	          hipHostMalloc((void**)&m->h_ctl, sizeof(ScanCtl) + 64) == hipSuccess &&
	          hipHostMalloc((void**)&m->h_res, sizeof(ScanCtl) + 64) == hipSuccess &&
	          m->b_ctl_init.reserve(sizeof(ScanCtl) + 64) == hipSuccess &&
	          m->b_pipe.reserve(sizeof(Pipe)) == hipSuccess && hipMemset(m->b_pipe.p, 0, sizeof(Pipe)) == hipSuccess &&
	          hipMalloc((void**)&m->sig_prep, 8) == hipSuccess && hipMemset(m->sig_prep, 0, 8) == hipSuccess &&
	          hipHostMalloc((void**)&m->h_root, sizeof(MapRoot)) == hipSuccess &&
	          hipHostMalloc((void**)&m->h_prep, 64) == hipSuccess &&
	          m->b_ctl.reserve(sizeof(ScanCtl) + 64) == hipSuccess && m->b_root.reserve(sizeof(MapRoot)) == hipSuccess;
	for (int i = 0; ok && i < kAlt; ++i) {
		HandOver& a = m->alt[i];
		ok = hipEventCreateWithFlags(&a.done_ev, hipEventDisableTiming) == hipSuccess &&
		     hipHostMalloc((void**)&a.h_ctl, sizeof(ScanCtl) + 64) == hipSuccess && hipHostMalloc((void**)&a.h_res, sizeof(ScanCtl) + 64) == hipSuccess &&
		     hipMalloc((void**)&a.sig_prep, 8) == hipSuccess && hipMemset(a.sig_prep, 0, 8) == hipSuccess &&
		     a.b_ctl.reserve(sizeof(ScanCtl) + 64) == hipSuccess;
		if (ok) memset(a.h_ctl, 0, sizeof(ScanCtl));
	}
	if (!ok) {
		fail(UFOMAP_ERR_DEVICE, "HIP resource creation failed");
		ufomap_map_destroy(m);
		return nullptr;
	}
	m->cs = m->stream;
	memset(m->h_ctl, 0, sizeof(ScanCtl));
	*m->h_prep = 0ull;
	if (hipMemset(m->b_root.p, 0, sizeof(MapRoot)) != hipSuccess || allocTable(m, 768u, 8192u, &m->t, &m->tb) || resetRoot(m)) {
		ufomap_map_destroy(m);
		return nullptr;
	}
	{
		// the DDA kernels use up to 144 KiB of dynamic LDS (whole-grid mode) / 128 KiB (filter mode)
		const int maxlds = (int)UFO_DDA_LDSGRID_MAX;
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dda<false, DDA_LDSGRID>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds);
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dda<true, DDA_LDSGRID>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds);
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dda<false, DDA_FILTER>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds);
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dda<true, DDA_FILTER>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds);
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dda_seg<DDA_LDSGRID>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds);
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dda_seg<DDA_FILTER>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds);
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_walk), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds);
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cast<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (160 << 10) - 512);
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cast<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (160 << 10) - 512);
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cast<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (160 << 10) - 512);
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fcast_simple<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (160 << 10) - 512);
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fcast_simple<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (160 << 10) - 512);

	}
	(void)hipDeviceSynchronize();  // (the fills above ran on the null stream, which the handle's non-blocking streams do not wait for)
	if (m->opt_vol) (void)volSelfTest(m);  // (the per-XCD atomics the volume path rests on, once per device -- here, not inside a scan)
	for (int a = 0; a < 3; ++a) {
		m->min_change[a] = g.hs[g.L];  // resetMinMaxChangeDetection (occupancy_map_base.h:806-810)
		m->max_change[a] = -g.hs[g.L];
	}
	return m;
}

void ufomap_map_destroy(ufomap_map* m)
{
	if (!m) return;
	(void)hipSetDevice(m->device);
	if (m->pstream) (void)hipStreamSynchronize(m->pstream);
	if (m->sstream) (void)hipStreamSynchronize(m->sstream);
	if (m->stream) (void)hipStreamSynchronize(m->stream);
	m->tb.release();
	m->b_changes.release();
	for (HandOver& a : m->alt) {
		DevBuf* abufs[] = {&a.b_ctl, &a.b_entries, &a.b_hh_keys, &a.b_in_xyz, &a.b_in_rgb, &a.b_gridM, &a.b_gridH, &a.b_part1, &a.b_hit_code, &a.b_first, &a.b_tilebits, &a.b_slabs, &a.b_keep, &a.b_keep_rgb};
		for (DevBuf* b : abufs) b->release();
		if (a.h_ctl) (void)hipHostFree(a.h_ctl);
		if (a.h_res) (void)hipHostFree(a.h_res);
		if (a.sig_prep) (void)hipFree(a.sig_prep);
		if (a.h_stage) (void)hipHostFree(a.h_stage);
		if (a.h_res_all) (void)hipHostFree(a.h_res_all);
		if (a.xchg_ev) (void)hipEventDestroy(a.xchg_ev);
		for (DevBuf* b : {&a.b_xsend, &a.b_xrecv, &a.b_bpipe}) b->release();
		if (a.done_ev) (void)hipEventDestroy(a.done_ev);
	}
	if (m->scan_ev) (void)hipEventDestroy(m->scan_ev);
	if (m->sstream) (void)hipStreamDestroy(m->sstream);
	if (m->pstream) (void)hipStreamDestroy(m->pstream);
	if (m->prep_ev) (void)hipEventDestroy(m->prep_ev);
	DevBuf* bufs[] = {&m->b_root,
	                  &m->b_ctl,     &m->b_pt_end,  &m->b_pt_flag,  &m->b_pt_slot, &m->b_ray_end, &m->b_hit_code, &m->b_hit_pt,
	                  &m->b_hh_keys, &m->b_gridM,   &m->b_crec,    &m->b_dlist,   &m->b_rays,   &m->b_part0,   &m->b_part1,   &m->b_slabs,   &m->b_hb_keys, &m->b_hb_mask, &m->b_hb_time,   &m->b_entries, &m->b_ent_slot, &m->b_newlist,
	                  &m->b_wl0,     &m->b_wl1,     &m->b_in_xyz,   &m->b_in_rgb,  &m->b_codes,   &m->b_dump,
	                  &m->b_first,   &m->b_tilebits, &m->b_tilerec, &m->b_gridH, &m->b_pipe, &m->b_keep, &m->b_keep_rgb, &m->b_blk_range, &m->b_ctl_init};
	for (DevBuf* b : bufs) b->release();
	for (PendingEvent& pe : m->pend_ev) {
		(void)hipEventDestroy(pe.a);
		(void)hipEventDestroy(pe.b);
	}
	for (hipEvent_t e : m->ev_pool) (void)hipEventDestroy(e);
	if (m->h_ctl) (void)hipHostFree(m->h_ctl);
	if (m->h_res) (void)hipHostFree(m->h_res);
	if (m->sig_prep) (void)hipFree(m->sig_prep);
	if (m->stager) {
		m->stager->stop();
		delete m->stager;
		m->stager = nullptr;
	}
	if (m->h_stage_flag) (void)hipHostFree(m->h_stage_flag);
	if (m->h_stage) (void)hipHostFree(m->h_stage);
	if (m->h_res_all) (void)hipHostFree(m->h_res_all);
	if (m->xchg_ev) (void)hipEventDestroy(m->xchg_ev);
	m->b_sig_xchg.release();
	for (DevBuf* b : {&m->b_xsend, &m->b_xrecv, &m->b_bpipe}) b->release();
	if (m->copy_ev) (void)hipEventDestroy(m->copy_ev);
	if (m->h_root) (void)hipHostFree(m->h_root);
	if (m->h_prep) (void)hipHostFree(m->h_prep);
	if (m->h_ser) (void)hipHostFree(m->h_ser);
	if (m->h_out) (void)hipHostFree(m->h_out);
	for (DevBuf& b : m->b_ser) b.release();
	if (m->done_ev) (void)hipEventDestroy(m->done_ev);
	if (m->xstream) (void)hipStreamDestroy(m->xstream);
	if (m->gstream) (void)hipStreamSynchronize(m->gstream), (void)hipStreamDestroy(m->gstream);
	if (m->pack_ev) (void)hipEventDestroy(m->pack_ev);
	if (m->stream) (void)hipStreamDestroy(m->stream);
	delete m;
}

int ufomap_map_clear(ufomap_map* m)
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	HIP_TRY(hipSetDevice(m->device));
	(void)ufomap_map_wait(m);
	m->poisoned = false;
	m->cs = m->stream;
	u32 cap = m->t.mask + 1;
	HIP_TRY(hipMemsetAsync(m->t.keyA, 0, (size_t)cap * 16, m->stream));  // (key, flags, stamp: contiguous, allocTable)
	HIP_TRY(hipMemsetAsync(m->t.tmax, 0, (size_t)cap * 8, m->stream));
	HIP_TRY(hipMemsetAsync(m->t.lu_fl, 0, (size_t)cap * 4, m->stream));
	HIP_TRY(hipMemsetAsync(m->t.gdir, 0, (size_t)m->t.nG * 8, m->stream));
	HIP_TRY(hipMemsetAsync(m->t.gcnt, 0, UFO_GCNT_WORDS * 4, m->stream));
	return resetRoot(m);
}

int ufomap_map_reserve(ufomap_map* m, size_t n_blocks)
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	HIP_TRY(hipSetDevice(m->device));
	(void)ufomap_map_wait(m);
	m->cs = m->stream;
	// n_blocks node blocks of a map as scans build it: ~40 live blocks per tile group, a seventh of the groups again above
	Need need;
	need.blocks = n_blocks;
	need.groups = (u64)n_blocks / 40 + 64;
	need.upper = (u64)n_blocks / 200 + 1024;
	need.groups = need.groups > m->used_g ? need.groups - m->used_g : 0;
	need.upper = need.upper > m->used_u ? need.upper - m->used_u : 0;
	if (tableTakes(m, need)) return UFOMAP_OK;
	return growFor(m, need);
}

int ufomap_map_set_scratch_limit(ufomap_map* m, size_t bytes)
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	m->scratch_limit = bytes;
	return UFOMAP_OK;
}

int ufomap_map_set_sensor_model(ufomap_map* m, double occupied_thres, double free_thres, double prob_hit, double prob_miss,
                                double clamping_thres_min, double clamping_thres_max)
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	int rc = ufomap_map_wait(m);
	if (rc) return rc;
	auto logit = [](double p) { return std::log(p / (1.0 - p)); };
	if (logit(occupied_thres) != m->model_log[0] || logit(free_thres) != m->model_log[1]) {
		// new thresholds change every stored contains_free / contains_unknown: re-evaluate the tree as the reference does
		rc = ufomap_map_set_occupied_free_thres(m, occupied_thres, free_thres);
		if (rc) return rc;
	}
	setSensorModel(m, occupied_thres, free_thres, prob_hit, prob_miss, clamping_thres_min, clamping_thres_max);
	return UFOMAP_OK;
}

int ufomap_map_insert_device(ufomap_map* m, const double sensor_origin[3], const double* d_xyz, const uint8_t* d_rgb, size_t n,
                             double max_range, unsigned depth, int discrete, int simple_ray_casting, unsigned early_stopping,
                             int async)
{
	return doInsert(m, sensor_origin, d_xyz, d_rgb, n, max_range, depth, discrete, simple_ray_casting, early_stopping, async, false);
}

// Host cloud -> HBM for the hand-over set of THIS scan (already rotated in) without stalling the caller on the GPU.
// The caller's buffers are never referenced after the insert call returns (SURVEY.md 8b ownership):
//  * pageable memory (the normal case, e.g. the std::vector of a PointCloud): the host copies it into the set's PINNED
//    staging buffer and an asynchronous H2D copy drains that on the scan stream -- no synchronisation at all, so with
//    async=true the copy of scan i+1 overlaps the kernels of scan i; the staging buffer is free again when the set's
//    integration has been joined (rotateSets), which is before the set is used for another scan;
//  * caller-owned pinned memory (hipHostMalloc / hipHostRegister): DMA straight from it; only the copy is awaited.
// the DMA out of a caller-owned pinned cloud has finished (the caller may reuse the buffer when the call returns)
static int copyDone(ufomap_map* m, int rc)
{
	if (m->stage_wait) {  // (the helper thread has read all of the caller's pageable cloud)
		m->stage_wait = false;
		m->stager->wait();
	}
	if (m->copy_wait) {
		m->copy_wait = false;
		const hipError_t e = hipEventSynchronize(m->copy_ev);
		if (e != hipSuccess && !rc) return fail(UFOMAP_ERR_DEVICE, hipGetErrorString(e));
	}
	return rc;
}
static int uploadCloud(ufomap_map* m, const void* a, size_t a_bytes, const void* b, size_t b_bytes, const void** d_a, const void** d_b)
{
	*d_a = *d_b = nullptr;
	if (0 == a_bytes) return UFOMAP_OK;
	HIP_TRY(m->b_in_xyz.reserve(a_bytes));
	if (b_bytes) HIP_TRY(m->b_in_rgb.reserve(b_bytes));
	auto isPinned = [](const void* p) {
		hipPointerAttribute_t at;
		const bool ok = hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeHost;
		(void)hipGetLastError();  // an unregistered pointer is an "error" on some runtimes: not ours
		return ok;
	};
	if (isPinned(a) && (!b_bytes || isPinned(b))) {
		HIP_TRY(hipMemcpyAsync(m->b_in_xyz.p, a, a_bytes, hipMemcpyHostToDevice, m->pstream));
		if (b_bytes) HIP_TRY(hipMemcpyAsync(m->b_in_rgb.p, b, b_bytes, hipMemcpyHostToDevice, m->pstream));
		// (the caller's pinned buffer is read by the DMA engine until this event: awaited when the CALL ends, copyDone -- not here:
		// round 6. Until round 5 the host sat through the 66 us of the copy and only then enqueued the scan's 30 us of launches.)
		HIP_TRY(hipEventRecord(m->copy_ev, m->pstream));
		m->copy_wait = true;
	} else {
		const size_t off_b = (a_bytes + 255) & ~(size_t)255, need = off_b + b_bytes;
		if (need > m->h_stage_cap) {
			if (m->h_stage) HIP_TRY(hipHostFree(m->h_stage));
			m->h_stage = nullptr;
			m->h_stage_cap = 0;
			const size_t want = (need + need / 2 + 4095) & ~(size_t)4095;
			HIP_TRY(hipHostMalloc(&m->h_stage, want));
			m->h_stage_cap = want;
		}
		if (m->opt_stage_thread && a_bytes >= (256u << 10) && useGates(m)) {
			// Round 6: the copy into the staging buffer (60 us for a 3 MB cloud) by a helper thread WHILE this thread enqueues the scan
			// (30 us of launches): the asynchronous H2D copy waits behind a one-wave gate kernel for the helper's word. The call joins the
			// helper before it returns (copyDone): the caller's cloud is never referenced afterwards, as before.
			if (!m->stager) {
				if (hipHostMalloc((void**)&m->h_stage_flag, 64) != hipSuccess) return fail(UFOMAP_ERR_DEVICE, "pinned memory for the staging flag");
				*m->h_stage_flag = 0ull;
				m->stager = new StageWorker();
				m->stager->flag = m->h_stage_flag;
				m->stager->th = std::thread([w = m->stager] { w->run(); });
			}
			StageWorker::Job j{};
			j.dst[0] = m->h_stage;
			j.src[0] = a;
			j.bytes[0] = a_bytes;
			j.dst[1] = static_cast<char*>(m->h_stage) + off_b;
			j.src[1] = b;
			j.bytes[1] = b_bytes;
			// With nothing in flight -- a caller that waits for every scan, like the reference's server -- the cloud goes on in PIECES: the
			// workgroups of k_stage_copy read piece k across PCIe while the helper copies piece k + 1, and the cloud is in HBM one piece's
			// transfer after the helper's last byte instead of a whole DMA transfer (server loop 0.420 -> 0.380 ms). In a row of
			// asynchronous calls the transfer overlaps the scan before anyway, and the DMA engine does it without a CU (91 vs 95-105 us/scan).
			const bool in_pieces = m->opt_stage_pieces > 0 && 0 == countPendingAlts(m);
			const u32 np = in_pieces ? (u32)std::min(16, m->opt_stage_pieces) : 1u;
			j.chunk = std::max<size_t>(((a_bytes + np - 1) / np + 4095) & ~(size_t)4095, 64u << 10);
			m->stager->post(j);
			m->stage_wait = true;
			const unsigned long long job = m->stager->posted.load(std::memory_order_relaxed);
			const u32 pa = (u32)((a_bytes + j.chunk - 1) / j.chunk), wgs = 4u;
			if (!in_pieces) {
				hipLaunchKernelGGL(k_host_gate, dim3(1), dim3(64), 0, m->pstream, m->h_stage_flag, (job << 8) | (unsigned long long)(pa + (b_bytes ? 1u : 0u)), 200000000ull,
				                   (u32*)nullptr);
				HIP_TRY(hipMemcpyAsync(m->b_in_xyz.p, m->h_stage, a_bytes, hipMemcpyHostToDevice, m->pstream));
				if (b_bytes) HIP_TRY(hipMemcpyAsync(m->b_in_rgb.p, static_cast<char*>(m->h_stage) + off_b, b_bytes, hipMemcpyHostToDevice, m->pstream));
			} else {
				hipLaunchKernelGGL(k_stage_copy, dim3(pa * wgs), dim3(256), 0, m->pstream, m->h_stage_flag, job, 0u, static_cast<const uint8_t*>(m->h_stage), m->b_in_xyz.as<uint8_t>(),
				                   (unsigned long long)a_bytes, (unsigned long long)j.chunk, wgs, 200000000ull);
				if (b_bytes)
					hipLaunchKernelGGL(k_stage_copy, dim3(wgs), dim3(256), 0, m->pstream, m->h_stage_flag, job, pa, static_cast<const uint8_t*>(m->h_stage) + off_b,
					                   m->b_in_rgb.as<uint8_t>(), (unsigned long long)b_bytes, (unsigned long long)((b_bytes + 15) & ~(size_t)15), wgs, 200000000ull);
			}
		} else {
			memcpy(m->h_stage, a, a_bytes);
			if (b_bytes) memcpy(static_cast<char*>(m->h_stage) + off_b, b, b_bytes);
			HIP_TRY(hipMemcpyAsync(m->b_in_xyz.p, m->h_stage, a_bytes, hipMemcpyHostToDevice, m->pstream));
			if (b_bytes) HIP_TRY(hipMemcpyAsync(m->b_in_rgb.p, static_cast<char*>(m->h_stage) + off_b, b_bytes, hipMemcpyHostToDevice, m->pstream));
		}
	}
	*d_a = m->b_in_xyz.p;
	if (b_bytes) *d_b = m->b_in_rgb.p;
	return UFOMAP_OK;
}

int ufomap_map_insert(ufomap_map* m, const double sensor_origin[3], const double* xyz, const uint8_t* rgb, size_t n,
                      double max_range, unsigned depth, int discrete, int simple_ray_casting, unsigned early_stopping, int async)
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	HIP_TRY(hipSetDevice(m->device));
	if (m->poisoned) return fail(UFOMAP_ERR_CAPACITY, "the map is inconsistent after a node table overflow: ufomap_map_clear it");
	(void)rotateSets(m);  // the staging buffers belong to the hand-over set of THIS scan
	const void *d_xyz = nullptr, *d_rgb = nullptr;
	const int urc = uploadCloud(m, xyz, n * 24, rgb, rgb ? n * 3 : 0, &d_xyz, &d_rgb);
	if (urc) return copyDone(m, urc);
	return copyDone(m, doInsert(m, sensor_origin, static_cast<const double*>(d_xyz), static_cast<const uint8_t*>(d_rgb), n, max_range, depth, discrete,
	                            simple_ray_casting, early_stopping, async, true));
}

int ufomap_map_insert_pointcloud2(ufomap_map* m, const double translation[3], const double rotation_wxyz[4], const void* data,
                                  int data_on_device, size_t n_points, uint32_t point_step, int off_x, int off_y, int off_z, int off_r,
                                  int off_g, int off_b, double max_range, unsigned depth, int discrete, int simple_ray_casting,
                                  unsigned early_stopping, int async)
{
	if (!m || !translation || !rotation_wxyz || (n_points && !data)) return fail(UFOMAP_ERR_INVALID, "null argument");
	const bool has_rgb = off_r >= 0 && off_g >= 0 && off_b >= 0;
	if (off_x < 0 || off_y < 0 || off_z < 0 || (u64)std::max(std::max(off_x, off_y), off_z) + 4 > point_step ||
	    (has_rgb && (u32)std::max(std::max(off_r, off_g), off_b) >= point_step))
		return fail(UFOMAP_ERR_INVALID, "field offsets outside the point record");
	if (m->g.color && !discrete)
		return fail(UFOMAP_ERR_UNSUPPORTED, "OccupancyMapColor::insertPointCloud<PointCloudColor> does not compile in the reference (SURVEY.md 4)");
	HIP_TRY(hipSetDevice(m->device));
	if (m->poisoned) return fail(UFOMAP_ERR_CAPACITY, "the map is inconsistent after a node table overflow: ufomap_map_clear it");
	(void)rotateSets(m);  // the staging buffers belong to the hand-over set of THIS scan
	const uint8_t* d_data = static_cast<const uint8_t*>(data);
	hipError_t e = hipSuccess;
	if (n_points && !data_on_device) {
		// the caller's message is never referenced after this call returns; what crosses PCIe is the raw record
		// stream (point_step bytes per point), not 24 bytes of float64 per point
		const void *d_a = nullptr, *d_b = nullptr;
		const int urc = uploadCloud(m, data, n_points * (size_t)point_step, nullptr, 0, &d_a, &d_b);
		if (urc) return copyDone(m, urc);
		d_data = static_cast<const uint8_t*>(d_a);
	}
	uint8_t* d_rgb = nullptr;
	if (e == hipSuccess && n_points && m->g.color) {
		e = m->b_in_rgb.reserve(n_points * 3);
		d_rgb = m->b_in_rgb.as<uint8_t>();
	}
	if (e != hipSuccess) return copyDone(m, fail(UFOMAP_ERR_DEVICE, hipGetErrorString(e)));
	Ingest ing{};
	ing.data = d_data;
	ing.step = point_step;
	ing.ox = (u32)off_x;
	ing.oy = (u32)off_y;
	ing.oz = (u32)off_z;
	ing.orr = has_rgb ? off_r : -1;
	ing.og = has_rgb ? off_g : -1;
	ing.ob = has_rgb ? off_b : -1;
	for (int k = 0; k < 4; ++k) ing.q[k] = rotation_wxyz[k];
	for (int k = 0; k < 3; ++k) ing.t[k] = translation[k];
	ing.rgb_out = d_rgb;
	m->ing = ing;
	// insertPointCloudDiscrete(transform.translation(), cloud, ...) (ufomap_mapping/src/server.cpp:118-120)
	const int rc = doInsert(m, translation, reinterpret_cast<const double*>(d_data), d_rgb, n_points, max_range, depth, discrete,
	                        simple_ray_casting, early_stopping, async, true, 0, data_on_device ? 1 : 0);
	m->ing = Ingest{};
	return copyDone(m, rc);
}

int ufomap_map_query(ufomap_map* m, const double* xyz, int xyz_on_device, size_t n, unsigned depth, float* logodds, uint8_t* state)
{
	if (!m || (n && (!xyz || !logodds || !state))) return fail(UFOMAP_ERR_INVALID, "null argument");
	if (depth >= m->g.L) return fail(UFOMAP_ERR_INVALID, "depth must be < depth_levels");
	if (n > 0x7FFFFFFFull) return fail(UFOMAP_ERR_INVALID, "more than 2^31 queries");
	HIP_TRY(hipSetDevice(m->device));
	int rc = ufomap_map_wait(m);
	if (rc || 0 == n) return rc;
	DevBuf b_in, b_lo, b_st;
	const double* d_xyz = xyz;
	hipError_t e = hipSuccess;
	if (!xyz_on_device) {
		e = b_in.reserve(n * 24);
		if (e == hipSuccess) e = hipMemcpyAsync(b_in.p, xyz, n * 24, hipMemcpyHostToDevice, m->stream);
		d_xyz = b_in.as<double>();
	}
	if (e == hipSuccess) e = b_lo.reserve(n * 4);
	if (e == hipSuccess) e = b_st.reserve(n);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_query, gridFor(n), dim3(256), 0, m->stream, m->t, m->g, d_xyz, (u32)n, (u32)depth, b_lo.as<float>(),
		                   b_st.as<uint8_t>());
		e = hipMemcpyAsync(logodds, b_lo.p, n * 4, hipMemcpyDeviceToHost, m->stream);
	}
	if (e == hipSuccess) e = hipMemcpyAsync(state, b_st.p, n, hipMemcpyDeviceToHost, m->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
	b_in.release();
	b_lo.release();
	b_st.release();
	if (e != hipSuccess) return fail(UFOMAP_ERR_DEVICE, hipGetErrorString(e));
	return UFOMAP_OK;
}

int ufomap_map_clamping_thres(ufomap_map* m, double* thres_min, double* thres_max)
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	// getClampingThresMin/Max (occupancy_map_base.h:742-744): toProb(LogitType) with LogitType = float, std::exp(float)
	if (thres_min) *thres_min = 1.0 / (1.0 + std::exp(-m->g.cmin));
	if (thres_max) *thres_max = 1.0 / (1.0 + std::exp(-m->g.cmax));
	return UFOMAP_OK;
}

int ufomap_map_set_value_volume(ufomap_map* m, const double aabb_min[3], const double aabb_max[3], double occupancy_value,
                                unsigned min_depth)
{
	if (!m || !aabb_min || !aabb_max) return fail(UFOMAP_ERR_INVALID, "null argument");
	double vc[3], vh[3];
	for (int k = 0; k < 3; ++k) {
		vh[k] = (aabb_max[k] - aabb_min[k]) / 2.0;  // AABB(min, max), geometry/aabb.h:62-65
		vc[k] = aabb_min[k] + vh[k];
	}
	return ufomap_map_set_value_volume_ch(m, vc, vh, occupancy_value, min_depth);
}

int ufomap_map_set_value_volume_ch(ufomap_map* m, const double aabb_center[3], const double aabb_half[3], double occupancy_value,
                                   unsigned min_depth)
{
	if (!m || !aabb_center || !aabb_half) return fail(UFOMAP_ERR_INVALID, "null argument");
	HIP_TRY(hipSetDevice(m->device));
	int rc = ufomap_map_wait(m);
	if (rc) return rc;
	if ((rc = phaseGuard(m))) return rc;
	const u32 L = m->g.L;
	if (L < min_depth) return UFOMAP_OK;  // OMB:495-497
	VolArgs a;
	for (int k = 0; k < 3; ++k) {
		a.vh[k] = aabb_half[k];
		a.vc[k] = aabb_center[k];
	}
	a.min_depth = min_depth;
	{
		// the root's box (OMB:499-505)
		const double half = m->g.hs[L];
		for (int k = 0; k < 3; ++k) {
			const double min1 = a.vc[k] - a.vh[k], max1 = a.vc[k] + a.vh[k], min2 = 0.0 - half, max2 = 0.0 + half;
			if (!(min1 <= max2) || !(min2 <= max1)) return UFOMAP_OK;  // no node intersects
		}
	}
	{
		const float nv = (float)std::log(occupancy_value / (1.0 - occupancy_value));  // toLogit (OMB:909) -> LogitType
		a.val = (nv < m->g.cmin) ? m->g.cmin : ((m->g.cmax < nv) ? m->g.cmax : nv);   // std::clamp (OMB:1154)
	}
	m->cs = m->stream;
	m->args = ScanArgs{};
	bool res_direct = false;
	ScanCtl init;
	memset(&init, 0, sizeof(init));
	for (int k = 0; k < 3; ++k) {
		init.aabb_min[k] = ~0ull;
		init.aabb_max[k] = 0ull;
	}
	*m->h_ctl = init;
	// (the one-workgroup walk writes the block's start state itself: no upload in front of it)
	const bool one_wg = L != min_depth && 0 == min_depth && m->opt_vol_fused;
	if (!one_wg) HIP_TRY(hipMemcpyAsync(m->b_ctl.p, m->h_ctl, sizeof(ScanCtl), hipMemcpyHostToDevice, m->stream));
	bool ctl_uploaded = !one_wg;
	m->ctl_clean = false;  // (the set's device control block no longer holds the fast path's start state)
	ScanCtl* ctl = m->b_ctl.as<ScanCtl>();
	m->scan_id += 1;
	if (L == min_depth) {
		hipLaunchKernelGGL(k_vol_root, gridFor((u64)m->t.mask + 1), dim3(256), 0, m->cs, m->t, m->g, a.val);
	} else {
		// records per level: nodes of that depth whose box can intersect the volume
		u64 level_cap[24] = {0}, total = 0;
		for (u32 cd = L; cd > min_depth; --cd) {
			long double c = 1;
			for (int k = 0; k < 3; ++k) c *= std::floor((long double)(2.0 * a.vh[k]) / (long double)nodeSize(m->g, cd)) + 2.0L;
			level_cap[cd] = c > 4e9L ? (u64)4e9 : (u64)c;
			if (cd == L) level_cap[cd] = 1;
			total += level_cap[cd];
		}
		if (total > 0x7FFFFFFFull || total * sizeof(VolRec) > m->scratch_limit)
			return fail(UFOMAP_ERR_CAPACITY, "setValueVolume: the volume covers too many nodes for the scratch limit");
		const u32 rcap = (u32)total;
		// every visited node may get a new children block
		{
			// (a node of depth cd gets a children block of level cd: tile groups for the depth-3 nodes, the first region above)
			Need need;
			need.blocks = total;
			for (u32 cd = L; cd > min_depth; --cd) {
				if (m->g.L < 4 || cd >= 4) need.upper += level_cap[cd];
				else if (3 == cd) need.groups += level_cap[cd];
			}
			if (!tableTakes(m, need)) {
				rc = growFor(m, need);
				if (rc) return rc;
			}
		}
		const u32 kcap = (u32)std::min<u64>(m->used_est + 8, 0x7FFFFFFFull);
		HIP_TRY(m->b_crec.reserve((size_t)rcap * sizeof(VolRec)));
		HIP_TRY(m->b_dlist.reserve((size_t)kcap * 4));
		VolRec* rec = m->b_crec.as<VolRec>();
		u32* kill = m->b_dlist.as<u32>();
		if (!(0 == min_depth && total <= 8192 && m->opt_vol_fused) && !ctl_uploaded) {
			HIP_TRY(hipMemcpyAsync(m->b_ctl.p, m->h_ctl, sizeof(ScanCtl), hipMemcpyHostToDevice, m->stream));
			ctl_uploaded = true;
		}
		if (0 == min_depth && total <= UFO_VOL_SMALL && m->opt_vol_fused && !ctl_uploaded) {
			// (the robot's box: the records of all levels in LDS, the blocks looked up at once, map_kernels.h: k_vol_small; reports itself)
			m->seq = ++m->latest_seq;
			m->h_res->err = ERR_NOT_STORED;
			*reinterpret_cast<volatile unsigned long long*>(m->h_res + 1) = 0ull;
			res_direct = true;
			hipLaunchKernelGGL(k_vol_small, dim3(1), dim3(1024), 0, m->cs, m->t, m->g, a, L, m->scan_id, ctl, m->h_res, (unsigned long long)m->seq);
		} else if (0 == min_depth && total <= 8192 && m->opt_vol_fused) {
			// (a small volume: one workgroup walks all levels, map_kernels.h: k_vol_all)
			// (... and reports to the pinned result block itself: no synchronisation, no read-back below)
			m->seq = ++m->latest_seq;
			m->h_res->err = ERR_NOT_STORED;
			*reinterpret_cast<volatile unsigned long long*>(m->h_res + 1) = 0ull;
			res_direct = true;
			hipLaunchKernelGGL(k_vol_all, dim3(1), dim3(1024), 0, m->cs, m->t, m->g, a, L, rec, rcap, kill, kcap, m->scan_id, ctl, m->h_res, (unsigned long long)m->seq, ctl_uploaded ? 0u : 1u);
		} else {
		hipLaunchKernelGGL(k_vol_begin, dim3(1), dim3(1), 0, m->cs, rec, ctl, L);
		for (u32 cd = L; cd > min_depth; --cd) {
			hipLaunchKernelGGL(k_vol_down, gridFor(level_cap[cd], 256, 4096), dim3(256), 0, m->cs, m->t, m->g, a, cd, rec, rcap, kill, kcap,
			                   m->scan_id, ctl);
			if (cd - 1 > min_depth && cd >= 2) hipLaunchKernelGGL(k_coarse_mark, dim3(1), dim3(1), 0, m->cs, ctl, cd - 2);
		}
		if (min_depth >= 1) {
			hipLaunchKernelGGL(k_vol_kill_mark, dim3(1), dim3(1), 0, m->cs, ctl, 0u);
			for (u32 l = 0; l < min_depth; ++l) {
				hipLaunchKernelGGL(k_vol_kill, gridFor(std::max<u64>(kcap, 256), 256, 4096), dim3(256), 0, m->cs, m->t, kill, kcap, ctl);
				hipLaunchKernelGGL(k_vol_kill_mark, dim3(1), dim3(1), 0, m->cs, ctl, 1u);
			}
		}
		for (u32 cd = min_depth + 1; cd <= L; ++cd)
			hipLaunchKernelGGL(k_vol_up, gridFor(level_cap[cd], 256, 4096), dim3(256), 0, m->cs, m->t, m->g, cd, rec, rcap, ctl);
		}
	}
	HIP_TRY(hipGetLastError());
	m->pending = true;
	if (res_direct) {
		m->done_by_flag = true;
		m->res_direct = true;
		volatile unsigned long long* done = reinterpret_cast<volatile unsigned long long*>(m->h_res + 1);
		const auto t0 = std::chrono::steady_clock::now();
		for (u32 spins = 0; *done != (unsigned long long)m->seq; ++spins) {
			if (0 == (spins & 1023u) && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) {
				HIP_TRY(hipStreamSynchronize(m->stream));
				if (*done != (unsigned long long)m->seq) return fail(UFOMAP_ERR_DEVICE, "setValueVolume: the walk did not report");
				break;
			}
		}
		std::atomic_thread_fence(std::memory_order_acquire);
	} else {
		HIP_TRY(hipStreamSynchronize(m->stream));
	}
	return finishPending(m);
}

int ufomap_map_clear_to(ufomap_map* m, double resolution, unsigned depth_levels)
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	if (depth_levels < 2 || depth_levels > 21) return fail(UFOMAP_ERR_INVALID, "depth_levels has to be [2, 21]");
	if (!(resolution > 0)) return fail(UFOMAP_ERR_INVALID, "resolution must be positive");
	int rc = ufomap_map_clear(m);
	if (rc) return rc;
	MapGeom& g = m->g;
	g.res = resolution;
	g.rf = 1.0 / resolution;
	g.L = depth_levels;
	m->t.L = depth_levels;  // (the table tells a key's level by the position of its sentinel bit: table.h)
	g.M = (u32)std::pow(2, depth_levels - 1);
	g.hs[0] = resolution / 2.0;
	g.hs[1] = resolution;
	for (unsigned i = 2; i < 23; ++i) g.hs[i] = g.hs[i - 1] * 2.0;
	m->spec_valid = false;
	return UFOMAP_OK;
}

int ufomap_map_get_sensor_model(ufomap_map* m, double out[6])
{
	if (!m || !out) return fail(UFOMAP_ERR_INVALID, "null argument");
	// toProb(LogitType) with LogitType = float: the stored double is narrowed first, std::exp(float) (OMB:734-744, 911)
	for (int k = 0; k < 6; ++k) out[k] = 1.0 / (1.0 + std::exp(-(float)m->model_log[k]));
	return UFOMAP_OK;
}

int ufomap_map_set_model_value(ufomap_map* m, int which, double probability)
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	if (which < 2 || which > 5) return fail(UFOMAP_ERR_INVALID, "which: 2 prob_hit, 3 prob_miss, 4 clamping_thres_min, 5 clamping_thres_max");
	if (!(probability > 0.0 && probability < 1.0)) return fail(UFOMAP_ERR_INVALID, "a probability strictly between 0 and 1 is needed (its logit is stored)");
	const int rc = ufomap_map_wait(m);
	if (rc) return rc;  // (the model is not changed under an integration that failed)
	m->model_log[which] = std::log(probability / (1.0 - probability));
	applyModel(m);
	return UFOMAP_OK;
}

int ufomap_map_enable_minmax_change_detection(ufomap_map* m, int enable)
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	int rc = ufomap_map_wait(m);
	if (!m->minmax_enabled && enable) {  // occupancy_map_base.h:793-795
		for (int a = 0; a < 3; ++a) {
			m->min_change[a] = m->g.hs[m->g.L];
			m->max_change[a] = -m->g.hs[m->g.L];
		}
	}
	m->minmax_enabled = enable != 0;
	return rc;
}

int ufomap_map_enable_change_detection(ufomap_map* m, int enable)
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	HIP_TRY(hipSetDevice(m->device));
	int rc = ufomap_map_wait(m);
	m->chg_enabled = enable != 0;
	if (m->chg_enabled && 0 == m->chg_cap) {
		HIP_TRY(m->b_changes.reserve((size_t)(1u << 20) * 8));
		m->chg_cap = 1u << 20;
	}
	return rc;
}

int ufomap_map_reset_change_detection(ufomap_map* m)
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	HIP_TRY(hipSetDevice(m->device));
	int rc = ufomap_map_wait(m);
	// (on the map stream and awaited: a fill on the null stream is not ordered with the handle's non-blocking streams -- an update
	// enqueued right after this call could log changes BEFORE the counter was zeroed and lose them)
	HIP_TRY(hipMemsetAsync(reinterpret_cast<char*>(m->b_root.p) + offsetof(MapRoot, n_changes), 0, 8, m->stream));
	HIP_TRY(hipStreamSynchronize(m->stream));
	return rc;
}

size_t ufomap_map_changes(ufomap_map* m, uint64_t* codes, uint8_t* depths, size_t cap)
{
	if (!m || ufomap_map_wait(m) < 0) return (size_t)-1;
	MapRoot root;
	if (hipMemcpy(&root, m->b_root.p, sizeof(MapRoot), hipMemcpyDeviceToHost) != hipSuccess) return (size_t)-1;
	if (root.chg_overflow) {
		fail(UFOMAP_ERR_CAPACITY, "change log overflowed (internal bound violated)");
		return (size_t)-1;
	}
	const size_t n = root.n_changes;
	std::vector<uint64_t> h(n);
	if (n && hipMemcpy(h.data(), m->b_changes.p, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return (size_t)-1;
	std::sort(h.begin(), h.end());
	h.erase(std::unique(h.begin(), h.end()), h.end());
	if (h.size() < n) {
		// keep the log compact: the set, once
		const u32 nn = (u32)h.size();
		if ((nn && hipMemcpy(m->b_changes.p, h.data(), h.size() * 8, hipMemcpyHostToDevice) != hipSuccess) ||
		    hipMemcpy(reinterpret_cast<char*>(m->b_root.p) + offsetof(MapRoot, n_changes), &nn, 4, hipMemcpyHostToDevice) != hipSuccess)
			return (size_t)-1;
	}
	// a record is (1 << 3*(L-depth)) | (code >> 3*depth) (table.h: ChangeLog): the sentinel's position gives the depth. Out in
	// (depth, code) order.
	std::vector<std::pair<uint8_t, uint64_t>> recs(h.size());
	for (size_t i = 0; i < h.size(); ++i) {
		const int p = 63 - __builtin_clzll(h[i] | 1ULL);
		recs[i] = {(uint8_t)(m->g.L - (u32)p / 3u), h[i] ^ (1ULL << p)};
	}
	std::sort(recs.begin(), recs.end());
	for (size_t i = 0; i < recs.size() && i < cap; ++i) {
		if (codes) codes[i] = recs[i].second;
		if (depths) depths[i] = recs[i].first;
	}
	return h.size();
}

size_t ufomap_map_iterate(ufomap_map* m, const double* aabb_center, const double* aabb_half, int occupied_space, int free_space,
                          int unknown_space, int contains, unsigned min_depth, int only_leaves, uint64_t* codes, uint8_t* depths,
                          float* logodds, uint8_t* rgb, uint8_t* flags, size_t cap)
{
	if (!m || ((nullptr == aabb_center) != (nullptr == aabb_half))) {
		fail(UFOMAP_ERR_INVALID, "null map / half a bounding volume");
		return (size_t)-1;
	}
	if (ufomap_map_wait(m) < 0) return (size_t)-1;
	auto bad = [&](hipError_t e) {
		if (e != hipSuccess) {
			fail(UFOMAP_ERR_DEVICE, hipGetErrorString(e));
			return true;
		}
		return false;
	};
	const u32 L = m->g.L;
	IterArgs a{};
	a.has_bv = aabb_center ? 1u : 0u;
	for (int k = 0; k < 3 && aabb_center; ++k) {
		a.vc[k] = aabb_center[k];
		a.vh[k] = aabb_half[k];
	}
	a.occ = occupied_space ? 1u : 0u;
	a.fre = free_space ? 1u : 0u;
	a.unk = unknown_space ? 1u : 0u;
	a.contains = contains ? 1u : 0u;
	a.min_depth = min_depth;
	a.only_leaf = only_leaves ? 1u : 0u;
	m->cs = m->stream;
	// one record per inner node that is descended into: at most the live blocks
	{
		MapRoot root;
		if (bad(hipMemcpy(&root, m->b_root.p, sizeof(MapRoot), hipMemcpyDeviceToHost))) return (size_t)-1;
		m->used_est = root.used;
	}
	const u32 rcap = (u32)std::min<u64>(m->used_est + 8, 0x7FFFFFF0ull);
	if (bad(m->b_crec.reserve((size_t)rcap * sizeof(IterRec)))) return (size_t)-1;
	IterRec* rec = m->b_crec.as<IterRec>();
	ScanCtl* ctl = m->b_ctl.as<ScanCtl>();
	DevBuf bc, bd, bo, br, bf;
	std::vector<uint64_t> hc;
	std::vector<uint8_t> hd, hf;
	std::vector<float> ho;
	std::vector<u32> hr;
	size_t total = 0;
	for (int pass = 0; pass < 2; ++pass) {
		const u32 ocap = pass ? (u32)total : 0u;
		if (pass) {
			if (0 == total) break;
			if (bad(bc.reserve((size_t)ocap * 8)) || bad(bd.reserve(ocap)) || bad(bo.reserve((size_t)ocap * 4)) || bad(br.reserve((size_t)ocap * 4)) ||
			    bad(bf.reserve(ocap)))
				return (size_t)-1;
		}
		if (bad(hipMemsetAsync(ctl, 0, sizeof(ScanCtl), m->stream))) return (size_t)-1;
		m->ctl_clean = false;  // (the set's device control block no longer holds the fast path's start state)
		const IterOut out{bc.as<u64>(), bd.as<u8>(), bo.as<float>(), br.as<u32>(), bf.as<u8>(), ocap};
		hipLaunchKernelGGL(k_iter_root, dim3(1), dim3(1), 0, m->stream, m->t, m->g, a, rec, rcap, out, ctl);
		for (u32 cd = L; cd >= 1 && cd > min_depth; --cd) {
			hipLaunchKernelGGL(k_iter_level, gridFor((u64)rcap * 8, 256, 2048), dim3(256), 0, m->stream, m->t, m->g, a, cd, rec, rcap, out, ctl);
			if (cd >= 2) hipLaunchKernelGGL(k_coarse_mark, dim3(1), dim3(1), 0, m->stream, ctl, cd - 2);
		}
		if (bad(hipMemcpyAsync(m->h_ctl, ctl, sizeof(ScanCtl), hipMemcpyDeviceToHost, m->stream)) || bad(hipStreamSynchronize(m->stream)))
			return (size_t)-1;
		if (m->h_ctl->err) {
			fail(UFOMAP_ERR_CAPACITY, "iterate: record list too small (internal bound violated)");
			return (size_t)-1;
		}
		total = m->h_ctl->n_codes;
		if (pass) {
			hc.resize(total);
			hd.resize(total);
			ho.resize(total);
			hr.resize(total);
			hf.resize(total);
			if (bad(hipMemcpy(hc.data(), bc.p, total * 8, hipMemcpyDeviceToHost)) || bad(hipMemcpy(hd.data(), bd.p, total, hipMemcpyDeviceToHost)) ||
			    bad(hipMemcpy(ho.data(), bo.p, total * 4, hipMemcpyDeviceToHost)) || bad(hipMemcpy(hr.data(), br.p, total * 4, hipMemcpyDeviceToHost)) ||
			    bad(hipMemcpy(hf.data(), bf.p, total, hipMemcpyDeviceToHost)))
				return (size_t)-1;
		} else if (0 == cap || (!codes && !depths && !logodds && !rgb && !flags)) {
			break;  // the caller only asked for the count
		}
	}
	bc.release();
	bd.release();
	bo.release();
	br.release();
	bf.release();
	if (hc.size() != total) return total;
	// pre-order: ascending position of the node's first voxel, a node before its descendants
	std::vector<size_t> order(total);
	for (size_t i = 0; i < total; ++i) order[i] = i;
	std::sort(order.begin(), order.end(), [&](size_t x, size_t y) {
		const uint64_t fx = hc[x] << (3 * hd[x]), fy = hc[y] << (3 * hd[y]);
		return fx != fy ? fx < fy : hd[x] > hd[y];
	});
	for (size_t i = 0; i < total && i < cap; ++i) {
		const size_t j = order[i];
		if (codes) codes[i] = hc[j];
		if (depths) depths[i] = hd[j];
		if (logodds) logodds[i] = ho[j];
		if (flags) flags[i] = hf[j];
		if (rgb) {
			rgb[3 * i] = (uint8_t)(hr[j] & 0xFF);
			rgb[3 * i + 1] = (uint8_t)((hr[j] >> 8) & 0xFF);
			rgb[3 * i + 2] = (uint8_t)((hr[j] >> 16) & 0xFF);
		}
	}
	return total;
}

int ufomap_map_wait(ufomap_map* m)
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	HIP_TRY(hipSetDevice(m->device));
	// (round 5) The scans that were waiting for company get their walk NOW, behind whatever is on the map stream -- its claim kernel
	// waits for their scan halves on the device -- instead of after the host has seen the scan stream and the earlier walks drain:
	// one host round trip and four launches less in the tail of every timed region (the driver's 20-step regions carry 10 us of
	// that tail per step). A flagged integration found below still makes everything behind it stand back and be repeated in order.
	if (!m->pending && 0 == countPendingAlts(m) && !m->sd_pending && UFOMAP_OK == m->async_status && !m->prev_flagged) {
		// (nothing in flight -- every kernel of the prep and scan streams belongs to an integration, and the last one has been joined:
		// the getters that call this before they read a host-side value do not pay two stream synchronisations for it)
		m->chain_ok = false;
		return UFOMAP_OK;
	}
	int early_frc = UFOMAP_OK;
	if (!m->prev_flagged && UFOMAP_OK == m->async_status && m->opt_wait_flush_first) early_frc = flushDeferred(m);
	HIP_TRY(hipStreamSynchronize(m->pstream));
	HIP_TRY(hipStreamSynchronize(m->sstream));
	// what has been enqueued first, oldest first (a flagged integration is repeated before anything newer is applied) ...
	int rc = joinEnqueued(m);
	if (!rc) rc = early_frc;
	{
		// ... then the scans whose tree update was still waiting for company
		const int frc = flushDeferred(m);
		if (!rc) rc = frc;
		const int rc1 = joinEnqueued(m);
		if (!rc) rc = rc1;
	}
	if (UFOMAP_OK == rc && UFOMAP_OK != m->async_status) {
		rc = m->async_status;
		g_err = "an earlier asynchronous integration failed";
	}
	m->async_status = UFOMAP_OK;
	m->prev_flagged = false;
	m->chain_ok = false;  // (whatever touches the map next: no walk takes scans from either side of it together)
	return rc;
}

int ufomap_map_done(ufomap_map* m)
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	// nothing finishes that has not been enqueued: the tree updates that were waiting for company go first
	{
		const int frc = flushDeferred(m);
		if (frc) return frc;
	}
	// the map stream runs the integrations in order: the most recent pending one is the last to finish
	const HandOver* newest = nullptr;
	if (!m->pending)
		for (int i = 0; i < kAlt; ++i)
			if (m->alt[i].pending && (!newest || m->alt[i].seq > newest->seq)) newest = &m->alt[i];
	const bool any = m->pending || newest;
	if (!any) return 1;
	const bool by_flag = m->pending ? m->done_by_flag : newest->done_by_flag;
	if (by_flag) {
		// (fast path: the integration's end is a word in pinned memory, not an event)
		const ScanCtl* hr = m->pending ? m->h_res : newest->h_res;
		const uint64_t sq = m->pending ? m->seq : newest->seq;
		return *reinterpret_cast<const volatile unsigned long long*>(hr + 1) == (unsigned long long)sq ? 1 : 0;
	}
	hipEvent_t ev = m->pending ? m->done_ev : newest->done_ev;
	if (!ev) return 1;
	hipError_t e = hipEventQuery(ev);
	if (e == hipSuccess) return 1;
	if (e == hipErrorNotReady) return 0;
	return fail(UFOMAP_ERR_DEVICE, hipGetErrorString(e));
}

static bool recLess(const uint64_t* codes, const uint8_t* depths, size_t a, size_t b)
{
	return depths[a] != depths[b] ? depths[a] < depths[b] : codes[a] < codes[b];
}

static size_t exportCommon(ufomap_map* m, bool inner, int include_unknown, uint64_t* codes, uint8_t* depths, float* logodds,
                           uint8_t* flags, uint8_t* rgb, size_t cap)
{
	if (!m) {
		fail(UFOMAP_ERR_INVALID, "null map");
		return (size_t)-1;
	}
	if (ufomap_map_wait(m) < 0) return (size_t)-1;
	auto bad = [&](hipError_t e) {
		if (e != hipSuccess) {
			fail(UFOMAP_ERR_DEVICE, hipGetErrorString(e));
			return true;
		}
		return false;
	};
	// pass 1: count; pass 2: dump everything; host sorts into canonical order
	DevBuf dcbuf;
	if (bad(dcbuf.reserve(sizeof(DumpCtl)))) return (size_t)-1;
	DumpCtl* d_dc = dcbuf.as<DumpCtl>();
	DumpCtl h{};
	unsigned long long n_live = 0;
	size_t total = 0;
	std::vector<uint64_t> hc;
	std::vector<uint8_t> hd, hf;
	std::vector<float> ho;
	std::vector<u32> hr;
	for (int pass = 0; pass < 2; ++pass) {
		unsigned long long dcap = pass ? total : 0;
		if (pass && 0 == total) break;
		DevBuf bc, bd, bo, bf, br;
		if (pass) {
			if (bad(bc.reserve(dcap * 8)) || bad(bd.reserve(dcap)) || bad(bo.reserve(dcap * 4)) || bad(bf.reserve(dcap)) ||
			    bad(br.reserve(dcap * 4)))
				return (size_t)-1;
		}
		if (bad(hipMemsetAsync(d_dc, 0, sizeof(DumpCtl), m->stream))) return (size_t)-1;
		if (inner)
			hipLaunchKernelGGL(k_export_inner, gridFor(((u64)m->t.mask + UFO_EXPORT_SLOTS) / UFO_EXPORT_SLOTS), dim3(256), 0, m->stream, m->t, m->g, bc.as<u64>(),
			                   bd.as<uint8_t>(), bo.as<float>(), bf.as<uint8_t>(), br.as<u32>(), dcap, d_dc);
		else
			hipLaunchKernelGGL(k_export_leaves, gridFor(((u64)m->t.mask + UFO_EXPORT_SLOTS) / UFO_EXPORT_SLOTS), dim3(256), 0, m->stream, m->t, m->g, include_unknown,
			                   bc.as<u64>(), bd.as<uint8_t>(), bo.as<float>(), br.as<u32>(), dcap, d_dc);
		if (bad(hipMemcpyAsync(&h, d_dc, sizeof(DumpCtl), hipMemcpyDeviceToHost, m->stream))) return (size_t)-1;
		if (bad(hipStreamSynchronize(m->stream))) return (size_t)-1;
		if (!pass) {
			total = h.n_out;
			n_live = h.n_live;
		} else {
			hc.resize(total);
			hd.resize(total);
			ho.resize(total);
			hr.resize(total);
			if (bad(hipMemcpy(hc.data(), bc.p, total * 8, hipMemcpyDeviceToHost)) ||
			    bad(hipMemcpy(hd.data(), bd.p, total, hipMemcpyDeviceToHost)) ||
			    bad(hipMemcpy(ho.data(), bo.p, total * 4, hipMemcpyDeviceToHost)) ||
			    bad(hipMemcpy(hr.data(), br.p, total * 4, hipMemcpyDeviceToHost)))
				return (size_t)-1;
			if (inner) {
				hf.resize(total);
				if (bad(hipMemcpy(hf.data(), bf.p, total, hipMemcpyDeviceToHost))) return (size_t)-1;
			}
		}
		bc.release();
		bd.release();
		bo.release();
		bf.release();
		br.release();
	}
	dcbuf.release();
	// the root alone: a map whose root block is absent or collapsed is a single leaf
	MapRoot root;
	if (bad(hipMemcpy(&root, m->b_root.p, sizeof(MapRoot), hipMemcpyDeviceToHost))) return (size_t)-1;
	if (!inner && 0 == n_live) {  // n_live is counted by the leaf kernel: no live block <=> the root is a leaf
		bool unknown = isUnknownV(m->g, root.occ);
		if (include_unknown || !unknown) {
			hc.push_back(0);
			hd.push_back((uint8_t)m->g.L);
			ho.push_back(root.occ);
			hr.push_back(root.rgb);
			total += 1;
		}
	}
	std::vector<size_t> order(total);
	for (size_t i = 0; i < total; ++i) order[i] = i;
	std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return recLess(hc.data(), hd.data(), a, b); });
	size_t nout = std::min(total, cap);
	for (size_t i = 0; i < nout; ++i) {
		size_t j = order[i];
		if (codes) codes[i] = hc[j];
		if (depths) depths[i] = hd[j];
		if (logodds) logodds[i] = ho[j];
		if (flags && inner) flags[i] = hf[j];
		if (rgb) {
			rgb[3 * i] = (uint8_t)(hr[j] & 0xFF);
			rgb[3 * i + 1] = (uint8_t)((hr[j] >> 8) & 0xFF);
			rgb[3 * i + 2] = (uint8_t)((hr[j] >> 16) & 0xFF);
		}
	}
	return total;
}

size_t ufomap_map_export_leaves(ufomap_map* m, int include_unknown, uint64_t* codes, uint8_t* depths, float* logodds,
                                uint8_t* rgb, size_t cap)
{
	return exportCommon(m, false, include_unknown, codes, depths, logodds, nullptr, rgb, cap);
}

size_t ufomap_map_export_inner(ufomap_map* m, uint64_t* codes, uint8_t* depths, float* logodds, uint8_t* flags, uint8_t* rgb,
                               size_t cap)
{
	return exportCommon(m, true, 1, codes, depths, logodds, flags, rgb, cap);
}

int ufomap_map_minmax_change(ufomap_map* m, double mn[3], double mx[3])
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	int rc = ufomap_map_wait(m);
	for (int a = 0; a < 3; ++a) {
		mn[a] = m->min_change[a];
		mx[a] = m->max_change[a];
	}
	return rc;
}

int ufomap_map_reset_minmax_change(ufomap_map* m)
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	int rc = ufomap_map_wait(m);
	for (int a = 0; a < 3; ++a) {
		m->min_change[a] = m->g.hs[m->g.L];
		m->max_change[a] = -m->g.hs[m->g.L];
	}
	return rc;
}

int ufomap_map_stats(ufomap_map* m, uint64_t* n_inner, uint64_t* n_leaf, uint64_t* bytes)
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	int rc = ufomap_map_wait(m);
	if (rc) return rc;
	DevBuf dcbuf;
	HIP_TRY(dcbuf.reserve(sizeof(DumpCtl)));
	HIP_TRY(hipMemsetAsync(dcbuf.p, 0, sizeof(DumpCtl), m->stream));
	hipLaunchKernelGGL(k_export_leaves, gridFor(((u64)m->t.mask + UFO_EXPORT_SLOTS) / UFO_EXPORT_SLOTS), dim3(256), 0, m->stream, m->t, m->g, 1, (u64*)nullptr,
	                   (uint8_t*)nullptr, (float*)nullptr, (u32*)nullptr, 0ull, dcbuf.as<DumpCtl>());
	DumpCtl h{};
	HIP_TRY(hipMemcpyAsync(&h, dcbuf.p, sizeof(DumpCtl), hipMemcpyDeviceToHost, m->stream));
	HIP_TRY(hipStreamSynchronize(m->stream));
	dcbuf.release();
	if (n_inner) *n_inner = h.n_live;
	if (n_leaf) *n_leaf = h.n_live ? h.n_leaf : 1;
	if (bytes) {
		u64 cap = (u64)m->t.mask + 1;
		*bytes = cap * (UFO_SLOT_BYTES + 8 + 8 + (m->g.color ? 32 + 4 : 0)) + (u64)m->t.nG * 8 + 512;  // record + tmax + last-update record (+ colours); tile directory
	}
	return UFOMAP_OK;
}

int ufomap_map_digest(ufomap_map* m, int include_unknown, uint64_t out[6])
{
	if (!m || !out) return fail(UFOMAP_ERR_INVALID, "null argument");
	int rc = ufomap_map_wait(m);
	if (rc) return rc;
	DevBuf db;
	HIP_TRY(db.reserve(6 * 8));
	hipError_t e = hipMemsetAsync(db.p, 0, 6 * 8, m->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_digest, gridFor((u64)m->t.mask + 1), dim3(256), 0, m->stream, m->t, m->g, include_unknown,
		                   db.as<unsigned long long>());
		e = hipMemcpyAsync(out, db.p, 6 * 8, hipMemcpyDeviceToHost, m->stream);
	}
	if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
	db.release();
	if (e != hipSuccess) return fail(UFOMAP_ERR_DEVICE, hipGetErrorString(e));
	if (0 == out[3]) {
		// no live block: the map is its root, one leaf of depth depth_levels (as in the leaf export)
		MapRoot root;
		HIP_TRY(hipMemcpy(&root, m->b_root.p, sizeof(MapRoot), hipMemcpyDeviceToHost));
		if (include_unknown || !isUnknownV(m->g, root.occ)) {
			const u64 h = digestRecord(0, m->g.L, root.occ, root.rgb, 0u);
			out[0] = 1;
			out[1] = h;
			out[2] = h;
		}
	}
	return UFOMAP_OK;
}

size_t ufomap_map_last_hits(ufomap_map* m, uint64_t* codes, size_t cap)
{
	if (!m || ufomap_map_wait(m) < 0) return (size_t)-1;
	size_t n = (size_t)m->counts[3];
	std::vector<uint64_t> h(n);
	if (n && m->hit_grid) {
		// the last integration ran on the fast path: its hit voxels are the set bits of the scan's hit grid
		ScanCtl* ctl = m->b_ctl.as<ScanCtl>();
		if (m->b_hit_code.reserve(n * 8) != hipSuccess || hipMemsetAsync(&ctl->n_codes, 0, 4, m->stream) != hipSuccess) return (size_t)-1;
		hipLaunchKernelGGL(k_grid_codes_bits, gridFor(m->fgeo.gr.bytes >> 2, 256, 8192), dim3(256), 0, m->stream, m->g, m->fgeo.gr, m->b_gridH.as<u32>(),
		                   m->b_hit_code.as<u64>(), (u32)n, ctl);
		if (hipStreamSynchronize(m->stream) != hipSuccess) return (size_t)-1;
		(void)hipMemsetAsync(&ctl->n_codes, 0, 4, m->stream);  // (the fast path's start state of the block)
	}
	if (n && hipMemcpy(h.data(), m->b_hit_code.p, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return (size_t)-1;
	std::sort(h.begin(), h.end());
	if (codes) memcpy(codes, h.data(), std::min(n, cap) * 8);
	return n;
}

size_t ufomap_map_last_misses(ufomap_map* m, uint64_t* codes, size_t cap)
{
	if (!m || ufomap_map_wait(m) < 0) return (size_t)-1;
	if (!m->haveM) return 0;
	if (m->vol && !m->opt_vol_keep) {
		fail(UFOMAP_ERR_UNSUPPORTED, "the ray cells of a scan on the volume path are not kept (option vol_keep = 1 keeps them)");
		return (size_t)-1;
	}
	ScanCtl* ctl = m->b_ctl.as<ScanCtl>();
	std::vector<uint64_t> h;
	u32 total = 0;
	for (int pass = 0; pass < 2; ++pass) {
		if (hipMemsetAsync(&ctl->n_codes, 0, 4, m->stream) != hipSuccess) return (size_t)-1;
		if (pass && m->b_codes.reserve((size_t)total * 8) != hipSuccess) return (size_t)-1;
		if (m->vol) {
			hipLaunchKernelGGL(k_vcodes, gridFor((u64)m->vol_count * 8u, 256, 8192), dim3(256), 0, m->stream, m->vplan.vg, m->b_vMm.as<u64>(), m->b_vlist.as<u32>(),
			                   m->vol_count, pass ? m->b_codes.as<u64>() : (u64*)nullptr, pass ? total : 0u, ctl);
		} else if (2 == m->gridM.layout) {
			const u64 slots = m->miss_set_slots;
			const MissSet ms{m->b_gridM.as<u64>(), reinterpret_cast<u32*>(m->b_gridM.as<u64>() + slots), (u32)(slots - 1), nullptr};
			hipLaunchKernelGGL(k_set_codes, gridFor(slots, 256, 8192), dim3(256), 0, m->stream, ms, pass ? m->b_codes.as<u64>() : (u64*)nullptr,
			                   pass ? total : 0u, ctl);
		} else if (1 == m->gridM.layout)
			hipLaunchKernelGGL(k_grid_codes_bits, gridFor(m->gridM.bytes >> 2, 256, 8192), dim3(256), 0, m->stream, m->g, m->gridM,
			                   m->b_gridM.as<u32>(), pass ? m->b_codes.as<u64>() : (u64*)nullptr, pass ? total : 0u, ctl);
		else
			hipLaunchKernelGGL(k_grid_codes, gridFor(m->gridM.bytes >> 2, 256, 8192), dim3(256), 0, m->stream, m->g, m->gridM,
			                   m->b_gridM.as<u32>(), pass ? m->b_codes.as<u64>() : (u64*)nullptr, pass ? total : 0u, ctl);
		if (hipMemcpyAsync(&total, &ctl->n_codes, 4, hipMemcpyDeviceToHost, m->stream) != hipSuccess) return (size_t)-1;
		if (hipStreamSynchronize(m->stream) != hipSuccess) return (size_t)-1;
		if (0 == total) break;
	}
	h.resize(total);
	if (total && hipMemcpy(h.data(), m->b_codes.p, (size_t)total * 8, hipMemcpyDeviceToHost) != hipSuccess) return (size_t)-1;
	std::sort(h.begin(), h.end());
	m->counts[4] = total;
	if (codes) memcpy(codes, h.data(), std::min<size_t>(total, cap) * 8);
	return total;
}

int ufomap_map_last_counts(ufomap_map* m, uint64_t counts[8])
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	int rc = ufomap_map_wait(m);
	for (int k = 0; k < 8; ++k) counts[k] = m->counts[k];
	return rc;
}

int ufomap_map_set_profiling(ufomap_map* m, int on)
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	m->profiling = on != 0;
	return UFOMAP_OK;
}

int ufomap_map_kernel_times(ufomap_map* m, const char** names, uint64_t* launches, double* total_ms, int cap)
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	(void)ufomap_map_wait(m);
	drainEvents(m);
	int n = (int)m->stats.size();
	for (int i = 0; i < n && i < cap; ++i) {
		if (names) names[i] = m->stats[i].name;
		if (launches) launches[i] = m->stats[i].launches;
		if (total_ms) total_ms[i] = m->stats[i].total_ms;
	}
	return n;
}

int ufomap_map_reset_kernel_times(ufomap_map* m)
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	(void)ufomap_map_wait(m);
	drainEvents(m);
	for (KernelStat& s : m->stats) {
		s.launches = 0;
		s.total_ms = 0;
	}
	return UFOMAP_OK;
}

int ufomap_map_scan_keys(ufomap_map* m, const double sensor_origin[3], const double* d_xyz, size_t n, double max_range,
                         unsigned depth, int discrete, int simple_ray_casting, ufomap_keys_info* info)
{
	return ufomap_map_scan_keys_rgb(m, sensor_origin, d_xyz, nullptr, n, max_range, depth, discrete, simple_ray_casting, info);
}

int scanKeysCore(ufomap_map* m, const double sensor_origin[3], const double* d_xyz, const uint8_t* d_rgb, size_t n, double max_range, unsigned depth,
                 int discrete, int simple_ray_casting, ufomap_keys_info* info, unsigned early_stopping = 0);
int applyKeysBatchCore(ufomap_map* m, const void* const* d_lists, const ufomap_keys_info* infos, int n_lists, bool sync);

int ufomap_map_scan_keys_rgb(ufomap_map* m, const double sensor_origin[3], const double* d_xyz, const uint8_t* d_rgb, size_t n, double max_range,
                             unsigned depth, int discrete, int simple_ray_casting, ufomap_keys_info* info)
{
	if (!m || !info) return fail(UFOMAP_ERR_INVALID, "null argument");
	if (d_rgb && !m->g.color) return fail(UFOMAP_ERR_INVALID, "colours for a map without colour");
	if (d_rgb && !discrete) return fail(UFOMAP_ERR_UNSUPPORTED, "colour integration exists for the discrete integrator only (occupancy_map_color.h:177)");
	if (d_rgb && 0 != depth) return fail(UFOMAP_ERR_UNSUPPORTED, "update lists with colour: insert depth 0 only");
	HIP_TRY(hipSetDevice(m->device));
	// Ray casting never reads the map: it runs on the scan stream with its own hand-over set while an update
	// enqueued earlier (asynchronous insert / apply_keys_batch) may still be walking the tree on the map stream.
	HIP_TRY(hipStreamSynchronize(m->sstream));
	(void)rotateSets(m);
	m->args = ScanArgs{};
	return scanKeysCore(m, sensor_origin, d_xyz, d_rgb, n, max_range, depth, discrete, simple_ray_casting, info);
}

// (the scan itself, on the current hand-over set)
int scanKeysCore(ufomap_map* m, const double sensor_origin[3], const double* d_xyz, const uint8_t* d_rgb, size_t n, double max_range, unsigned depth,
                 int discrete, int simple_ray_casting, ufomap_keys_info* info, unsigned early_stopping)
{
	memset(info, 0, sizeof(*info));
	info->depth = depth;
	m->fast = false;
	m->deferred = false;
	m->batch_world = 0;
	struct RestoreStream {  // (every exit: the helpers launch on the map stream again, finished events are taken in)
		ufomap_map* m;
		~RestoreStream()
		{
			drainEvents(m);
			m->cs = m->stream;
		}
	} restore{m};
	m->cs = m->sstream;
	u32 n_hits = 0, n_rays = 0;
	m->keys_mode = true;  // (the scan's ray cells are wanted as an update list: not the volume path's brick grids)
	int rc = scanPhase(m, sensor_origin, d_xyz, d_rgb, n, max_range, depth, discrete, simple_ray_casting, early_stopping, &n_hits, &n_rays);
	m->keys_mode = false;
	if (rc || 0 == n) return rc;
	u64 capH = 0, capM = 0;
	// insert depth 0: ONE list, a block with hits and misses appears once with both masks (flagged in `reserved`)
	const bool merged = 0 == depth && 0 != m->opt_merge;
	rc = extractPhase(m, n_hits, n_rays, &capH, &capM, merged);
	if (rc) return rc;
	rc = readCtl(m);  // on the scan stream: waits for the extraction
	if (rc) return rc;
	rc = ctlError(m);  // e.g. a runaway ray
	if (rc) return rc;
	// compact: miss entries directly behind the hit entries
	const u32 nh = m->h_ctl->n_entries[0], nm = m->h_ctl->n_entries[1];
	info->reserved = merged ? 1u : 0u;
	if (!merged && nh != capH && nm) {
		HIP_TRY(m->b_codes.reserve((size_t)nm * sizeof(Entry)));
		HIP_TRY(hipMemcpyAsync(m->b_codes.p, m->b_entries.as<Entry>() + capH, (size_t)nm * sizeof(Entry), hipMemcpyDeviceToDevice, m->cs));
		HIP_TRY(hipMemcpyAsync(m->b_entries.as<Entry>() + nh, m->b_codes.p, (size_t)nm * sizeof(Entry), hipMemcpyDeviceToDevice, m->cs));
		HIP_TRY(hipStreamSynchronize(m->cs));
	}
	info->n_hit = nh;
	info->n_miss = nm;
	if (d_rgb && nh) {
		// colour section behind the records: 8 colours per hit record (merged list: every record)
		const HitHash hh{m->b_hh_keys.as<u64>(), reinterpret_cast<u32*>(m->b_hh_keys.as<u64>() + ((size_t)m->hh_mask + 1)), m->hh_mask};
		hipLaunchKernelGGL(k_list_colors, gridFor(nh), dim3(256), 0, m->cs, m->g, m->b_entries.as<Entry>(), nh, hh, d_rgb,
		                   reinterpret_cast<u32*>(m->b_entries.as<Entry>() + ((size_t)nh + nm)));
		HIP_TRY(hipStreamSynchronize(m->cs));
		info->reserved |= 2u;
	}
	for (int a = 0; a < 3; ++a) {
		info->nb_hit[a] = m->haveH ? m->gridH.nb[a] : 0;
		info->nb_miss[a] = m->haveM ? m->gridM.nb[a] : 0;
	}
	m->counts[2] = m->h_ctl->n_steps;
	m->counts[5] = (u64)nh + nm;
	drainEvents(m);
	m->cs = m->stream;
	return UFOMAP_OK;
}

int ufomap_map_get_keys(ufomap_map* m, void* d_dst, size_t cap_entries, const ufomap_keys_info* info)
{
	if (!m || !info) return fail(UFOMAP_ERR_INVALID, "null argument");
	size_t tot = (size_t)info->n_hit + info->n_miss;
	if (info->reserved & 2u) tot += 2 * (size_t)info->n_hit;  // colour section: 32 bytes per hit record
	if (tot > cap_entries) return fail(UFOMAP_ERR_CAPACITY, "destination too small for the update list");
	HIP_TRY(hipSetDevice(m->device));
	if (tot) HIP_TRY(hipMemcpyAsync(d_dst, m->b_entries.p, tot * sizeof(Entry), hipMemcpyDeviceToDevice, m->sstream));
	HIP_TRY(hipStreamSynchronize(m->sstream));
	return UFOMAP_OK;
}

int ufomap_map_apply_keys(ufomap_map* m, const void* d_entries, const ufomap_keys_info* info)
{
	if (!m || !info) return fail(UFOMAP_ERR_INVALID, "null argument");
	if (m->g.color) return ufomap_map_apply_keys_batch(m, &d_entries, info, 1);  // (checks that the list carries colours)
	if (const int grc = phaseGuard(m)) return grc;
	if (info->depth >= m->g.L) return fail(UFOMAP_ERR_INVALID, "depth must be < depth_levels");
	HIP_TRY(hipSetDevice(m->device));
	{
		int prc = ufomap_map_wait(m);
		if (prc) return prc;
	}
	if (info->reserved & 1u) return ufomap_map_apply_keys_batch(m, &d_entries, info, 1);  // merged list
	const u32 nh = info->n_hit, nm = info->n_miss;
	if (0 == nh + nm) return UFOMAP_OK;
	m->cs = m->stream;
	m->args = ScanArgs{};
	// fresh control block: the entry counts are known exactly here
	ScanCtl init;
	memset(&init, 0, sizeof(init));
	init.n_entries[0] = nh;
	init.n_entries[1] = nm;
	for (int a = 0; a < 3; ++a) {
		init.aabb_min[a] = ~0ull;
		init.aabb_max[a] = 0ull;
	}
	*m->h_ctl = init;
	HIP_TRY(hipMemcpyAsync(m->b_ctl.p, m->h_ctl, sizeof(ScanCtl), hipMemcpyHostToDevice, m->stream));
	m->ctl_clean = false;  // (the set's device control block no longer holds the fast path's start state)
	const Entry* ent = static_cast<const Entry*>(d_entries);
	int rc = sizeTable(m, ent, nh, info->nb_hit, ent + nh, nm, info->nb_miss, info->depth);
	if (rc) return rc;
	const float miss = (float)(m->g.miss_log / double((2.0 * info->depth) + 1));
	rc = applyEntries(m, ent, nh, 0, 1, info->nb_hit, m->g.hit, nullptr, false, nh, nm);
	if (rc) return rc;
	rc = applyEntries(m, ent + nh, nm, 1, info->depth + 1, info->nb_miss, miss, nullptr, false, nh, nm);
	if (rc) return rc;
	m->pending = true;
	HIP_TRY(hipStreamSynchronize(m->stream));
	return finishPending(m);
}

int ufomap_map_apply_keys_batch(ufomap_map* m, const void* const* d_lists, const ufomap_keys_info* infos, int n_lists)
{
	if (!m || !d_lists || !infos || n_lists < 0) return fail(UFOMAP_ERR_INVALID, "null argument");
	if (m->g.color)
		for (int j = 0; j < n_lists; ++j)
			if (infos[j].n_hit && !(infos[j].reserved & 2u))
				return fail(UFOMAP_ERR_INVALID, "a colour map needs update lists with a colour section (ufomap_map_scan_keys_rgb)");
	if (const int grc = phaseGuard(m)) return grc;
	if (n_lists > 128) return fail(UFOMAP_ERR_INVALID, "at most 128 update lists per batch");
	for (int j = 0; j < n_lists; ++j)
		if (0 != infos[j].depth) return fail(UFOMAP_ERR_UNSUPPORTED, "apply_keys_batch: insert depth 0 only (apply deeper scans one by one)");
	HIP_TRY(hipSetDevice(m->device));
	{
		// join what is in flight on the map stream (not the scan stream: ray casting of the next batch may overlap)
		int prc = joinOlder(m);
		if (!prc && m->pending) {  // (an update enqueued on this very set: scan_keys was not called in between)
			HIP_TRY(hipStreamSynchronize(m->stream));
			prc = finishPending(m);
		}
		if (UFOMAP_OK == prc && UFOMAP_OK != m->async_status) {
			prc = m->async_status;
			g_err = "an earlier asynchronous integration failed";
		}
		m->async_status = UFOMAP_OK;
		if (prc) return prc;
	}
	return applyKeysBatchCore(m, d_lists, infos, n_lists, false);
}

// (the update itself, on the current hand-over set; nothing is joined here. sync: wait for it whatever option async_apply says)
int applyKeysBatchCore(ufomap_map* m, const void* const* d_lists, const ufomap_keys_info* infos, int n_lists, bool sync)
{
	u64 total = 0;
	for (int j = 0; j < n_lists; ++j) total += (u64)infos[j].n_hit + infos[j].n_miss;
	if (0 == total) return UFOMAP_OK;
	if (total > 0x7FFFFFFFull) return fail(UFOMAP_ERR_CAPACITY, "update lists exceed 2^31 entries");
	m->cs = m->stream;
	m->args = ScanArgs{};
	m->chain_ok = false;
	m->fast = false;
	ScanCtl init;
	memset(&init, 0, sizeof(init));
	for (int a = 0; a < 3; ++a) {
		init.aabb_min[a] = ~0ull;
		init.aabb_max[a] = 0ull;
	}
	*m->h_ctl = init;
	HIP_TRY(hipMemcpyAsync(m->b_ctl.p, m->h_ctl, sizeof(ScanCtl), hipMemcpyHostToDevice, m->stream));
	m->ctl_clean = false;  // (the set's device control block no longer holds the fast path's start state)
	ScanCtl* ctl = m->b_ctl.as<ScanCtl>();
	ScanCtl::PhaseCtr* pc = &ctl->ph[0];
	// sub-lists in application order: scan 0 hits, scan 0 misses, scan 1 hits, ...; their counts live on the device
	struct Sub {
		const Entry* ent;
		const u32* rgb;          // colour section of the list (hit / merged sub-lists of a colour map), else nullptr
		u32 n, off, mode, scan;  // mode: 0 misses, 1 hits, 2 merged (k_apply_values)
		const i32* nbA;          // grid the entries lie in ...
		const i32* nbB;          // ... or, merged lists, this one (nullptr otherwise)
	};
	auto subNew = [&](const Sub& sb, u64 cnt) {
		Need b = needBound(m, std::min<u64>(cnt, sb.n), sb.nbA, 1);
		if (sb.nbB) b += needBound(m, std::min<u64>(cnt, sb.n), sb.nbB, 1);
		return b;
	};
	std::vector<Sub> subs;
	std::vector<u32> h_cnt;
	u32 off = 0;
	Need new_bound;
	for (int j = 0; j < n_lists; ++j) {
		const Entry* e = static_cast<const Entry*>(d_lists[j]);
		const u32 nh = infos[j].n_hit, nm = infos[j].n_miss;
		const u32* rgbs = (m->g.color && (infos[j].reserved & 2u)) ? reinterpret_cast<const u32*>(e + ((size_t)nh + nm)) : nullptr;
		if (infos[j].reserved & 1u) {
			// merged list: n_hit records, each with the hit and the miss mask of its block
			if (nm) return fail(UFOMAP_ERR_INVALID, "merged update list with a separate miss list");
			if (nh) {
				subs.push_back(Sub{e, rgbs, nh, off, 2u, (u32)j, infos[j].nb_hit, infos[j].nb_miss});
				off += nh;
			}
			continue;
		}
		if (nh) {
			subs.push_back(Sub{e, rgbs, nh, off, 1u, (u32)j, infos[j].nb_hit, nullptr});
			off += nh;
		}
		if (nm) {
			subs.push_back(Sub{e + nh, nullptr, nm, off, 0u, (u32)j, infos[j].nb_miss, nullptr});
			off += nm;
		}
	}
	for (const Sub& sb : subs) new_bound += subNew(sb, sb.n);
	for (const Sub& sb : subs) h_cnt.push_back(sb.n);
	HIP_TRY(m->b_crec.reserve(h_cnt.size() * 4 + 16));
	{
		// the counts travel as kernel arguments (captured at launch): an asynchronous copy from this local vector could
		// still be reading it after an async_apply call has returned
		SmallCounts sc{};
		for (size_t k = 0; k < h_cnt.size() && k < 256; ++k) sc.v[k] = h_cnt[k];
		hipLaunchKernelGGL(k_store_counts, dim3(1), dim3(256), 0, m->stream, sc, (u32)std::min<size_t>(h_cnt.size(), 256), m->b_crec.as<u32>());
	}
	const u32* d_cnt = m->b_crec.as<u32>();
	// node table: every entry new is a true upper bound; on a warm map count the missing blocks before growing
	m->scan_new_bound = new_bound;
	if (!tableTakes(m, new_bound)) {
		const bool cheap = ((m->used_g + new_bound.groups) * (u64)UFO_GROUP + m->used_u + new_bound.upper) * 2 <= (1ull << 22);  // (slots: sizeTable)
		if (!cheap) {
			u32* d_miss = reinterpret_cast<u32*>(&ctl->dbg[60]);
			HIP_TRY(hipMemsetAsync(d_miss, 0, 8, m->cs));
			for (size_t k = 0; k < subs.size(); ++k)
				hipLaunchKernelGGL(k_count_missing, gridFor(subs[k].n), dim3(256), 0, m->cs, m->t, subs[k].ent, d_cnt + k, subs[k].n, d_miss);
			int rc = readCtl(m);
			if (rc) return rc;
			u32 cnt = 0;
			memcpy(&cnt, &m->h_ctl->dbg[60], 4);
			Need b;
			for (const Sub& sb : subs) b += subNew(sb, cnt);
			m->scan_new_bound = new_bound = needMin(new_bound, b);
		}
		if (!tableTakes(m, new_bound)) {
			int rc = growFor(m, new_bound);
			if (rc) return rc;
		}
	}
	if (m->chg_enabled) {
		const int crc = ensureChangeCap(m, total * 8);
		if (crc) return crc;
	}
	const ChangeLog cl = changeLog(m);
	m->scan_id += 1;  // ONE phase for the whole batch
	const u32 newcap = (u32)std::min<u64>(new_bound.blocks, 0xFFFFFFFFull);
	HIP_TRY(m->b_ent_slot.reserve((size_t)total * 4));
	HIP_TRY(m->b_newlist.reserve((size_t)newcap * 4));
	HIP_TRY(m->b_wl0.reserve((size_t)total * 4 + 32));
	HIP_TRY(m->b_wl1.reserve((size_t)total * 4 + 32));
	u32* wl[2] = {m->b_wl0.as<u32>(), m->b_wl1.as<u32>()};
	for (size_t k = 0; k < subs.size(); ++k) {
		ProfScope ps(m, "k_ensure");
		hipLaunchKernelGGL(k_ensure, gridFor(subs[k].n), dim3(256), 0, m->cs, m->t, m->g, subs[k].ent, d_cnt + k, 0xFFFFFFFFu, 0xFFFFFFFFu, m->scan_id,
		                   m->b_ent_slot.as<u32>() + subs[k].off, m->b_newlist.as<u32>(), newcap, pc, ctl, (const ScanCtl*)nullptr);
	}
	{
		ProfScope ps(m, "k_init_new");
		hipLaunchKernelGGL(k_init_new, gridFor(std::min<u64>(newcap, total)), dim3(256), 0, m->cs, m->t, m->g, m->b_newlist.as<u32>(), newcap,
		                   m->scan_id, pc, ctl);
	}
	const float miss = (float)m->g.miss_log;  // insert depth 0 (OMB:311)
	for (size_t k = 0; k < subs.size(); ++k) {
		ProfScope ps(m, "k_apply_values");
		const u64 time_hi = (u64)subs[k].scan << 30;
		hipLaunchKernelGGL(k_apply_values, gridFor(subs[k].n), dim3(256), 0, m->cs, m->t, m->g, subs[k].ent, d_cnt + k,
		                   m->b_ent_slot.as<u32>() + subs[k].off, m->g.hit, miss, subs[k].mode, m->scan_id, time_hi, wl[1], pc, ctl, cl, subs[k].rgb);
	}
	{
		ProfScope ps(m, "k_finish_leaf");
		hipLaunchKernelGGL(k_finish_leaf, gridFor(total, 256, 1024), dim3(256), 0, m->cs, m->t, m->g, wl[1], wl[0], m->scan_id, pc, ctl);
	}
	auto bound = [&](u32 l) {
		u64 b = 0;
		for (const Sub& sb : subs) {
			u64 x = levelBound(sb.nbA, l - 1);
			if (sb.nbB) x += levelBound(sb.nbB, l - 1);
			b += std::min<u64>(sb.n, x);
		}
		return b;
	};
	propagateLevels(m, 2, bound, pc, 0);
	HIP_TRY(hipGetLastError());
	m->pending = true;
	m->bound = m->scan_new_bound;
	if (m->opt_async_apply && !m->chg_enabled && !sync) {
		// the caller keeps the lists alive until the next call on this map has joined the update
		m->done_by_flag = false;
		HIP_TRY(hipEventRecord(m->done_ev, m->stream));
		return UFOMAP_OK;
	}
	HIP_TRY(hipStreamSynchronize(m->stream));
	return finishPending(m);
}

#include "host_multi_gpu.inl"
#include "host_serialise.inl"
int ufomap_map_set_occupied_free_thres(ufomap_map* m, double occupied_thres, double free_thres)
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	// setOccupiedFreeThres (occupancy_map_base.h:746-759): write, change the thresholds, read back
	long long usize = 0;
	const size_t n = ufomap_map_write_ex(m, nullptr, nullptr, 0, 0, 1, 0, 0, nullptr, 0, &usize);
	if (n == (size_t)-1) return UFOMAP_ERR_DEVICE;
	std::vector<uint8_t> data(n);
	if ((size_t)-1 == ufomap_map_write_ex(m, nullptr, nullptr, 0, 0, 1, 0, 0, data.data(), data.size(), &usize)) return UFOMAP_ERR_DEVICE;
	m->model_log[0] = std::log(occupied_thres / (1.0 - occupied_thres));
	m->model_log[1] = std::log(free_thres / (1.0 - free_thres));
	applyModel(m);
	return ufomap_map_read_data(m, data.data(), data.size(), nullptr, nullptr, m->g.res, m->g.L, (int)usize, 0);
}

size_t ufomap_map_write(ufomap_map* m, uint8_t* buf, size_t cap)
{
	return ufomap_map_write_ex(m, nullptr, nullptr, 0, 0, 1, 0, 1, buf, cap, nullptr);
}

int ufomap_map_set_option(ufomap_map* m, const char* key, long long value)
{
	if (!m || !key) return fail(UFOMAP_ERR_INVALID, "null argument");
	if (0 == strcmp(key, "dda_mode")) {
		if (value < -1 || value > 2) return fail(UFOMAP_ERR_INVALID, "dda_mode: -1 auto, 0 LDS grid, 1 LDS filter, 2 direct");
		m->opt_dda_mode = (int)value;
	} else if (0 == strcmp(key, "dda_seg")) {
		m->opt_dda_seg = value ? 1 : 0;
	} else if (0 == strcmp(key, "early_map")) {
		m->opt_early = value ? 1 : 0;
	} else if (0 == strcmp(key, "spec")) {
		m->opt_spec = value ? 1 : 0;
		if (!m->opt_spec) m->spec_valid = false;
	} else if (0 == strcmp(key, "fast")) {
		m->opt_fast = (int)value;
	} else if (0 == strcmp(key, "async_apply")) {
		m->opt_async_apply = value ? 1 : 0;
	} else if (0 == strcmp(key, "gates")) {
		m->opt_gates = value ? 1 : 0;
	} else if (0 == strcmp(key, "batch_max")) {
		m->opt_batch_max = (int)std::max<long long>(1, std::min<long long>(value, (long long)UFO_BATCH_MAX));
	} else if (0 == strcmp(key, "tstamps")) {
		const int wrc = ufomap_map_wait(m);
		if (wrc) return wrc;
		unsigned long long* ts = nullptr;
		if (value) {
			HIP_TRY(m->b_ts.reserve((size_t)UFO_TS_SCANS * 64));
			HIP_TRY(hipMemset(m->b_ts.p, 0, (size_t)UFO_TS_SCANS * 64));
			HIP_TRY(hipDeviceSynchronize());
			ts = m->b_ts.as<unsigned long long>();
		} else {
			m->b_ts.release();
		}
		HIP_TRY(hipMemcpy(reinterpret_cast<char*>(m->b_pipe.p) + offsetof(Pipe, ts), &ts, sizeof(ts), hipMemcpyHostToDevice));
	} else if (0 == strcmp(key, "ser_short")) {
		m->opt_ser_short = value ? 1 : 0;
	} else if (0 == strcmp(key, "big")) {
		m->opt_big = value ? 1 : 0;
	} else if (0 == strcmp(key, "fast_color")) {
		m->opt_fast_color = value ? 1 : 0;
	} else if (0 == strcmp(key, "solo")) {
		m->opt_solo = value ? 1 : 0;
	} else if (0 == strcmp(key, "cast_threads")) {
		m->opt_cast_threads = (int)value;
	} else if (0 == strcmp(key, "cast_batch")) {
		m->opt_cast_batch = (int)value;
	} else if (0 == strcmp(key, "cast_qcap")) {
		m->opt_cast_qcap = (int)value;
	} else if (0 == strcmp(key, "cast_prio")) {
		m->opt_cast_prio = (int)std::max<long long>(0, std::min<long long>(3, value));
	} else if (0 == strcmp(key, "lazy_done")) {
		m->opt_lazy_done = value ? 1 : 0;
	} else if (0 == strcmp(key, "stage_pieces")) {
		m->opt_stage_pieces = (int)std::max<long long>(0, std::min<long long>(value, 16));
	} else if (0 == strcmp(key, "batch_depth")) {
		m->opt_batch_depth = (int)std::max<long long>(1, std::min<long long>(value, kAlt - 2));
	} else if (0 == strcmp(key, "hold")) {
		m->opt_hold = (int)std::max<long long>(0, std::min<long long>(value, kAlt));
	} else if (0 == strcmp(key, "gate_us")) {
		m->opt_gate_us = (int)std::max<long long>(100, std::min<long long>(value, 10000000));
	} else if (0 == strcmp(key, "sparse_set")) {
		m->opt_sparse_set = value ? 1 : 0;
	} else if (0 == strcmp(key, "phase_limit")) {
		m->phase_limit = (u32)std::max<long long>(4, value);
	} else if (0 == strcmp(key, "scan_id")) {
		(void)ufomap_map_wait(m);
		m->scan_id = (u32)value;  // (tests: start near the limit)
	} else if (0 == strcmp(key, "cast_global")) {
		m->opt_cast_global = (int)value;
	} else if (0 == strcmp(key, "cast")) {
		m->opt_cast = value ? 1 : 0;
	} else if (0 == strcmp(key, "cast_wgs")) {
		m->opt_cast_wgs = (int)value;
	} else if (0 == strcmp(key, "cast_k")) {
		m->opt_cast_k = (int)value;
	} else if (0 == strcmp(key, "dda_bits")) {
		m->opt_bits = value ? 1 : 0;
	} else if (0 == strcmp(key, "dda_lanes")) {
		m->opt_dda_lanes = (int)value;
	} else if (0 == strcmp(key, "vol")) {
		m->opt_vol = (int)std::max<long long>(0, std::min<long long>(2, value));
	} else if (0 == strcmp(key, "vol_pregrow")) {
		m->opt_vol_pregrow = value ? 1 : 0;
	} else if (0 == strcmp(key, "gather_stream")) {
		m->opt_gather_stream = value ? 1 : 0;
	} else if (0 == strcmp(key, "fast_simple")) {
		m->opt_fast_simple = value ? 1 : 0;
	} else if (0 == strcmp(key, "fail_scan")) {
		m->opt_fail_scan = value ? 1 : 0;
	} else if (0 == strcmp(key, "vol_mode")) {
		m->opt_vol_mode = (int)(value & 31);
	} else if (0 == strcmp(key, "vol_seg")) {
		m->opt_vol_seg = (int)std::max<long long>(5, std::min<long long>(8192, value));  // (a segment has < 4 K cells; k_vwalk gives up at 65 536)
	} else if (0 == strcmp(key, "vol_walk_blocks")) {
		m->opt_vol_walk_blocks = (int)std::max<long long>(1, std::min<long long>(65536, value));
	} else if (0 == strcmp(key, "vol_walk_lds")) {
		m->opt_vol_walk_lds = (int)std::max<long long>(0, std::min<long long>(128 << 10, value));
	} else if (0 == strcmp(key, "cast2_k")) {
		m->opt_cast2_k = (int)std::max<long long>(8, std::min<long long>(1024, value));
	} else if (0 == strcmp(key, "cast_oct")) {
		m->opt_cast_oct = value ? 1 : 0;
	} else if (0 == strcmp(key, "cast_oct_lds")) {
		m->opt_cast_oct_lds = (int)std::max<long long>(16, std::min<long long>(159, value));
	} else if (0 == strcmp(key, "es_sparse")) {
		m->opt_es_sparse = (int)std::max<long long>(0, std::min<long long>(2, value));
	} else if (0 == strcmp(key, "stage_thread")) {
		m->opt_stage_thread = value ? 1 : 0;
	} else if (0 == strcmp(key, "ctl_dbg")) {
		m->opt_ctl_dbg = value ? 1 : 0;
	} else if (0 == strcmp(key, "cast_fused")) {
		m->opt_cast_fused = (int)std::max<long long>(0, std::min<long long>(2, value));
	} else if (0 == strcmp(key, "vol_async")) {
		m->opt_vol_async = value ? 1 : 0;
	} else if (0 == strcmp(key, "merge_phases")) {
		m->opt_merge = value ? 1 : 0;
	} else if (0 == strcmp(key, "entry_guess")) {
		m->opt_entry_guess = value > 0 ? (u64)value : 0;
	} else {
		return fail(UFOMAP_ERR_INVALID, std::string("unknown option ") + key);
	}
	return UFOMAP_OK;
}

int ufomap_map_debug(ufomap_map* m, uint64_t* out, int n)
{
	if (!m) return fail(UFOMAP_ERR_INVALID, "null map");
	int rc = ufomap_map_wait(m);
	for (int i = 0; i < n && i < 64; ++i) out[i] = m->h_ctl->dbg[i];
	if (n > 62) out[62] = m->n_spec;       // scans enqueued on a predicted grid
	if (n > 63) out[63] = m->n_spec_redo;  // ... of which had to be repeated
	if (n > 61) out[61] = m->n_fast;       // scans enqueued on the fast path (fast_kernels.h)
	if (n > 60) out[60] = m->n_walks;      // ... walks of the tree that applied them (one walk takes every scan that has queued up)
	if (n > 59) out[59] = m->n_walk_scans; // ... scans in those walks
	if (n > 58) out[58] = m->n_gate_timeouts;  // stream hand-overs that timed out (the handle uses events from then on)
	if (n > 51) out[51] = m->n_phase_resets;  // phaseGuard
	if (n > 50) out[50] = m->n_vol;            // scans on the volume path (vol_kernels.h)
	if (n > 49) out[49] = m->n_vol_grow;       // ... times the node table was exchanged in the middle of such a scan's tree update
	if (n > 46) out[46] = m->n_fill_zero;     // (must be 0: every scan of a walk reports the table's fill)
	if (n > 47) out[47] = m->es_rounds;        // rounds the last scan with early_stopping > 0 took to settle
	if (n > 48) out[48] = m->n_vol_fallback;   // ... scans that turned to the general path (a ray clipped at the map cube)
	for (int k = 0; k < 4 && 52 + k < n; ++k) out[52 + k] = m->host_ns[k];  // host time inside doInsert (ns): scan enqueue, map enqueue, join, total
	return rc;
}

// device allocations of this process so far: hipMalloc calls, hipFree calls, bytes asked for, host nanoseconds inside them,
// re-hashes of a node table
void ufomap_alloc_counters(uint64_t out[5])
{
	out[0] = g_n_malloc.load();
	out[1] = g_n_free.load();
	out[2] = g_bytes_malloc.load();
	out[3] = g_ns_alloc.load();
	out[4] = g_n_rehash.load();
}

void* ufomap_map_stream(ufomap_map* m) { return m ? (void*)m->stream : nullptr; }

int ufomap_map_timeline(ufomap_map* m, unsigned long long* out, size_t n_words, unsigned long long* newest)
{
	if (!m || !out) return fail(UFOMAP_ERR_INVALID, "null argument");
	const int wrc = ufomap_map_wait(m);
	if (wrc) return wrc;
	if (!m->b_ts.p) return fail(UFOMAP_ERR_INVALID, "option tstamps is off");
	HIP_TRY(hipMemcpy(out, m->b_ts.p, std::min<size_t>(n_words, (size_t)UFO_TS_SCANS * 8) * 8, hipMemcpyDeviceToHost));
	if (newest) *newest = m->n_fseq;
	return UFOMAP_OK;
}

int ufomap_dev_expf(const float* x, float* out, size_t n, int device)
{
	if ((n && (!x || !out)) || n > (1ull << 31)) return fail(UFOMAP_ERR_INVALID, "null argument / more than 2^31 values");
	HIP_TRY(hipSetDevice(device));
	DevBuf b;
	HIP_TRY(b.reserve(std::max<size_t>(n, 1) * 4));
	HIP_TRY(hipMemcpy(b.p, x, n * 4, hipMemcpyHostToDevice));
	hipLaunchKernelGGL(k_dev_expf, gridFor(n), dim3(256), 0, nullptr, b.as<float>(), (u32)n);
	HIP_TRY(hipDeviceSynchronize());
	HIP_TRY(hipMemcpy(out, b.p, n * 4, hipMemcpyDeviceToHost));
	return UFOMAP_OK;
}

}  // extern "C"
