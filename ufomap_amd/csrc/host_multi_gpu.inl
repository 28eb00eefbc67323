// ufomap_comm_* / ufomap_map_insert_batch and what they need (RCCL loaded at run time, the list and bit-grid forms of a batch
// step). Included by ufomap_hip.hip inside its extern "C" block.
// ------------------------------------------------------------------------------------------------------------------
// Batched multi-sensor integration across GPUs behind the C ABI (BASELINE config C4): one process per
// GPU, every rank ray-casts ITS scan into an update list, ONE RCCL all-gather of fixed-size slots (header + list)
// moves all lists to all ranks, every rank applies the N lists in rank order with one walk of its replica's tree
// (ufomap_map_apply_keys_batch) -- the same map on every rank as the reference integrating the N scans one after the
// other. RCCL is loaded at run time (an already loaded copy is preferred, e.g. the one torch brought): the library has
// no link-time dependency on it, and a single-GPU host never touches it.
// ------------------------------------------------------------------------------------------------------------------
extern "C++" {
struct IdBytes {
	char b[128];  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES)
};
namespace
{
struct Rccl {
	void* lib = nullptr;
	int (*GetUniqueId)(void*) = nullptr;
	int (*CommInitRank)(void**, int, /* ncclUniqueId by value */ IdBytes, int) = nullptr;
	int (*CommDestroy)(void*) = nullptr;
	int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
	const char* (*GetErrorString)(int) = nullptr;
};
Rccl* rccl()
{
	// (initialised once, thread-safely: a function-local static)
	static Rccl r = [] {
		Rccl x;
		const char* env = getenv("UFOMAP_RCCL_LIB");
		void* h = nullptr;
		if (env && *env) {
			// the host names the library (tests: a stand-in that runs the collective through shared memory): that one and no other
			h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
		} else {
			const char* names[] = {"librccl.so.1", "librccl.so"};
			for (const char* name : names) {  // a copy that is already in the process first
				h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
				if (h) break;
			}
			for (const char* name : names) {
				if (h) break;
				h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
			}
		}
		if (!h) return x;
		x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
		x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
		x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
		x.AllGather = reinterpret_cast<decltype(x.AllGather)>(dlsym(h, "ncclAllGather"));
		x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
		if (x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllGather) x.lib = h;
		return x;
	}();
	return r.lib ? &r : nullptr;
}
int rcclFail(int code, const char* what)
{
	Rccl* r = rccl();
	return fail(UFOMAP_ERR_DEVICE, std::string(what) + ": " + ((r && r->GetErrorString) ? r->GetErrorString(code) : "RCCL error ") + " (" +
	                                   std::to_string(code) + ")");
}
constexpr size_t kSlotHeader = 128;  // ufomap_keys_info (40 bytes) + status and ray-cell box of the rank's scan (HdrTail), padded: travels in front of the list
}  // namespace
}  // extern "C++"

struct ufomap_comm {
	void* comm = nullptr;  // ncclComm_t
	bool own = false;
	int world = 1, rank = 0, device = 0;
	size_t cap = 1u << 20;  // bytes per slot; all ranks hold the same value (it only grows, by a rule all ranks apply alike)
	DevBuf send, recv[2];
	int flip = 0;
	uint8_t* h_hdr = nullptr;  // pinned: world headers
	uint64_t n_regrow = 0;
	// the ranks' common ray grid for the fast-path form of a step: derived from gathered data only, hence equal on all ranks
	Grid spec_grid{};
	bool spec_valid = false;
	uint64_t n_fast_steps = 0, n_redo_steps = 0;
};

int ufomap_comm_unique_id(uint8_t id[UFOMAP_COMM_ID_BYTES])
{
	if (!id) return fail(UFOMAP_ERR_INVALID, "null argument");
	Rccl* r = rccl();
	if (!r) return fail(UFOMAP_ERR_UNSUPPORTED, "librccl not found (set UFOMAP_RCCL_LIB)");
	static_assert(UFOMAP_COMM_ID_BYTES == sizeof(IdBytes), "ncclUniqueId is 128 bytes");
	const int e = r->GetUniqueId(id);
	return e ? rcclFail(e, "ncclGetUniqueId") : UFOMAP_OK;
}

static ufomap_comm* commAlloc(int world, int rank, int device)
{
	if (world < 1 || world > 128 || rank < 0 || rank >= world) {
		(void)fail(UFOMAP_ERR_INVALID, "comm: need 1 <= world <= 128 and 0 <= rank < world");
		return nullptr;
	}
	if (hipSetDevice(device) != hipSuccess) {
		(void)fail(UFOMAP_ERR_DEVICE, "hipSetDevice");
		return nullptr;
	}
	ufomap_comm* c = new ufomap_comm;
	c->world = world;
	c->rank = rank;
	c->device = device;
	if (const char* e = getenv("UFOMAP_COMM_SLOT")) c->cap = std::max<size_t>(256, (size_t)atoll(e));  // (tests: a slot so small that it has to grow)
	if (hipHostMalloc((void**)&c->h_hdr, (size_t)world * kSlotHeader) != hipSuccess) {
		delete c;
		(void)fail(UFOMAP_ERR_DEVICE, "hipHostMalloc");
		return nullptr;
	}
	return c;
}

ufomap_comm* ufomap_comm_create(const uint8_t id[UFOMAP_COMM_ID_BYTES], int world, int rank, int device)
{
	Rccl* r = rccl();
	if (!r || !id) {
		(void)fail(UFOMAP_ERR_UNSUPPORTED, "librccl not found (set UFOMAP_RCCL_LIB)");
		return nullptr;
	}
	ufomap_comm* c = commAlloc(world, rank, device);
	if (!c) return nullptr;
	IdBytes ib;
	memcpy(ib.b, id, sizeof(ib.b));
	const int e = r->CommInitRank(&c->comm, world, ib, rank);
	if (e) {
		(void)rcclFail(e, "ncclCommInitRank");
		(void)hipHostFree(c->h_hdr);
		delete c;
		return nullptr;
	}
	c->own = true;
	return c;
}

ufomap_comm* ufomap_comm_from_nccl(void* nccl_comm, int world, int rank, int device)
{
	if (!nccl_comm || !rccl()) {
		(void)fail(UFOMAP_ERR_UNSUPPORTED, "no communicator / librccl not found");
		return nullptr;
	}
	ufomap_comm* c = commAlloc(world, rank, device);
	if (c) c->comm = nccl_comm;
	return c;
}

void ufomap_comm_destroy(ufomap_comm* c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	(void)hipDeviceSynchronize();
	if (c->own && c->comm) (void)rccl()->CommDestroy(c->comm);
	if (c->h_hdr) (void)hipHostFree(c->h_hdr);
	delete c;
}

int ufomap_comm_stats(const ufomap_comm* c, uint64_t out[4])
{
	if (!c || !out) return fail(UFOMAP_ERR_INVALID, "null argument");
	out[0] = (uint64_t)c->world;
	out[1] = (uint64_t)c->rank;
	out[2] = (uint64_t)c->cap;
	out[3] = c->n_regrow;
	return UFOMAP_OK;
}

extern "C++" {
namespace
{
constexpr size_t kResStride = (sizeof(ScanCtl) + 64 + 63) & ~(size_t)63;  // one pinned result block + the word behind it

// the W - 1 result blocks of the other ranks' scans of a batch step (the own scan reports to the set's h_res)
ScanCtl* otherResult(uint8_t* all, int w) { return reinterpret_cast<ScanCtl*>(all + (size_t)w * kResStride); }

// Update-list form of a batch step (colour maps, first steps, grids beyond LDS, and the collective repeat of a fast step
// that a rank's scan did not fit): this rank's scan -> update list (scan stream; never reads the map), ONE all-gather of
// fixed-size slots [64-byte header | list | padding], all ranks' lists in rank order through one walk of the tree
// (ufomap_map_apply_keys_batch). A rank whose scan FAILED still takes part in the collective -- with a status word in
// its header and an empty list -- and every rank returns that error: nobody is left waiting in the all-gather.
int listBatchStep(ufomap_map* m, ufomap_comm* c, const double origin[3], const double* d_xyz, const uint8_t* d_rgb, size_t n, double max_range,
                  int discrete, bool in_join, int simple = 0, unsigned early_stopping = 0, unsigned depth = 0)
{
	Rccl* r = rccl();
	const int W = c->world;
	ufomap_keys_info info;
	// (in_join: called while a step is being joined -- the current hand-over set is that step's, nothing else is joined or rotated)
	int scan_rc;
	if (in_join) {
		scan_rc = scanKeysCore(m, origin, d_xyz, m->g.color ? d_rgb : nullptr, n, max_range, depth, discrete, simple, &info, early_stopping);
	} else if (early_stopping) {
		// (ufomap_map_scan_keys_rgb's prologue, with the argument its signature does not have)
		scan_rc = (hipSetDevice(m->device) == hipSuccess && hipStreamSynchronize(m->sstream) == hipSuccess) ? UFOMAP_OK : fail(UFOMAP_ERR_DEVICE, "scan stream");
		if (!scan_rc) {
			(void)rotateSets(m);
			m->args = ScanArgs{};
			scan_rc = scanKeysCore(m, origin, d_xyz, m->g.color ? d_rgb : nullptr, n, max_range, depth, discrete, simple, &info, early_stopping);
		}
	} else {
		scan_rc = ufomap_map_scan_keys_rgb(m, origin, d_xyz, m->g.color ? d_rgb : nullptr, n, max_range, depth, discrete, simple, &info);
	}
	std::string scan_msg = scan_rc ? g_err : std::string();
	if (scan_rc) memset(&info, 0, sizeof(info));
	auto listBytes = [](const ufomap_keys_info& k) {  // records + colour section
		return ((size_t)k.n_hit + k.n_miss) * sizeof(Entry) + ((k.reserved & 2u) ? (size_t)k.n_hit * 32u : 0u);
	};
	const size_t my_bytes = listBytes(info);
	struct HdrTail {  // behind the 40 bytes of ufomap_keys_info in the header
		i32 status;    // 0, or the error code of this rank's scan (its list is empty then)
		i32 have_box;  // the scan cast rays: box = their cells' bounding box (cells at depth 0)
		i32 box[6];
		double chg[6];  // the scan's contribution to the min / max change box (OMB:1367; none: chg[0] > chg[3]) -- every replica's box grows by every rank's scan
	};
	static_assert(sizeof(ufomap_keys_info) + sizeof(HdrTail) <= kSlotHeader, "exchange header");
	std::vector<ufomap_keys_info> infos((size_t)W);
	std::vector<i32> boxes((size_t)W * 6, 0);
	std::vector<double> chgs((size_t)W * 6, 0.0);
	std::vector<char> have_box((size_t)W, 0);
	int first_status = 0, first_rank = -1;
	for (;;) {
		// header + list into this rank's slot, ONE all-gather of fixed-size slots, the W headers back to the host
		hipError_t he = c->send.reserve(c->cap);
		if (he == hipSuccess) he = c->recv[0].reserve(c->cap * (size_t)W);
		if (he == hipSuccess) he = c->recv[1].reserve(c->cap * (size_t)W);
		if (he != hipSuccess) return fail(UFOMAP_ERR_DEVICE, "exchange buffers: out of device memory");  // (before any rank's first collective on these buffers)
		uint8_t* send = c->send.as<uint8_t>();
		uint8_t* recv = c->recv[c->flip].as<uint8_t>();
		memset(c->h_hdr, 0, kSlotHeader);
		memcpy(c->h_hdr, &info, sizeof(info));
		{
			HdrTail t{};
			t.status = scan_rc;
			// (the box of ray cells predicts the ranks' common ray grid of depth-0 steps: cells at another insert depth do not)
			t.have_box = (!scan_rc && n && 0 == depth && m->h_ctl->n_rays && m->h_ctl->mb_min[0] <= m->h_ctl->mb_max[0]) ? 1 : 0;
			for (int a = 0; a < 3 && t.have_box; ++a) {
				t.box[a] = m->h_ctl->mb_min[a];
				t.box[3 + a] = m->h_ctl->mb_max[a];
			}
			t.chg[0] = 1.0;
			t.chg[3] = 0.0;
			if (!scan_rc && n && m->h_ctl->aabb_min[0] != ~0ull)
				for (int a = 0; a < 3; ++a) {
					t.chg[a] = decD(m->h_ctl->aabb_min[a]);
					t.chg[3 + a] = decD(m->h_ctl->aabb_max[a]);
				}
			memcpy(c->h_hdr + sizeof(info), &t, sizeof(t));
		}
		bool ok = hipMemcpyAsync(send, c->h_hdr, kSlotHeader, hipMemcpyHostToDevice, m->sstream) == hipSuccess;
		const bool fits = kSlotHeader + my_bytes <= c->cap;  // (if not, the header alone tells everybody how much room is needed)
		if (ok && fits && my_bytes) ok = hipMemcpyAsync(send + kSlotHeader, m->b_entries.p, my_bytes, hipMemcpyDeviceToDevice, m->sstream) == hipSuccess;
		// (the collectives of a communicator run in the order they are issued, on every rank alike: the gathers of earlier
		// bit-grid steps, issued on their own stream, have completed before this one goes out on the scan stream)
		if (m->gstream) HIP_TRY(hipStreamSynchronize(m->gstream));
		const int e = r->AllGather(send, recv, c->cap, /* ncclChar */ 0, c->comm, m->sstream);
		if (e) return rcclFail(e, "ncclAllGather");
		HIP_TRY(hipMemcpy2DAsync(c->h_hdr, kSlotHeader, recv, c->cap, kSlotHeader, (size_t)W, hipMemcpyDeviceToHost, m->sstream));
		HIP_TRY(hipStreamSynchronize(m->sstream));
		if (!ok) return fail(UFOMAP_ERR_DEVICE, "copy into the exchange slot failed");
		size_t need = 0;
		first_status = 0;
		first_rank = -1;
		for (int k = 0; k < W; ++k) {
			const uint8_t* h = c->h_hdr + (size_t)k * kSlotHeader;
			memcpy(&infos[(size_t)k], h, sizeof(ufomap_keys_info));
			HdrTail t;
			memcpy(&t, h + sizeof(ufomap_keys_info), sizeof(t));
			if (t.status && 0 == first_status) {
				first_status = t.status;
				first_rank = k;
			}
			have_box[(size_t)k] = t.have_box ? 1 : 0;
			for (int a = 0; a < 6; ++a) boxes[(size_t)k * 6 + a] = t.box[a];
			for (int a = 0; a < 6; ++a) chgs[(size_t)k * 6 + a] = t.chg[a];
			need = std::max(need, kSlotHeader + listBytes(infos[(size_t)k]));
		}
		if (need <= c->cap) break;
		// some rank's list did not fit: every rank sees that in the headers and grows to the same capacity; an update of
		// an earlier batch that still reads the old receive buffers finishes first
		if (in_join) HIP_TRY(hipStreamSynchronize(m->stream));
		else {
			const int wrc = ufomap_map_wait(m);
			if (wrc) return wrc;
		}
		while (c->cap < need) c->cap *= 2;
		++c->n_regrow;
	}
	if (first_status) {
		// (every rank takes this exit: the maps stay equal -- none has applied anything of the step)
		if (scan_rc) return fail(scan_rc, scan_msg);
		return fail(first_status, "ufomap_map_insert_batch: the scan of rank " + std::to_string(first_rank) + " failed; nothing of this step was applied");
	}
	// the ranks' common ray grid for the steps to come (every rank computes it from the same gathered boxes)
	{
		i32 mn[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, mx[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
		bool any = false;
		for (int k = 0; k < W; ++k) {
			if (!have_box[(size_t)k]) continue;
			any = true;
			for (int a = 0; a < 3; ++a) {
				mn[a] = std::min(mn[a], boxes[(size_t)k * 6 + a]);
				mx[a] = std::max(mx[a], boxes[(size_t)k * 6 + 3 + a]);
			}
		}
		const bool had = c->spec_valid;
		const Grid prev = c->spec_grid;
		c->spec_valid = any && gridFromBox(had, prev, mn, mx, &c->spec_grid);
	}
	// (the step is going to be applied: every rank's scan reaches the min / max change box of every replica, OMB:1367)
	if (m->minmax_enabled)
		for (int k = 0; k < W; ++k) {
			const double* q = &chgs[(size_t)k * 6];
			if (q[0] > q[3]) continue;
			for (int a = 0; a < 3; ++a) {
				m->min_change[a] = std::min(m->min_change[a], q[a]);
				m->max_change[a] = std::max(m->max_change[a], q[3 + a]);
			}
		}
	// the W lists in rank order, one walk of the tree; with option async_apply the call returns after enqueueing and
	// the next batch's scan overlaps it (two receive buffers, used alternately)
	std::vector<const void*> lists((size_t)W);
	uint8_t* recv = c->recv[c->flip].as<uint8_t>();
	for (int k = 0; k < W; ++k) lists[(size_t)k] = (infos[(size_t)k].n_hit + infos[(size_t)k].n_miss) ? recv + (size_t)k * c->cap + kSlotHeader : nullptr;
	c->flip ^= 1;
	if (depth > 0) {
		// Insert depth > 0 (occupancy_map_base.h:378-386, 1085-1120: a miss is applied to a whole subtree; the reference's recommended way
		// to run a fine map, ufomap_mapping/README.md:38-39): the lists of the ranks one by one, in rank order -- each its hits, then its
		// misses at the insert depth (ufomap_map_apply_keys), exactly what the reference does scan after scan. One walk for all ranks'
		// lists would have to order the first list's misses before the second list's hits, which one pass cannot.
		for (int k = 0; k < W; ++k) {
			if (!lists[(size_t)k]) continue;
			const int arc = ufomap_map_apply_keys(m, lists[(size_t)k], &infos[(size_t)k]);
			if (arc) return arc;
		}
		return UFOMAP_OK;
	}
	if (in_join) return applyKeysBatchCore(m, lists.data(), infos.data(), W, true);
	return ufomap_map_apply_keys_batch(m, lists.data(), infos.data(), W);
}

// The ranks' common ray grid after a fast-path step has been joined: from the boxes of all ranks' scans (the finished
// control blocks of the walk: every rank holds the same ones).
void predictCommonGrid(ufomap_map* m)
{
	ufomap_comm* c = m->comm;
	if (!c) return;
	i32 mn[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, mx[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
	bool any = false;
	for (int w = 0; w < m->batch_world; ++w) {
		const ScanCtl* rc = (w == c->rank) ? m->h_ctl : otherResult(m->h_res_all, w);
		if (0 == rc->n_rays || rc->mb_min[0] > rc->mb_max[0]) continue;
		any = true;
		for (int a = 0; a < 3; ++a) {
			mn[a] = std::min(mn[a], rc->mb_min[a]);
			mx[a] = std::max(mx[a], rc->mb_max[a]);
		}
	}
	// (... and the min / max change box from the OTHER ranks' scans: the own scan's is taken in by finishPending like any scan's)
	if (m->minmax_enabled)
		for (int w = 0; w < m->batch_world; ++w) {
			if (w == c->rank) continue;
			const ScanCtl* rc = otherResult(m->h_res_all, w);
			if (~0ull == rc->aabb_min[0]) continue;
			for (int a = 0; a < 3; ++a) {
				m->min_change[a] = std::min(m->min_change[a], decD(rc->aabb_min[a]));
				m->max_change[a] = std::max(m->max_change[a], decD(rc->aabb_max[a]));
			}
		}
	if (!any) return;  // (a step of empty clouds: the grid stays)
	const bool had = c->spec_valid;
	const Grid prev = c->spec_grid;
	c->spec_valid = gridFromBox(had, prev, mn, mx, &c->spec_grid);
}

// One step of ufomap_map_insert_batch on the fast path: this rank's scan on the ranks' common ray grid, merged to two bit
// grids on the scan stream, ONE all-gather of [control block | tile bitmap | ray cells | hit voxels] (~0.2 MB per rank), ONE
// walk of the tree for the scans of all ranks in rank order (k_tile / k_ftail over W scans) -- enqueued, not awaited: no
// host round trip inside the step; what the host needs to know (errors, boxes) it reads when the step is joined.
int fastBatchStep(ufomap_map* m, ufomap_comm* c, const double origin[3], const double* d_xyz, size_t n, double max_range, int discrete)
{
	Rccl* r = rccl();
	const int W = c->world;
	m->spec_grid = c->spec_grid;
	m->spec_valid = true;
	const FastGeo fg = makeFastGeo(c->spec_grid);
	const size_t G = (size_t)fg.gr.bytes, slot = (UFO_XSLOT_HDR + 2 * G + 255) & ~(size_t)255;
	m->batch_world = W;
	m->comm = c;
	HIP_TRY(m->b_xsend.reserve(slot));
	HIP_TRY(m->b_xrecv.reserve(slot * (size_t)W));
	{
		const size_t pc = m->b_bpipe.cap;
		HIP_TRY(m->b_bpipe.reserve(sizeof(Pipe)));
		if (pc != m->b_bpipe.cap) HIP_TRY(hipMemsetAsync(m->b_bpipe.p, 0, sizeof(Pipe), m->sstream));
	}
	if (m->h_res_all_world < W) {
		if (m->h_res_all) HIP_TRY(hipHostFree(m->h_res_all));
		m->h_res_all = nullptr;
		HIP_TRY(hipHostMalloc((void**)&m->h_res_all, kResStride * (size_t)W));
		m->h_res_all_world = W;
	}
	if (!m->xchg_ev) HIP_TRY(hipEventCreateWithFlags(&m->xchg_ev, hipEventDisableTiming));
	for (int w = 0; w < W; ++w) {  // (armed before anything of the step is enqueued)
		otherResult(m->h_res_all, w)->err = ERR_NOT_STORED;
		*reinterpret_cast<volatile unsigned long long*>(otherResult(m->h_res_all, w) + 1) = 0ull;
	}
	ScanCtl* ctl = m->b_ctl.as<ScanCtl>();
	uint8_t* send = m->b_xsend.as<uint8_t>();
	uint8_t* recv = m->b_xrecv.as<uint8_t>();
	{
		// (the set's tile bitmap: named by the walk's descriptors, which are built before the scan half is enqueued)
		const size_t ct = m->b_tilebits.cap;
		HIP_TRY(m->b_tilebits.reserve(UFO_FAST_MAX_TILES / 8));
		if (ct != m->b_tilebits.cap) m->first_dirty = true;
	}
	// the walk: the scans of ranks 0 .. W-1 in this order (W <= UFO_BATCH_MAX: ufomap_map_insert_batch_ex)
	DescPack pk{};
	for (int w = 0; w < W && w < (int)UFO_BATCH_MAX; ++w) {
		uint8_t* base = recv + (size_t)w * slot;
		ScanDesc& d = pk.d[w];
		const bool own = w == c->rank;
		// (the own scan's control block and tile bitmap are the set's: the walk leaves them in their start state)
		d.ctl = own ? ctl : reinterpret_cast<ScanCtl*>(base);
		d.tile_bits = own ? m->b_tilebits.as<u32>() : reinterpret_cast<u32*>(base + UFO_XSLOT_CTL);
		d.gridM = reinterpret_cast<u32*>(base + UFO_XSLOT_HDR);
		d.gridH = reinterpret_cast<u32*>(base + UFO_XSLOT_HDR + G);
		d.host_result = own ? m->h_res : otherResult(m->h_res_all, w);
		d.done_value = (unsigned long long)m->seq;
		d.fseq = (unsigned long long)w;
	}
	m->batch_pack = &pk;
	m->batch_B = (u32)std::min<int>(W, (int)UFO_BATCH_MAX);
	m->batch_send = send;
	int rc = UFOMAP_OK;
	bool failed_locally = false;
	if (n) {
		const auto t_scan = std::chrono::steady_clock::now();
		rc = fastScanPhase(m, origin, d_xyz, n, max_range, discrete, true);
		m->host_ns[0] += (u64)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_scan).count();
		if (m->opt_fail_scan) rc = fail(UFOMAP_ERR_DEVICE, "injected failure of the scan half (test aid)");
		failed_locally = 0 != rc;
	}
	if (0 == n || failed_locally) {
		// an empty cloud on this rank: an empty contribution (the collective is entered all the same)
		for (int k = 0; k < 8; ++k) m->counts[k] = 0;
		m->fgeo = fg;
		m->fast = true;
		m->fseq = 0;
		m->chain_ok = false;
		m->gridM = m->gridH = c->spec_grid;
		m->haveH = m->haveM = true;
		m->hit_grid = true;
		m->last_depth = 0;
		m->h_res->err = ERR_NOT_STORED;
		*reinterpret_cast<volatile unsigned long long*>(m->h_res + 1) = 0ull;
		m->done_by_flag = true;
		rc = flushDeferred(m);
		if (!m->ctl_init_done) {
			ScanCtl init;
			memset(&init, 0, sizeof(init));
			for (int a = 0; a < 3; ++a) {
				init.mb_min[a] = init.hb_min[a] = INT32_MAX;
				init.mb_max[a] = init.hb_max[a] = INT32_MIN;
				init.aabb_min[a] = ~0ull;
				init.aabb_max[a] = 0ull;
			}
			HIP_TRY(hipMemcpy(m->b_ctl_init.p, &init, sizeof(ScanCtl), hipMemcpyHostToDevice));
			m->ctl_init_done = true;
		}
		HIP_TRY(m->b_gridM.reserve(G));
		HIP_TRY(m->b_gridH.reserve(G));
		HIP_TRY(m->b_tilebits.reserve(UFO_FAST_MAX_TILES / 8));
		HIP_TRY(hipMemcpyAsync(ctl, m->b_ctl_init.p, sizeof(ScanCtl), hipMemcpyDeviceToDevice, m->sstream));
		HIP_TRY(hipMemsetAsync(m->b_gridM.p, 0, G, m->sstream));
		HIP_TRY(hipMemsetAsync(m->b_gridH.p, 0, G, m->sstream));
		HIP_TRY(hipMemsetAsync(m->b_tilebits.p, 0, UFO_FAST_MAX_TILES / 8, m->sstream));
		m->ctl_clean = false;
		m->first_dirty = true;
		if (failed_locally) {
			// The scan half of THIS rank failed (an allocation, a launch): the other ranks are on their way into the step's
			// all-gather and would wait there for ever. This rank enters it all the same, with an empty contribution whose
			// control block carries a flag: every rank's walk stands back and the step is repeated in list form when it is
			// joined -- where a failure that persists is reported by every rank, a passing one is not an error.
			const u32 e = ERR_SPEC;
			HIP_TRY(hipMemcpyAsync(&ctl->err, &e, 4, hipMemcpyHostToDevice, m->sstream));
			rc = UFOMAP_OK;
		}
	}
	m->batch_pack = nullptr;
	if (rc) return rc;  // (only a failure of the substitute contribution itself is left: fatal for the communicator)
	// (a scan that was enqueued has merged itself into the slot and left the walk's descriptors: k_fmerge_batch)
	const bool packed = 0 != n && !failed_locally;
	if (!packed) {
		const u32 n4 = (u32)(G >> 4);
		hipLaunchKernelGGL(k_pack_slot, dim3(64), dim3(256), 0, m->sstream, reinterpret_cast<uint4*>(send), reinterpret_cast<const uint4*>(ctl),
		                   m->b_tilebits.as<uint4>(), m->b_gridM.as<uint4>(), m->b_gridH.as<uint4>(), n4);
	}
	bool xgate = false;
	{
		// Option gather_stream: the collective on a stream of its own -- only the walk needs what it gathers, so the wire of step i
		// could overlap the scan half of step i + 1 (which queues behind it on the scan stream). Off by default: with one
		// more stream in play the runtime's stream -> hardware-queue mapping puts pipeline streams on one queue, and on
		// one GPU the step measured 0.12 ms against 0.08 (profiles/r04_ab_experiments.log); to be re-measured on a node whose
		// wire is real.
		hipStream_t gs = m->sstream;
		if (m->opt_gather_stream) {
			if (!m->gstream) {
				HIP_TRY(hipStreamCreateWithFlags(&m->gstream, hipStreamNonBlocking));
				HIP_TRY(hipEventCreateWithFlags(&m->pack_ev, hipEventDisableTiming));
			}
			HIP_TRY(hipEventRecord(m->pack_ev, m->sstream));
			HIP_TRY(hipStreamWaitEvent(m->gstream, m->pack_ev, 0));
			gs = m->gstream;
		}
		xgate = m->gates && gs == m->sstream;  // (a stream of its own may share a hardware queue with the map stream: a gate there could wait for ever)
		const int e = r->AllGather(send, recv, slot, /* ncclChar */ 0, c->comm, gs);
		if (e) return rcclFail(e, "ncclAllGather");
		// the walk waits for the gathered slots: a one-thread kernel behind the collective stores the step's number, a one-wave kernel in
		// front of the walk waits for it (k_signal / k_gate as everywhere in the steady state: 2-3 us per hand-over against 9-15 for an
		// event pair, and 4 us less on the host) -- events when kernels may be serialised across streams (useGates)
		if (xgate) {
			if (!m->b_sig_xchg.p) {
				// (zero BEFORE any gate can look at it: the gate of this very step is on another stream, and recycled device memory is
				// not zero -- a gate that found a large number there opened before the all-gather had delivered: scripts/dev/fuzz_api.py)
				HIP_TRY(m->b_sig_xchg.reserve(64));
				HIP_TRY(hipMemsetAsync(m->b_sig_xchg.p, 0, 64, gs));
				HIP_TRY(hipStreamSynchronize(gs));
			}
			hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, gs, m->b_sig_xchg.as<unsigned long long>(), (unsigned long long)m->seq, (unsigned long long*)nullptr,
			                   (unsigned long long*)nullptr, 0ull);
		} else {
			HIP_TRY(hipEventRecord(m->xchg_ev, gs));
		}
	}
	// ---- the walk: the scans of ranks 0 .. W-1 in this order (UFO_BATCH_MAX at a time) ----
	m->cs = m->stream;
	const Need bound = fastBound(m, fg.gr);
	{
		Need in_flight;
		for (int i = 0; i < kAlt; ++i)
			if (m->alt[i].pending && !(m->alt[i].fast && 0 == memcmp(m->alt[i].fgeo.gr.base, fg.gr.base, sizeof(fg.gr.base)) &&
			                           0 == memcmp(m->alt[i].fgeo.gr.nb, fg.gr.nb, sizeof(fg.gr.nb))))
				in_flight += m->alt[i].bound;
		if (!tableTakes(m, in_flight + bound)) {
			const int jrc = joinEnqueued(m);  // (deterministic: every rank's replica holds the same number of blocks)
			if (jrc < 0) return jrc;
			if (!tableTakes(m, bound)) {
				m->cs = m->stream;
				const int grc = growFor(m, bound);
				if (grc) return grc;
			}
		}
	}
	const u32* prev_stat = nullptr;
	{
		int pk = -1;
		for (int i = 0; i < kAlt; ++i) {
			const HandOver& o = m->alt[i];
			if (!o.pending || o.deferred || (o.done_by_flag && !o.has_slot)) continue;
			if (pk < 0 || o.seq > m->alt[pk].seq) pk = i;
		}
		if (pk >= 0) {
			const HandOver& o = m->alt[pk];
			prev_stat = !o.done_by_flag ? &o.b_ctl.as<ScanCtl>()->err
			            : o.batch_world ? &o.b_bpipe.as<Pipe>()->wstat[0] : &m->b_pipe.as<Pipe>()->wstat[o.fseq & (UFO_RING - 1u)];
		}
	}
	m->scan_new_bound = bound;
	{
		// (the tiles' hand-over records are told apart by the phase they were written in: a new array starts zeroed, as in enqueueSlot --
		// until round 6 a map whose FIRST steady-state update was a batch step walked over whatever the allocation held: scripts/dev/fuzz_api.py)
		const size_t want = (size_t)UFO_FAST_MAX_TILES * sizeof(TileRec);
		if (m->b_tilerec.cap < want) {
			HIP_TRY(hipStreamSynchronize(m->stream));  // (a walk in flight reads the old array)
			HIP_TRY(m->b_tilerec.reserve(want));
			HIP_TRY(hipMemsetAsync(m->b_tilerec.p, 0, m->b_tilerec.cap, m->stream));
		}
	}
	if (xgate) {
		// (a collective that has not delivered within 10 s has failed: the walk then stands back on this rank, which is reported when the
		// step is joined -- the communicator is beyond repair either way)
		hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, m->stream, m->b_sig_xchg.as<unsigned long long>(), (unsigned long long)m->seq, ctl, 1000000000ull,
		                   (unsigned long long*)nullptr, 0ull);
	} else {
		HIP_TRY(hipStreamWaitEvent(m->stream, m->xchg_ev, 0));
	}
	Pipe* bp = m->b_bpipe.as<Pipe>();
	const float miss = (float)m->g.miss_log;
	{
		const int B = std::min<int>((int)UFO_BATCH_MAX, W);
		m->scan_id += 1;
		if (!packed) hipLaunchKernelGGL(k_batch_descs, dim3(1), dim3(64), 0, m->stream, bp, pk, (u32)B);
		{
			ProfScope ps(m, "k_tile");
			const u32 tw = (m->opt_tile_waves >= 1 && m->opt_tile_waves <= 4) ? (u32)m->opt_tile_waves : 4u;
			hipLaunchKernelGGL(k_tile<false>, dim3((fg.ntiles + tw - 1) / tw), dim3(64u * tw), 0, m->stream, m->t, m->g, fg, bp, 0ull, m->b_tilerec.as<TileRec>(), m->g.hit,
			                   miss, m->scan_id, prev_stat, changeLog(m), TileVol{});
		}
		{
			ProfScope ps(m, "k_ftail");
			hipLaunchKernelGGL(k_ftail<false>, dim3(1), dim3(UFO_FTAIL_THREADS), 0, m->stream, m->t, m->g, fg, bp, 0ull, m->b_tilerec.as<TileRec>(), m->scan_id, prev_stat,
			                   m->b_ctl_init.as<ScanCtl>(), (u32*)nullptr, (fg.ntiles + 31u) / 32u);
		}
	}
	HIP_TRY(hipGetLastError());
	m->pending = true;
	m->deferred = false;
	m->has_slot = true;
	m->bound = bound;
	m->last_rgb = nullptr;
	++c->n_fast_steps;
	return UFOMAP_OK;
}

// A step of ufomap_map_insert_batch whose walk stood back (the scan of some rank did not fit the common ray grid) is
// repeated in update-list form by ALL ranks, here -- i.e. while the step is being joined, which every rank does at the same
// point of its sequence of calls. Nothing of the step has reached the map; the steps enqueued behind it have stood back
// too and are repeated by their own joins, in order.
int redoBatchStep(ufomap_map* m)
{
	const ScanArgs a = m->args;
	ufomap_comm* c = m->comm;
	m->batch_world = 0;
	m->args.spec = false;
	m->chain_ok = false;
	m->first_dirty = true;
	++m->n_spec_redo;
	if (!c) return fail(UFOMAP_ERR_INVALID, "batch step without a communicator");
	++c->n_redo_steps;  // (the list form extends the ranks' common grid by the boxes it gathers)
	HIP_TRY(hipStreamSynchronize(m->stream));
	// (the step may be repeated from inside ANOTHER call's join -- a ufomap_map_insert_pointcloud2 whose record layout is in m->ing at this
	// moment: the step's own cloud is plain float64 points. Round 6: the repeat read them through that layout, scripts/dev/fuzz_api.py.)
	const Ingest sing = m->ing;
	m->ing = Ingest{};
	const int rc = listBatchStep(m, c, a.origin, a.d_xyz, nullptr, a.n, a.max_range, a.discrete, true);
	m->ing = sing;
	return rc;
}
}  // namespace
}  // extern "C++"

int ufomap_comm_counters(const ufomap_comm* c, uint64_t out[4])
{
	if (!c || !out) return fail(UFOMAP_ERR_INVALID, "null argument");
	out[0] = c->n_fast_steps;
	out[1] = c->n_redo_steps;
	out[2] = c->spec_valid ? 1u : 0u;
	out[3] = 0;
	return UFOMAP_OK;
}

int ufomap_map_insert_batch(ufomap_map* m, ufomap_comm* c, const double sensor_origin[3], const double* d_xyz, const uint8_t* d_rgb, size_t n,
                            double max_range, unsigned depth, int discrete)
{
	return ufomap_map_insert_batch_ex(m, c, sensor_origin, d_xyz, d_rgb, n, max_range, depth, discrete, 0, 0);
}

int ufomap_map_insert_batch_ex(ufomap_map* m, ufomap_comm* c, const double sensor_origin[3], const double* d_xyz, const uint8_t* d_rgb, size_t n,
                               double max_range, unsigned depth, int discrete, int simple_ray_casting, unsigned early_stopping)
{
	if (!m || !c || !sensor_origin) return fail(UFOMAP_ERR_INVALID, "null argument");
	// (decided from the map's configuration and the call's mode, the same on every rank by contract -- never from this rank's cloud:
	// a rank that returned here while a rank with an empty cloud entered the all-gather would leave that one waiting, ADVICE r5.
	// A coloured cloud into a plain map is integrated without its colours, as ufomap_map_insert_batch always did.)
	if (m->g.color && !discrete)
		return fail(UFOMAP_ERR_UNSUPPORTED, "OccupancyMapColor::insertPointCloud<PointCloudColor> does not compile in the reference (SURVEY.md 4)");
	if (!m->g.color) d_rgb = nullptr;
	if (m->g.color && !d_rgb && n) return fail(UFOMAP_ERR_INVALID, "a colour map needs the points' colours");
	if (depth >= m->g.L) return fail(UFOMAP_ERR_INVALID, "depth must be < depth_levels");
	if (0 != depth && m->g.color) return fail(UFOMAP_ERR_UNSUPPORTED, "insert_batch: update lists with colour are built at insert depth 0 only");
	if (c->device != m->device) return fail(UFOMAP_ERR_INVALID, "the communicator was created on another device than the map");
	if (!rccl()) return fail(UFOMAP_ERR_UNSUPPORTED, "librccl not found (set UFOMAP_RCCL_LIB)");
	HIP_TRY(hipSetDevice(m->device));
	if (m->poisoned) return fail(UFOMAP_ERR_CAPACITY, "the map is inconsistent after a node table overflow: ufomap_map_clear it");
	// Which form the step takes is decided from what ALL ranks know alike: the common ray grid (derived from gathered boxes
	// only), the map's configuration (the same on every rank by contract), never from this rank's cloud.
	// (more ranks than one walk takes scans: the list form. Several walks per step would each look at their own chunk's flags
	// only, and a scan of a later chunk that does not fit the common grid would leave the first chunk applied -- on the ranks
	// of that chunk alone the join would then not repeat the step, and the others' collective repeat would hang: ADVICE r3)
	// (fixed-step casting and early stopping -- the reference's simple_ray_casting / early_stopping arguments, occupancy_map_base.h:
	// 340-344, the same on every rank by contract -- take the list form: the scan half of the general path casts that way)
	const bool fast = c->spec_valid && m->opt_fast && m->opt_spec && !m->g.color && !m->chg_enabled && m->g.L >= 5 && nullptr == m->ing.data &&
	                  !simple_ray_casting && 0 == early_stopping && 0 == depth && c->world <= (int)UFO_BATCH_MAX && fastEligible(m, c->spec_grid, 0, 0, nullptr, 1);
	if (!fast) {
		// (joins what is in flight where it has to: scan_keys / apply_keys_batch)
		if (0 != depth) {  // (bit-grid steps still in flight are joined first -- at the same point of every rank's sequence of calls)
			const int wrc = ufomap_map_wait(m);
			if (wrc) return wrc;
		}
		m->batch_world = 0;
		return listBatchStep(m, c, sensor_origin, d_xyz, d_rgb, n, max_range, discrete, false, simple_ray_casting, early_stopping, depth);
	}
	// Joins happen at fixed points of the sequence of calls -- the step `batch_depth` (3) before this one is joined here -- never "when it
	// happens to be complete": a step that has to be repeated is repeated by all ranks together (a collective).
	// (diagnostics, ufomap_map_debug 52..55: host time in the step's joins + wait for the cloud, scan half, rest of the step, total)
	const auto t_call = std::chrono::steady_clock::now();
	auto since = [&](std::chrono::steady_clock::time_point t) { return (u64)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t).count(); };
	int prc = rotateSets(m);
	while (countPendingAlts(m) > m->opt_batch_depth - 1) {
		const int jrc = joinOldestAlt(m);
		if (!prc) prc = jrc;
	}
	m->host_ns[2] += since(t_call);
	// (prc: an EARLIER step's failure, reported when this call returns -- after this rank has entered this step's collective:
	// the other ranks are on their way into it)
	if (!c->spec_valid) {  // (the join repeated a step through the list form and found no common grid after it)
		m->batch_world = 0;
		const int lrc = listBatchStep(m, c, sensor_origin, d_xyz, d_rgb, n, max_range, discrete, false, simple_ray_casting, early_stopping);
		return lrc ? lrc : prc;
	}
	m->seq = ++m->latest_seq;
	{
		ScanArgs& a = m->args;
		a = ScanArgs{};
		a.spec = true;
		for (int k = 0; k < 3; ++k) a.origin[k] = sensor_origin[k];
		a.d_xyz = d_xyz;
		a.n = n;
		a.max_range = max_range;
		a.discrete = discrete;
	}
	m->gates = useGates(m);
	const auto t_step = std::chrono::steady_clock::now();
	const int rc = fastBatchStep(m, c, sensor_origin, d_xyz, n, max_range, discrete);
	if (rc) return rc;
	m->host_ns[1] += since(t_step);
	if (n) {  // (the caller's cloud has been consumed when the call returns: k_fhits kept what a repeat of the step needs)
		const auto t_wait = std::chrono::steady_clock::now();
		const int wrc = awaitCloudConsumed(m);
		if (wrc) return wrc;
		m->host_ns[2] += since(t_wait);
	}
	m->host_ns[3] += since(t_call);
	if (m->opt_async_apply && !m->profiling) return prc;
	// not asynchronous: the step is joined here (by every rank)
	const int jrc = joinOlder(m);
	HIP_TRY(hipStreamSynchronize(m->stream));
	const int frc = finishPending(m);
	return frc ? frc : (jrc ? jrc : prc);
}
