// host side of the steady-state path (fast_kernels.h): ray-grid geometry and eligibility, the scan half (fastScanPhase), slots on
// the map stream (enqueueSlot, flushDeferred), grid prediction. Included by ufomap_hip.hip inside its anonymous namespace.
// ---- the fast path (fast_kernels.h): a depth-0 scan on the predicted ray grid, five + one launches ------------------
FastGeo makeFastGeo(const Grid& gr)
{
	FastGeo fg{};
	fg.gr = gr;
	fg.rowBits = gridRowBits(gr);
	fg.planeBits = fg.rowBits * 2u * (u32)gr.nb[1];
	fg.ncells = fg.planeBits * 2u * (u32)gr.nb[2];
	u64 nt = 1;
	for (int a = 0; a < 3; ++a) {
		fg.tbase[a] = gr.base[a] >> 3;  // arithmetic shift: floor
		const i32 last = (gr.base[a] + 2 * gr.nb[a] - 1) >> 3;
		fg.nt[a] = (u32)(last - fg.tbase[a] + 1);
		nt *= fg.nt[a];
	}
	fg.ntiles = (u32)std::min<u64>(nt, 0xFFFFFFFFull);
	fg.tl = 3;
	return fg;
}

// does the ray kernel of the steady-state path hold this grid in LDS (k_fcast)? Else the grid is a "big" one: its rays go
// through k_fselect / k_cast<2> (marks in HBM), its tree update through k_tile / k_up / k_ftail
bool gridFitsLds(const Grid& gr) { return 1 == gr.layout && gr.bytes + UFO_CAST_LDS_EXTRA <= (160u << 10) - 512u; }

// the level-4 cells of a tile grid as a tile grid of their own (what k_ftail works on after k_up)
FastGeo makeUpGeo(const FastGeo& fg)
{
	FastGeo u = fg;
	u64 nt = 1;
	for (int a = 0; a < 3; ++a) {
		u.tbase[a] = fg.tbase[a] >> 1;
		u.nt[a] = (u32)(((fg.tbase[a] + (i32)fg.nt[a] - 1) >> 1) - u.tbase[a] + 1);
		nt *= u.nt[a];
	}
	u.ntiles = (u32)std::min<u64>(nt, 0xFFFFFFFFull);
	u.tl = 4;
	return u;
}

// upper bound of the node blocks one scan inside grid gr can create: every level-1 block of the grid and all ancestors
Need fastBound(const ufomap_map* m, const Grid& gr)
{
	return needBound(m, (u64)gr.nb[0] * (u64)gr.nb[1] * (u64)gr.nb[2], gr.nb, 1);
}

// dense grids of the cells above the tiles (fast_kernels.h: UpperGeo); returns the total number of cells
u64 makeUpperGeo(const FastGeo& fg, u32 L, UpperGeo* ug)
{
	memset(ug, 0, sizeof(*ug));
	u64 off = 0;
	for (u32 l = fg.tl + 1u; l <= L; ++l) {
		const u32 sh = l - fg.tl;
		ug->off[l] = (u32)std::min<u64>(off, 0xFFFFFFFFull);
		u64 sz = 1;
		for (int a = 0; a < 3; ++a) {
			ug->lo[l][a] = fg.tbase[a] >> sh;
			ug->n[l][a] = (u32)(((fg.tbase[a] + (i32)fg.nt[a] - 1) >> sh) - ug->lo[l][a] + 1);
			sz *= ug->n[l][a];
		}
		off += sz;
	}
	for (u32 l = L + 1; l < 25; ++l) ug->off[l] = (u32)std::min<u64>(off, 0xFFFFFFFFull);
	return off;
}

bool fastEligible(const ufomap_map* m, const Grid& gr, unsigned depth, int simple, const uint8_t* d_rgb, size_t n, int discrete = 1)
{
	if (!m->opt_fast || 0 != depth || m->g.L < 5 || 0 == n || n > (1u << 29)) return false;
	if (simple && (!gridFitsLds(gr) || 0 == m->opt_fast_simple)) return false;  // (fixed-step casting: k_fcast_simple, grids in LDS only)
	// (a coloured cloud into a plain map, a coloured cloud in continuous mode: the general path reports them)
	if (d_rgb && (!m->g.color || !discrete)) return false;
	if (m->g.color && 0 == m->opt_fast_color) return false;
	if (1 != gr.layout) return false;
	const FastGeo fg = makeFastGeo(gr);
	UpperGeo ug;
	if (gridFitsLds(gr)) {
		if (fg.ntiles > UFO_FAST_MAX_TILES) return false;
		// node blocks above the tiles that the scan can touch: k_ftail finds them on dense per-level grids (the tile grid
		// coarsened level by level) and holds them in LDS -- their number is bounded by the number of cells
		return makeUpperGeo(fg, m->g.L, &ug) <= UFO_UPPER_MAX;
	}
	// a ray grid beyond LDS: level 4 goes through k_up, k_ftail starts above it
	if (!m->opt_big || m->g.L < 6 || fg.ntiles > UFO_BIG_MAX_TILES || (u64)fg.ncells * 4u > (1ull << 30)) return false;
	const FastGeo fu = makeUpGeo(fg);
	return fu.ntiles <= UFO_FAST_MAX_TILES && makeUpperGeo(fu, m->g.L, &ug) <= UFO_UPPER_MAX;
}

unsigned long long gateTicks(const ufomap_map* m) { return (unsigned long long)std::max(100, m->opt_gate_us) * 100ull; }  // wall_clock64: 100 MHz

// scan half on the scan stream: first-point array, rays, merged bit grid + tile bitmap
// The descriptor of the newest scan half, if the host has kept it back (fastScanPhase, lazy_done), is published now.
int publishScanDone(ufomap_map* m)
{
	if (!m->sd_pending) return UFOMAP_OK;
	hipLaunchKernelGGL(k_scan_done, dim3(1), dim3(1), 0, m->sstream, m->b_pipe.as<Pipe>(), m->sd_saved);
	m->sd_pending = false;
	HIP_TRY(hipGetLastError());
	return UFOMAP_OK;
}

int fastScanPhase(ufomap_map* m, const double origin[3], const double* d_xyz, size_t n, double max_range, int discrete, bool batch_step = false,
                  bool lazy_done = false, bool solo = false, bool uploaded = false, const uint8_t* d_rgb = nullptr, int simple = 0)
{
	HIP_TRY(hipSetDevice(m->device));
	for (int k = 0; k < 8; ++k) m->counts[k] = 0;
	m->counts[0] = n;
	m->last_depth = 0;
	m->vol = false;
	m->haveH = m->haveM = true;
	m->gridM = m->spec_grid;
	m->gridH = m->spec_grid;
	m->scan_id += 1;
	const FastGeo fg = makeFastGeo(m->spec_grid);
	m->fgeo = fg;
	m->fast = true;
	// Scans may share a walk if they follow one another on the map stream and use the same ray grid (fast_kernels.h: k_claim);
	// before a scan on a new grid, the scans that have no slot of their own yet get one
	m->solo = solo;
	if (batch_step || solo) {
		// (a step of ufomap_map_insert_batch: its walk is enqueued by the host for the scans of all ranks; no claims. Solo:
		// a synchronous call with nothing in flight -- the scan and its walk on the map stream, a Pipe of their own)
		const int frc = flushDeferred(m);
		if (frc) return frc;
		m->chain_ok = false;
		m->fseq = 0;
		if (solo) {
			if (uploaded) {  // (a host cloud is copied on the prep stream)
				HIP_TRY(hipEventRecord(m->prep_ev, m->pstream));
				HIP_TRY(hipStreamWaitEvent(m->stream, m->prep_ev, 0));
			}
			const size_t pc = m->b_bpipe.cap;
			HIP_TRY(m->b_bpipe.reserve(sizeof(Pipe)));
			if (pc != m->b_bpipe.cap) HIP_TRY(hipMemsetAsync(m->b_bpipe.p, 0, sizeof(Pipe), m->stream));
		}
	} else {
		if (!m->chain_ok || 0 != memcmp(m->chain_geo.gr.base, fg.gr.base, sizeof(fg.gr.base)) || 0 != memcmp(m->chain_geo.gr.nb, fg.gr.nb, sizeof(fg.gr.nb))) {
			const int frc = flushDeferred(m);
			if (frc) return frc;
			++m->geo_id;
		}
		m->fseq = ++m->n_fseq;
		m->chain_ok = true;
		m->chain_geo = fg;
	}
	// where the walk that takes this scan reports: armed BEFORE the scan half is enqueued -- an earlier slot may claim the
	// scan as soon as its scan half has finished, i.e. before this call has enqueued the scan's own slot
	m->h_res->err = ERR_NOT_STORED;
	*reinterpret_cast<volatile unsigned long long*>(m->h_res + 1) = 0ull;  // k_ftail's "done" word
	m->done_by_flag = true;
	const u32 N = (u32)n;
	const D3 sensor{origin[0], origin[1], origin[2]};
	(void)makeUpperGeo(fg, m->g.L, &m->ugeo);
	const size_t cf = m->b_first.cap, ct = m->b_tilebits.cap;  // (a re-allocation may well return the old address: compare sizes)
	HIP_TRY(m->b_first.reserve(((size_t)fg.gr.bytes * 8 + 127) / 128 * 128 * 4));  // (one entry per bit of the grid, whole 128-entry columns: k_fmerge)
	const bool big = !gridFitsLds(m->spec_grid);  // the ray grid lives in HBM: k_fselect + k_cast<2> instead of k_fcast
	HIP_TRY(m->b_tilebits.reserve((big ? UFO_BIG_MAX_TILES : UFO_FAST_MAX_TILES) / 8));
	if (cf != m->b_first.cap || ct != m->b_tilebits.cap) m->first_dirty = true;
	// k_fhits depends on nothing but the cloud: on the prep stream it overlaps the ray kernel of the scan before
	m->cs = solo ? m->stream : m->pstream;
	if (m->first_dirty || 2 == m->opt_fast) {  // (option fast = 2: never trust the self-cleaning, a debugging aid)
		HIP_TRY(hipMemsetAsync(m->b_first.p, 0xFF, m->b_first.cap, m->cs));
		HIP_TRY(hipMemsetAsync(m->b_tilebits.p, 0, m->b_tilebits.cap, m->cs));
		m->first_dirty = false;
	}
	HIP_TRY(m->b_gridM.reserve(fg.gr.bytes));
	HIP_TRY(m->b_gridH.reserve(fg.gr.bytes));  // hit voxels, the ray grid's layout: zeroed by k_fhits, marked by k_fcast, read by k_tile
	m->hit_grid = true;
	HIP_TRY(m->b_hit_code.reserve(((n + 255) / 256) * 256 * sizeof(PointRec)));  // (per-point records of the head loop: k_fhits -> k_fcast; whole 256-point stretches)
	// a cloud in the caller's device memory (or raw records) is kept as float64 points for a possible repeat of the scan; a
	// host cloud already lies in the set's own staging buffer
	double* keep = nullptr;
	if (m->ing.data || d_xyz != m->b_in_xyz.as<double>()) {
		HIP_TRY(m->b_keep.reserve(n * 24));
		keep = m->b_keep.as<double>();
		m->args.d_xyz = keep;
		m->args.ing = Ingest{};
	}
	// (colours: read by the tree update -- and by a repeat of the scan -- after the call has returned)
	uint8_t* keep_rgb = nullptr;
	const uint8_t* scan_rgb = d_rgb;
	if (d_rgb && d_rgb != m->b_in_rgb.as<uint8_t>()) {
		HIP_TRY(m->b_keep_rgb.reserve(n * 3));
		keep_rgb = m->b_keep_rgb.as<uint8_t>();
		m->args.d_rgb = keep_rgb;
		scan_rgb = keep_rgb;
	}
	const u32 color_variant = d_rgb ? 1u : 0u;  // (the head loop of OccupancyMapColor::insertPointCloudDiscrete, OMC.h:195-233)
	ScanCtl init;
	memset(&init, 0, sizeof(init));
	for (int a = 0; a < 3; ++a) {
		init.mb_min[a] = init.hb_min[a] = INT32_MAX;
		init.mb_max[a] = init.hb_max[a] = INT32_MIN;
		init.aabb_min[a] = ~0ull;
		init.aabb_max[a] = 0ull;
	}
	ScanCtl* ctl = m->b_ctl.as<ScanCtl>();
	if (!m->ctl_init_done) {
		HIP_TRY(hipMemcpy(m->b_ctl_init.p, &init, sizeof(ScanCtl), hipMemcpyHostToDevice));
		m->ctl_init_done = true;
	}
	if (!m->ctl_clean || 2 == m->opt_fast) {
		// (steady state: the tree update of the set's previous scan has left the block in this very state, k_ftail)
		*m->h_ctl = init;
		HIP_TRY(hipMemcpyAsync(ctl, m->h_ctl, sizeof(ScanCtl), hipMemcpyHostToDevice, m->cs));
	}
	m->ctl_clean = false;  // (until that scan's tree update has been joined and found clean)
	const dim3 gp((N + 255) / 256);
	HIP_TRY(m->b_part1.reserve((size_t)gp.x * sizeof(BoxPartial)));
	// Round 6: octant sub-boxes (k_fcast4) -- a workgroup of the ray kernel takes rays of ONE octant around the sensor's cell and keeps
	// only that sub-box of the grid in LDS; k_fhits leaves the points' records sorted by octant. Needs the sub-boxes to be small (a
	// sensor in a corner of its grid looks into one octant that is the whole grid: then the whole-grid kernel).
	OctGeo og{};
	u32 oct_wgs = 0, oct_threads = 0, oct_rcap = 0, oct_qcap = 0, oct_lcap = 0;
	size_t oct_lds = 0, oct_sp = 0, oct_tab = 0, oct_bcnt = 0;
	bool oct = !big && !simple && 0 != m->opt_cast_oct && makeOctGeo(m->g, fg, sensor, &og);
	if (oct) {
		oct_threads = m->opt_cast_threads >= 512 ? 512u : 256u;
		oct_rcap = oct_threads / 2u;
		oct_qcap = 2u * oct_threads;
		oct_lcap = 2u * oct_threads;
		const size_t ns2 = ((size_t)gp.x + 2u) & ~(size_t)1u;  // (per stretch of the cloud: where the octant's records start, how many lie before)
		oct_lds = (size_t)og.wmax4 * 16u + (size_t)oct_rcap * sizeof(RayConst) + (size_t)oct_qcap * sizeof(SegRec) + 4u * (size_t)oct_lcap + 8u * ns2 + 64u + 64u * 4u;
		oct = gp.x <= 4096u && oct_lds <= (size_t)std::max(16, m->opt_cast_oct_lds) * 1024u;
		// two workgroups per CU (three in four CUs when scans are pipelined: the kernels of the other two streams want CUs of their own)
		oct_wgs = m->opt_cast_wgs > 0 ? (u32)m->opt_cast_wgs : (u32)(lazy_done ? 3 * m->n_cus / 2 : 2 * m->n_cus);
		oct_wgs = std::max<u32>(8u, std::min<u32>(oct_wgs, std::max<u32>(8u, (N + 63u) / 64u)));
	}
	if (oct) {
		// slabs | per-workgroup steps / rays / hits | the table for k_fmerge | the stretches' counts -- one buffer of the set
		oct_sp = (size_t)oct_wgs * og.wmax4 * 16u;
		oct_tab = (oct_sp + (size_t)oct_wgs * 8u * 3u + 15u) & ~(size_t)15u;
		oct_bcnt = oct_tab + ((sizeof(OctTab) + 15u) & ~(size_t)15u);
		HIP_TRY(m->b_slabs.reserve(oct_bcnt + (size_t)gp.x * 48u));  // (per stretch: eight 16-bit counts, eight 32-bit weights)
	}
	{
		ProfScope ps(m, "k_fhits");
		uint4* const bc = oct ? reinterpret_cast<uint4*>(m->b_slabs.as<char>() + oct_bcnt) : nullptr;
		if (oct && discrete)
			hipLaunchKernelGGL((k_fhits<true, true>), gp, dim3(256), 0, m->cs, m->g, fg, sensor, d_xyz, N, max_range, color_variant, m->b_first.as<u32>(),
			                   m->b_part1.as<BoxPartial>(), ctl, m->ing, m->b_hit_code.as<PointRec>(), keep, d_rgb, keep_rgb, bc, reinterpret_cast<u32*>(bc + gp.x));
		else if (oct)
			hipLaunchKernelGGL((k_fhits<false, true>), gp, dim3(256), 0, m->cs, m->g, fg, sensor, d_xyz, N, max_range, 0u, m->b_first.as<u32>(),
			                   m->b_part1.as<BoxPartial>(), ctl, m->ing, m->b_hit_code.as<PointRec>(), keep, (const uint8_t*)nullptr, (uint8_t*)nullptr, bc, reinterpret_cast<u32*>(bc + gp.x));
		else if (discrete)
			hipLaunchKernelGGL((k_fhits<true, false>), gp, dim3(256), 0, m->cs, m->g, fg, sensor, d_xyz, N, max_range, color_variant, m->b_first.as<u32>(),
			                   m->b_part1.as<BoxPartial>(), ctl, m->ing, m->b_hit_code.as<PointRec>(), keep, d_rgb, keep_rgb, (uint4*)nullptr, (u32*)nullptr);
		else
			hipLaunchKernelGGL((k_fhits<false, false>), gp, dim3(256), 0, m->cs, m->g, fg, sensor, d_xyz, N, max_range, 0u, m->b_first.as<u32>(),
			                   m->b_part1.as<BoxPartial>(), ctl, m->ing, m->b_hit_code.as<PointRec>(), keep, (const uint8_t*)nullptr, (uint8_t*)nullptr, (uint4*)nullptr, (u32*)nullptr);
	}
	// stream-to-stream hand-overs of this path: k_signal / k_gate (fast_kernels.h), not events
	m->gates = useGates(m);
	if (solo) {
		// (one stream: nothing to hand over)
	} else if (m->gates) {
		hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, m->pstream, m->sig_prep, (unsigned long long)m->seq, m->h_prep, batch_step ? nullptr : m->b_ts.as<unsigned long long>(),
		                   (unsigned long long)m->fseq);
		if (m->sd_pending) {
			// (the scan half before this one ends and this one's gate opens in one launch)
			hipLaunchKernelGGL(k_done_gate, dim3(1), dim3(1), 0, m->sstream, m->b_pipe.as<Pipe>(), m->sd_saved, m->sig_prep, (unsigned long long)m->seq, ctl,
			                   gateTicks(m), (unsigned long long)m->fseq);
			m->sd_pending = false;
		} else {
			hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, m->sstream, m->sig_prep, (unsigned long long)m->seq, ctl, gateTicks(m),
			                   batch_step ? nullptr : m->b_ts.as<unsigned long long>(), (unsigned long long)m->fseq);
		}
	} else {
		const int prc = publishScanDone(m);
		if (prc) return prc;
		HIP_TRY(hipEventRecord(m->prep_ev, m->pstream));
		HIP_TRY(hipStreamWaitEvent(m->sstream, m->prep_ev, 0));
	}
	m->cs = solo ? m->stream : m->sstream;
	if (big) {
		// ---- a ray grid beyond LDS: the surviving rays are compacted (k_fselect) and walked by the ray kernel of the general
		// path (k_cast<2>: a workgroup takes consecutive stretches of the cloud and marks an LDS box of the grid, ORed into
		// the grid in HBM) -- no slabs; the walk derives the hit grid and the tile bitmap from the grid (k_fmerge) ----
		const u32 n_blk = gp.x;  // workgroups of k_fselect = 256-point stretches of the cloud
		HIP_TRY(m->b_ray_end.reserve((size_t)n_blk * 256u * sizeof(D3)));
		HIP_TRY(m->b_blk_range.reserve((size_t)n_blk * 8));
		u32 nwg = m->opt_cast_wgs > 0 ? (u32)m->opt_cast_wgs : std::min<u32>(n_blk, 1024u);
		nwg = std::max<u32>(std::max<u32>(1u, nwg), (n_blk + UFO_CAST_STRETCHES - 1u) / UFO_CAST_STRETCHES);
		HIP_TRY(m->b_slabs.reserve((size_t)std::max(nwg, n_blk) * 8 + (size_t)n_blk * 8));  // k_cast<2>'s step counts | k_fselect's per-stretch counts
		unsigned long long* parts = m->b_slabs.as<unsigned long long>() + std::max(nwg, n_blk);
		HIP_TRY(hipMemsetAsync(m->b_gridM.p, 0, fg.gr.bytes, m->cs));
		HIP_TRY(hipMemsetAsync(m->b_gridH.p, 0, fg.gr.bytes, m->cs));
		{
			ProfScope ps(m, "k_fselect");
			if (discrete)
				hipLaunchKernelGGL(k_fselect<true>, gp, dim3(256), 0, m->cs, N, m->b_first.as<u32>(), m->b_hit_code.as<PointRec>(), m->b_ray_end.as<D3>(),
				                   m->b_blk_range.as<u32>(), parts, m->b_gridH.as<u32>(), scan_rgb ? 0u : 1u, ctl);
			else
				hipLaunchKernelGGL(k_fselect<false>, gp, dim3(256), 0, m->cs, N, m->b_first.as<u32>(), m->b_hit_code.as<PointRec>(), m->b_ray_end.as<D3>(),
				                   m->b_blk_range.as<u32>(), parts, m->b_gridH.as<u32>(), scan_rgb ? 0u : 1u, ctl);
		}
		{
			ProfScope ps(m, "k_cast_global");
			const u32 grid_lds = ((160u << 10) - 1024u - (u32)UFO_CAST2_LDS_EXTRA) & ~15u;
			hipLaunchKernelGGL(k_cast<2>, dim3(nwg), dim3(512), (size_t)grid_lds + UFO_CAST2_LDS_EXTRA, m->cs, m->g, sensor, 0u, fg.gr, m->b_gridM.as<u32>(),
			                   m->b_ray_end.as<D3>(), (u32)std::max(8, m->opt_cast_k), ctl, ctl, m->b_slabs.as<unsigned long long>(), grid_lds,
			                   m->b_blk_range.as<u32>(), n_blk, grid_lds);
		}
		ScanDesc d{};
		d.slabs = nullptr;
		d.parts = parts;
		d.gridM = m->b_gridM.as<u32>();
		d.gridH = m->b_gridH.as<u32>();
		d.first = m->b_first.as<u32>();
		d.tile_bits = m->b_tilebits.as<u32>();
		d.ctl = ctl;
		d.host_result = m->h_res;
		d.boxes = m->b_part1.as<BoxPartial>();
		d.done_value = (unsigned long long)m->seq;
		d.fseq = (unsigned long long)m->fseq;
		d.n_slabs = 0;
		d.nboxes = gp.x;
		d.geo = m->geo_id;
		d.rgb = scan_rgb;
		if (solo) {
			DescPack pk{};
			pk.d[0] = d;
			hipLaunchKernelGGL(k_batch_descs, dim3(1), dim3(64), 0, m->cs, m->b_bpipe.as<Pipe>(), pk, 1u);
		} else if (lazy_done && m->gates) {
			m->sd_saved = d;
			m->sd_pending = true;
		} else {
			hipLaunchKernelGGL(k_scan_done, dim3(1), dim3(1), 0, m->sstream, m->b_pipe.as<Pipe>(), d);
		}
	} else {
		// One workgroup per CU is what the ray kernel's LDS allows, and alone it is fastest with one on every CU. In a row
		// of asynchronous scans it shares the chip with the first-point pass of the next scan and the tree update of the
		// scan before: with a workgroup on three CUs in four it does not wait for the last CUs those kernels hold, and they
		// have CUs where nothing else competes (measured, scripts/dev/dev_ab.py cast_wgs=...: 256 -> 0.052, 192 -> 0.046 ms/scan).
		u32 nwg = m->opt_cast_wgs > 0 ? (u32)m->opt_cast_wgs : (lazy_done ? (u32)(3 * m->n_cus / 4) : (u32)m->n_cus);
		nwg = std::max<u32>(1u, std::min<u32>(nwg, (N + 63u) / 64u));
		if (oct) nwg = oct_wgs;
		const u32 cap_wg = (N + nwg - 1) / nwg;
		if (!oct) HIP_TRY(m->b_slabs.reserve((size_t)nwg * fg.gr.bytes + (size_t)nwg * 8 * 3));  // slabs + per-workgroup steps / rays / hits (of this set: merged by the walk)
		unsigned long long* sp = reinterpret_cast<unsigned long long*>(m->b_slabs.as<char>() + (oct ? oct_sp : (size_t)nwg * fg.gr.bytes));
		// end of the scan half: the scan's descriptor and number become visible to the walks (k_claim)
		ScanDesc d{};
		d.slabs = m->b_slabs.as<uint4>();
		d.parts = sp;
		d.gridM = m->b_gridM.as<u32>();
		d.gridH = m->b_gridH.as<u32>();
		d.first = m->b_first.as<u32>();
		d.tile_bits = m->b_tilebits.as<u32>();
		d.ctl = ctl;
		d.host_result = m->h_res;
		d.boxes = m->b_part1.as<BoxPartial>();
		d.done_value = (unsigned long long)m->seq;
		d.fseq = (unsigned long long)m->fseq;
		d.n_slabs = nwg;
		d.nboxes = gp.x;
		d.geo = m->geo_id;
		d.rgb = scan_rgb;
		d.oct = oct ? reinterpret_cast<const OctTab*>(m->b_slabs.as<char>() + oct_tab) : nullptr;
		Pipe* const solo_pipe = solo ? m->b_bpipe.as<Pipe>() : nullptr;
		{
			ProfScope ps(m, "k_fcast");
			// LDS beside the bit grid: ray constants + segment queue. Sized for the rays a workgroup gets (a round of `batch`
			// rays; more rays = more rounds), not for the worst case: what the ray kernel leaves free on a CU is what the
			// kernels of the other two streams can run in beside it.
			u32 batch = (u32)std::min<long long>(512, std::max<long long>(64, m->opt_cast_batch));
			u32 qcap = (u32)std::min<long long>(2048, std::max<long long>(2 * batch, m->opt_cast_qcap));
			const u32 prio = (u32)m->opt_cast_prio;
			const u32 cthreads = m->opt_cast_threads >= 1024 ? 1024u : (m->opt_cast_threads >= 768 ? 768u : (m->opt_cast_threads >= 512 ? 512u : 256u));
			auto ldsFor = [&](u32 b, u32 q) { return (size_t)fg.gr.bytes + (size_t)b * (sizeof(RayConst) + sizeof(RayHdr)) + (size_t)q * sizeof(SegRec) + 256u; };
			if (ldsFor(batch, qcap) > (160u << 10) - 256u) {
				batch = UFO_CAST_BATCH;
				qcap = UFO_CAST_QCAP;
			}
			{
				const size_t lds = ldsFor(batch, qcap);
				static const bool trace = nullptr != getenv("UFOMAP_TRACE_GRID");
				if (trace)
					fprintf(stderr, "[ufomap] fast grid: %d x %d x %d blocks, %llu bytes; k_fcast: %u workgroups, %zu bytes of LDS each\n", fg.gr.nb[0], fg.gr.nb[1],
					        fg.gr.nb[2], (unsigned long long)fg.gr.bytes, nwg, lds);
			}
			if (simple) {
				// fixed-step casting (freeSpaceSimple): one lane per ray, the grid alone in LDS
				if (discrete)
					hipLaunchKernelGGL(k_fcast_simple<true>, dim3(nwg), dim3(cthreads), (size_t)fg.gr.bytes, m->cs, m->g, fg, sensor, N, m->b_first.as<u32>(), m->b_slabs.as<u32>(), ctl, ctl,
					                   sp, m->b_hit_code.as<PointRec>(), solo_pipe, d);
				else
					hipLaunchKernelGGL(k_fcast_simple<false>, dim3(nwg), dim3(cthreads), (size_t)fg.gr.bytes, m->cs, m->g, fg, sensor, N, m->b_first.as<u32>(), m->b_slabs.as<u32>(), ctl, ctl,
					                   sp, m->b_hit_code.as<PointRec>(), solo_pipe, d);
			} else if (oct) {
				OctTab* const tab = reinterpret_cast<OctTab*>(m->b_slabs.as<char>() + oct_tab);
				const uint4* const bc = reinterpret_cast<const uint4*>(m->b_slabs.as<char>() + oct_bcnt);
				const u32 Kc = (u32)std::max(8, m->opt_cast2_k);
				if (discrete)
					hipLaunchKernelGGL(k_fcast4<true>, dim3(nwg), dim3(oct_threads), oct_lds, m->cs, m->g, fg, sensor, og, (u32)gp.x, bc, reinterpret_cast<const u32*>(bc + gp.x), m->b_hit_code.as<PointRec>(), m->b_first.as<u32>(),
					                   m->b_slabs.as<uint4>(), tab, Kc, ctl, ctl, sp, oct_rcap, oct_qcap, oct_lcap, prio, solo_pipe, d);
				else
					hipLaunchKernelGGL(k_fcast4<false>, dim3(nwg), dim3(oct_threads), oct_lds, m->cs, m->g, fg, sensor, og, (u32)gp.x, bc, reinterpret_cast<const u32*>(bc + gp.x), m->b_hit_code.as<PointRec>(), m->b_first.as<u32>(),
					                   m->b_slabs.as<uint4>(), tab, Kc, ctl, ctl, sp, oct_rcap, oct_qcap, oct_lcap, prio, solo_pipe, d);
			} else if (m->opt_cast_fused >= 2) {
				// (round 6: the points that cast a ray are packed into a list before anything is set up; the list takes what the grid, the
				// ray constants and the segment queue leave of the CU's LDS, up to the workgroup's share of the cloud)
				const size_t fixed = (size_t)fg.gr.bytes + (size_t)batch * sizeof(RayConst) + (size_t)qcap * sizeof(SegRec) + 256u;
				const size_t room = ((160u << 10) - 512u - fixed) / 4u;
				const u32 lcap = (u32)std::max<size_t>(64u, std::min<size_t>(room, ((size_t)cap_wg + 63u) & ~(size_t)63u));
				const size_t lds3 = fixed + 4u * (size_t)lcap;
				if (discrete)
					hipLaunchKernelGGL(k_fcast3<true>, dim3(nwg), dim3(cthreads), lds3, m->cs, m->g, fg, sensor, N, m->b_first.as<u32>(), m->b_slabs.as<u32>(), (u32)std::max(8, m->opt_cast2_k),
					                   ctl, ctl, sp, m->b_hit_code.as<PointRec>(), batch, qcap, lcap, prio, solo_pipe, d);
				else
					hipLaunchKernelGGL(k_fcast3<false>, dim3(nwg), dim3(cthreads), lds3, m->cs, m->g, fg, sensor, N, m->b_first.as<u32>(), m->b_slabs.as<u32>(), (u32)std::max(8, m->opt_cast2_k),
					                   ctl, ctl, sp, m->b_hit_code.as<PointRec>(), batch, qcap, lcap, prio, solo_pipe, d);
			} else {
				// (round 5's form, kept as the cross-check of k_fcast3 -- option cast_fused = 1: head loop, set-up and cuts by the lane that looks at the point)
				const size_t lds2 = (size_t)fg.gr.bytes + (size_t)batch * sizeof(RayConst) + (size_t)qcap * sizeof(SegRec) + 256u;
				if (discrete)
					hipLaunchKernelGGL(k_fcast2<true>, dim3(nwg), dim3(cthreads), lds2, m->cs, m->g, fg, sensor, N, m->b_first.as<u32>(), m->b_slabs.as<u32>(), (u32)std::max(8, m->opt_cast2_k),
					                   ctl, ctl, sp, m->b_hit_code.as<PointRec>(), batch, qcap, prio, solo_pipe, d);
				else
					hipLaunchKernelGGL(k_fcast2<false>, dim3(nwg), dim3(cthreads), lds2, m->cs, m->g, fg, sensor, N, m->b_first.as<u32>(), m->b_slabs.as<u32>(), (u32)std::max(8, m->opt_cast2_k),
					                   ctl, ctl, sp, m->b_hit_code.as<PointRec>(), batch, qcap, prio, solo_pipe, d);
			}
		}
		if (solo) {
			// (k_fcast has written the descriptor itself)
		} else if (!batch_step) {
			// Asynchronous calls in a row: the descriptor is kept back and published by the next scan's gate kernel (k_done_gate)
			// -- or by whatever needs this scan's tree update first (flushDeferred) -- one launch less per scan on this stream.
			if (lazy_done && m->gates) {
				m->sd_saved = d;
				m->sd_pending = true;
			} else {
				hipLaunchKernelGGL(k_scan_done, dim3(1), dim3(1), 0, m->sstream, m->b_pipe.as<Pipe>(), d);
			}
		} else {
			// the other ranks get this scan as two bit grids, not as 256 slabs: merged here, on the scan stream -- straight into this
			// rank's exchange slot, with the slot's header and the descriptors of the step's walk (k_fmerge_batch)
			ScanDesc own = d;
			own.fseq = 0;
			own.gridM = reinterpret_cast<u32*>(m->batch_send + UFO_XSLOT_HDR);
			own.gridH = reinterpret_cast<u32*>(m->batch_send + UFO_XSLOT_HDR + (size_t)fg.gr.bytes);
			ProfScope ps(m, "k_fmerge");
			const u32 n4 = (u32)(fg.gr.bytes >> 4);
			hipLaunchKernelGGL(k_fmerge_batch, dim3(std::min<u32>((n4 + 63) / 64, 1024) + 1u), dim3(1024), 0, m->sstream, fg, m->b_bpipe.as<Pipe>(), own, n4,
			                   reinterpret_cast<unsigned long long*>(m->batch_send), *m->batch_pack, m->batch_B);
		}
	}
	HIP_TRY(hipGetLastError());
	++m->n_fast;
	return UFOMAP_OK;
}

// The call returns once the caller's device cloud has been consumed: k_fhits has run (fast path: the word k_signal stores
// in pinned memory, normally there long before the rest of the call has been enqueued), or the prep stream's event.
int awaitCloudConsumed(ufomap_map* m)
{
	if (m->gates) {
		volatile unsigned long long* hp = m->h_prep;
		const auto t0 = std::chrono::steady_clock::now();
		for (u32 spins = 0; *hp < (unsigned long long)m->seq; ++spins) {
			if (0 == (spins & 1023u) && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) {
				HIP_TRY(hipStreamSynchronize(m->pstream));
				break;
			}
		}
		std::atomic_thread_fence(std::memory_order_acquire);
		return UFOMAP_OK;
	}
	HIP_TRY(hipEventSynchronize(m->prep_ev));
	return UFOMAP_OK;
}

// A slot on the map stream for a fast-path scan: k_claim (waits for the scan's scan half, claims the scans before it that
// have no slot of their own and the scans behind it that are ready), k_fmerge, k_tile, k_ftail -- ONE walk of the tree for
// the whole run (fast_kernels.h). k < 0: the current set's scan; else the scan of m->alt[k].
int enqueueSlot(ufomap_map* m, int k)
{
	HandOver* const a = k < 0 ? nullptr : &m->alt[k];
	const FastGeo fg = a ? a->fgeo : m->fgeo;
	const uint64_t f = a ? a->fseq : m->fseq;
	ScanCtl* const ctl = (a ? a->b_ctl : m->b_ctl).as<ScanCtl>();
	const Need bound = fastBound(m, fg.gr);  // (every block of the grid new: no more, however many scans the walk takes)
	// The update enqueued just before this one, if it has not been joined: this walk looks at its status when it starts and
	// stands back if that one did (everything flagged is then repeated in order when it is joined).
	const u32* prev_stat = nullptr;
	Need in_flight;
	auto scanQueue = [&]() {
		prev_stat = nullptr;
		in_flight = Need{};
		int pk = -1;
		for (int i = 0; i < kAlt; ++i) {
			const HandOver& o = m->alt[i];
			if (!o.pending || o.deferred || i == k) continue;
			// (scans on this scan's own ray grid add nothing: `bound` is every block of that grid, whoever creates it)
			const bool same_grid = o.fast && 0 == memcmp(o.fgeo.gr.base, fg.gr.base, sizeof(fg.gr.base)) && 0 == memcmp(o.fgeo.gr.nb, fg.gr.nb, sizeof(fg.gr.nb));
			if (!same_grid) in_flight += o.bound;
			// (a fast-path scan without a slot of its own goes with a later slot -- possibly this one: its status word is not
			// written before this walk starts; the scan before it that has a slot is the predecessor to look at)
			if (o.done_by_flag && !o.has_slot) continue;
			if (pk < 0 || o.seq > m->alt[pk].seq) pk = i;
		}
		if (pk >= 0)
			// (a batch step's walk reports in the step's own Pipe -- round 6: this looked at the shared ring's word for a batch predecessor, so a
			// scan of ufomap_map_insert enqueued behind a batch step that stood back was applied BEFORE the step's repeat: scripts/dev/fuzz_api.py)
			prev_stat = !m->alt[pk].done_by_flag ? &m->alt[pk].b_ctl.as<ScanCtl>()->err
			            : m->alt[pk].batch_world ? &m->alt[pk].b_bpipe.as<Pipe>()->wstat[0] : &m->b_pipe.as<Pipe>()->wstat[m->alt[pk].fseq & (UFO_RING - 1u)];
	};
	scanQueue();
	m->cs = m->stream;
	{
		// node table: room for what this walk can create on top of what the updates in flight can
		if (!tableTakes(m, in_flight + bound)) {
			if (in_flight.blocks || countPendingAlts(m) > 0) {
				const int jrc = joinEnqueued(m);  // (the table cannot be exchanged under an update in flight -- scans on this scan's own grid included)
				if (jrc < 0) return jrc;
				scanQueue();
			}
			if (!tableTakes(m, bound)) {
				m->cs = m->stream;
				const int rc = growFor(m, bound);
				if (rc) return rc;
			}
		}
	}
	if (m->chg_enabled) {
		// change detection: every voxel of the ray grid may change (updates run one at a time in this mode, doInsert)
		const int crc = ensureChangeCap(m, (u64)fg.gr.bytes * 8u);
		if (crc) return crc;
	}
	m->scan_new_bound = bound;
	m->scan_id += 1;
	const bool big_grid = !gridFitsLds(fg.gr);
	{
		// hand-over records: the tiles' (k_tile), behind them the level-4 blocks' of a grid beyond LDS (k_up); new memory is
		// zeroed -- a record counts if it carries the walk's number
		const size_t want = (big_grid ? (size_t)UFO_BIG_MAX_TILES + UFO_FAST_MAX_TILES : (size_t)UFO_FAST_MAX_TILES) * sizeof(TileRec);
		if (m->b_tilerec.cap < want) {
			HIP_TRY(hipStreamSynchronize(m->stream));  // (a walk in flight reads the old array)
			HIP_TRY(m->b_tilerec.reserve(want));
			HIP_TRY(hipMemsetAsync(m->b_tilerec.p, 0, m->b_tilerec.cap, m->stream));
		}
		if (!m->b_upguess.p) {
			HIP_TRY(m->b_upguess.reserve(UFO_UPPER_MAX * sizeof(u32)));
			HIP_TRY(hipMemsetAsync(m->b_upguess.p, 0xFF, m->b_upguess.cap, m->stream));
		}
		if (big_grid && !m->b_upbits.p) {
			HIP_TRY(m->b_upbits.reserve(UFO_FAST_MAX_TILES / 8));
			HIP_TRY(hipMemsetAsync(m->b_upbits.p, 0, m->b_upbits.cap, m->stream));
		}
	}
	(a ? a->pending : m->pending) = true;
	(a ? a->deferred : m->deferred) = false;
	(a ? a->has_slot : m->has_slot) = true;
	if (!(!a && m->solo) && f > m->last_slot_fseq) m->last_slot_fseq = f;
	(a ? a->bound : m->bound) = bound;
	m->cs = m->stream;
	const bool solo = !a && m->solo;
	Pipe* pipe = solo ? m->b_bpipe.as<Pipe>() : m->b_pipe.as<Pipe>();
	const u32 bmax = (u32)std::max(1, std::min<int>(m->opt_batch_max, (int)UFO_BATCH_MAX));
	// without gates (a tool serialises kernels across streams) the map stream waits for the event behind the newest scan
	// half; k_claim then finds the scan complete and only takes its decision
	if (!solo) {
		if (!m->gates) HIP_TRY(hipStreamWaitEvent(m->stream, m->scan_ev, 0));
		hipLaunchKernelGGL(k_claim, dim3(1), dim3(64), 0, m->stream, pipe, (unsigned long long)f, bmax, ctl, gateTicks(m), a ? a->h_res : m->h_res,
		                   (unsigned long long)(a ? a->seq : m->seq));
	}
	{
		ProfScope ps(m, "k_fmerge");
		const u32 n4 = (u32)(fg.gr.bytes >> 4);
		// (a grid beyond LDS has no slabs to merge: sixteen of the kernel's seventeen waves per workgroup would only meet at its barriers)
		hipLaunchKernelGGL(k_fmerge, dim3(std::min<u32>((n4 + 63) / 64, 1024) + 1u, (u32)std::max(1, std::min(m->opt_fmerge_rows, 8))), dim3(big_grid ? 64 : 1024), 0, m->cs, fg, pipe,
		                   (unsigned long long)f, n4);
	}
	const float miss = (float)m->g.miss_log;  // insert depth 0 (OMB:311)
	{
		ProfScope ps(m, "k_tile");
		const u32 tw = (m->opt_tile_waves >= 1 && m->opt_tile_waves <= 4) ? (u32)m->opt_tile_waves : 4u;  // wavefronts (= tiles) per workgroup
		if (m->g.color)
			hipLaunchKernelGGL(k_tile<true>, dim3((fg.ntiles + tw - 1) / tw), dim3(64u * tw), 0, m->cs, m->t, m->g, fg, pipe, (unsigned long long)f,
			                   m->b_tilerec.as<TileRec>(), m->g.hit, miss, m->scan_id, prev_stat, changeLog(m), TileVol{});
		else
			hipLaunchKernelGGL(k_tile<false>, dim3((fg.ntiles + tw - 1) / tw), dim3(64u * tw), 0, m->cs, m->t, m->g, fg, pipe, (unsigned long long)f,
			                   m->b_tilerec.as<TileRec>(), m->g.hit, miss, m->scan_id, prev_stat, changeLog(m), TileVol{});
	}
	const u32 nwords3 = (fg.ntiles + 31u) / 32u;
	if (big_grid) {
		// a ray grid beyond LDS: level 4 in parallel (k_up), k_ftail starts above it -- the level-4 blocks are its "tiles"
		const FastGeo fu = makeUpGeo(fg);
		TileRec* recs_up = m->b_tilerec.as<TileRec>() + UFO_BIG_MAX_TILES;
		u32* up_bits = m->b_upbits.as<u32>();
		{
			ProfScope ps(m, "k_up");
			const dim3 gu((fu.ntiles * 8u + 255u) / 256u);
			if (m->g.color)
				hipLaunchKernelGGL(k_up<true>, gu, dim3(256), 0, m->cs, m->t, m->g, fg, pipe, (unsigned long long)f, m->b_tilerec.as<TileRec>(), recs_up, up_bits,
				                   m->scan_id, prev_stat);
			else
				hipLaunchKernelGGL(k_up<false>, gu, dim3(256), 0, m->cs, m->t, m->g, fg, pipe, (unsigned long long)f, m->b_tilerec.as<TileRec>(), recs_up, up_bits,
				                   m->scan_id, prev_stat);
		}
		ProfScope ps(m, "k_ftail");
		if (m->g.color)
			hipLaunchKernelGGL(k_ftail<true>, dim3(1), dim3(UFO_FTAIL_THREADS), 0, m->cs, m->t, m->g, fu, pipe, (unsigned long long)f, recs_up, m->scan_id, prev_stat,
			                   m->b_ctl_init.as<ScanCtl>(), up_bits, nwords3, (u32*)nullptr, m->b_upguess.as<u32>(), (u32)m->opt_ctl_dbg);
		else
			hipLaunchKernelGGL(k_ftail<false>, dim3(1), dim3(UFO_FTAIL_THREADS), 0, m->cs, m->t, m->g, fu, pipe, (unsigned long long)f, recs_up, m->scan_id, prev_stat,
			                   m->b_ctl_init.as<ScanCtl>(), up_bits, nwords3, (u32*)nullptr, m->b_upguess.as<u32>(), (u32)m->opt_ctl_dbg);
	} else {
		ProfScope ps(m, "k_ftail");
		if (m->g.color)
			hipLaunchKernelGGL(k_ftail<true>, dim3(1), dim3(UFO_FTAIL_THREADS), 0, m->cs, m->t, m->g, fg, pipe, (unsigned long long)f, m->b_tilerec.as<TileRec>(),
			                   m->scan_id, prev_stat, m->b_ctl_init.as<ScanCtl>(), (u32*)nullptr, nwords3, (u32*)nullptr, m->b_upguess.as<u32>(), (u32)m->opt_ctl_dbg);
		else
			hipLaunchKernelGGL(k_ftail<false>, dim3(1), dim3(UFO_FTAIL_THREADS), 0, m->cs, m->t, m->g, fg, pipe, (unsigned long long)f, m->b_tilerec.as<TileRec>(),
			                   m->scan_id, prev_stat, m->b_ctl_init.as<ScanCtl>(), (u32*)nullptr, nwords3, (u32*)nullptr, m->b_upguess.as<u32>(), (u32)m->opt_ctl_dbg);
	}
	HIP_TRY(hipGetLastError());
	return UFOMAP_OK;
}

// The scans that have no slot on the map stream yet get one: a slot for the newest of them takes the others along (k_claim;
// one slot per batch_max scans). (The current set holds the newest integration; what waits is always the newest scans,
// and they share a ray grid: fastScanPhase.)
int flushDeferred(ufomap_map* m, bool publish)
{
	// (publish = false: a scan whose descriptor the host still keeps back stays as it is -- no slot may wait for it --
	// and what is older gets its slots)
	if (publish) {
		const int prc = publishScanDone(m);
		if (prc) return prc;
	}
	int idx[kAlt + 1], n = 0;
	for (int i = 0; i < kAlt; ++i)
		if (m->alt[i].pending && m->alt[i].deferred) idx[n++] = i;
	std::sort(idx, idx + n, [&](int a, int b) { return m->alt[a].seq < m->alt[b].seq; });
	if (m->pending && m->deferred && !m->sd_pending) idx[n++] = -1;
	const int bmax = std::max(1, std::min<int>(m->opt_batch_max, (int)UFO_BATCH_MAX));
	for (int a = 0; a < n; ++a) {
		if (a + 1 == n || 0 == (a + 1) % bmax) {
			const int rc = enqueueSlot(m, idx[a]);
			if (rc) return rc;
		} else {
			// (goes with the slot enqueued for a newer scan: joined like any other integration, by its own word in pinned memory)
			HandOver* const h = idx[a] < 0 ? nullptr : &m->alt[idx[a]];
			(h ? h->deferred : m->deferred) = false;
			(h ? h->has_slot : m->has_slot) = false;
			(h ? h->bound : m->bound) = Need{};
		}
	}
	return UFOMAP_OK;
}

// The ray grid for the next depth-0 scans from a box of ray cells [mn, mx]: first choice the union of the box with the
// grid predicted so far (a sensor that moves about a room keeps producing boxes inside one hull, and a prediction that
// covers the hull never misses again), second choice the box alone, each with up to two node blocks of margin for sensor
// motion -- as long as the ray kernel still fits its bit grid and segment queue in LDS.
bool gridFromBox(bool had, const Grid& prev, const i32 bmn[3], const i32 bmx[3], Grid* out, bool allow_big = false)
{
	// (big: no grid that the ray kernel can hold in LDS -- then a grid in HBM, k_fselect / k_cast<2> / k_up, up to 8 MiB of bits)
	for (int big = 0; big <= (allow_big ? 1 : 0); ++big)
	for (int pass = (had && 0 == prev.depth) ? 0 : 1; pass < 2; ++pass) {
		for (int margin = 2; margin >= 0; --margin) {
			i32 mn[3], mx[3];
			for (int k = 0; k < 3; ++k) {
				mn[k] = bmn[k] - 2 * margin;
				mx[k] = bmx[k] + 2 * margin;
				if (0 == pass) {
					// interior of the previous grid (makeGrid pads by one block on either side)
					mn[k] = std::min(mn[k], prev.base[k] + 2);
					mx[k] = std::max(mx[k], prev.base[k] + 2 * prev.nb[k] - 3);
				}
			}
			Grid gr;
			if (makeGrid(mn, mx, 0, &gr)) continue;
			const bool packed = 2 * gr.nb[0] < 1023 && 2 * gr.nb[1] < 1023 && 2 * gr.nb[2] < 1023;
			const u64 bytes1 = (u64)(gridRowBits(gr) >> 3) * (2ull * (u64)gr.nb[1]) * (2ull * (u64)gr.nb[2]);
			if (!packed) continue;
			if (big ? bytes1 > (8ull << 20) : ((bytes1 + 15) & ~15ull) + UFO_CAST_LDS_EXTRA > (160u << 10) - 512u) continue;
			gr.layout = 1;
			gr.bytes = (bytes1 + 15) & ~15ull;
			*out = gr;
			return true;
		}
	}
	return false;
}

// Predict the ray grid of the next depth-0 scan from the box of the one just finished.
void predictGrid(ufomap_map* m)
{
	const bool had = m->spec_valid;
	const Grid prev = m->spec_grid;
	m->spec_valid = false;
	const ScanArgs& a = m->args;
	if (!m->opt_spec || !m->opt_merge || !m->opt_cast || !m->opt_bits || !m->opt_dda_seg || m->opt_dda_mode > 0) return;
	if (0 != a.depth || 0 == a.n || 0 == m->h_ctl->n_rays) return;  // (simple ray casting: the same box of ray cells, k_fcast_simple)
	m->spec_valid = gridFromBox(had, prev, m->h_ctl->mb_min, m->h_ctl->mb_max, &m->spec_grid, 0 != m->opt_big && 0 != m->opt_fast && m->g.L >= 6);
}

int redoBatchStep(ufomap_map* m);
void predictCommonGrid(ufomap_map* m);
