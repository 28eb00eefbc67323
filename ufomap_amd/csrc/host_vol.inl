// host side of the volume path (vol_kernels.h): eligibility and the tile grids of the levels, the scan half (volScan, called by
// scanPhase once the boxes are known) and the tree update (volMapPhase). Included by ufomap_hip.hip inside its anonymous namespace.

// b_vaux, 32-bit words: [0..63] the reserve's counters, [64] tiles listed, [65] tiles done, [66..67] blocks the listed tiles touch,
// from UFO_VAUX_UPCNT on k_up's 64 pairs of counters (blocks touched / created above the tiles; k_ftail folds and clears them)
#define UFO_VAUX_UPCNT 256u
// ... then the reserve's 64 counters, a cache line each (UFO_RESV_STRIDE words apart; words [0..63] are no longer used)
#define UFO_VAUX_RESV (UFO_VAUX_UPCNT + UFO_UPCNT_WORDS)
// ... then 64 64-bit counters, 128 bytes apart: the step counts of k_vwalk's waves (k_vlist folds and clears them)
#define UFO_VAUX_STEPS (UFO_VAUX_RESV + 64u * UFO_RESV_STRIDE)
#define UFO_VAUX_BYTES ((UFO_VAUX_STEPS + 64u * 32u) * 4u)

// the grid of the level above fg's (what k_up writes when fg is what it reads)
FastGeo upGeoOf(const FastGeo& fg)
{
	FastGeo u = fg;
	u64 nt = 1;
	for (int a = 0; a < 3; ++a) {
		u.tbase[a] = fg.tbase[a] >> 1;
		u.nt[a] = (u32)(((fg.tbase[a] + (i32)fg.nt[a] - 1) >> 1) - u.tbase[a] + 1);
		nt *= u.nt[a];
	}
	u.ntiles = (u32)std::min<u64>(nt, 0xFFFFFFFFull);
	u.tl = fg.tl + 1u;
	return u;
}

// The assumptions behind the per-XCD copies of the brick grid (vol_kernels.h: k_vselftest), checked on the device once per process and device:
// 1 = they hold, -1 = they do not (the volume path stays off). Also -1 under a compiler mode that splits a workgroup's waves over
// CUs with separate L1s is not needed: the atomics go to the L2 either way.
int volSelfTest(ufomap_map* m)
{
	// (per DEVICE: what is tested -- XCC_ID's range, workgroup-scope atomics at the XCD's L2 -- is a property of the device, and a
	// process may hold maps on several; run when a map is created, ufomap_map_create, not inside a scan that may be timed: ADVICE r5)
	static std::atomic<int> states[64];
	std::atomic<int>& state = states[(unsigned)m->device & 63u];
	int s = state.load(std::memory_order_acquire);
	if (s) return s;
	u32* d = nullptr;
	if (hipMalloc((void**)&d, (8 * 64 + 64) * 4) != hipSuccess) return -1;
	u32 h[8 * 64 + 64];
	bool ok = hipMemsetAsync(d, 0, sizeof(h), m->stream) == hipSuccess;
	if (ok) {
		hipLaunchKernelGGL(k_vselftest, dim3(8u * UFO_VSELF_BLOCKS), dim3(256), 0, m->stream, d, d + 8 * 64);
		ok = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, m->stream) == hipSuccess && hipStreamSynchronize(m->stream) == hipSuccess;
	}
	(void)hipFree(d);
	u64 sum = 0;
	if (ok)
		for (int k = 0; k < 8; ++k) sum += h[k * 64];
	ok = ok && 0 == h[8 * 64] && sum == (u64)8 * UFO_VSELF_BLOCKS * 256 * UFO_VSELF_ADDS;
	s = ok ? 1 : -1;
	state.store(s, std::memory_order_release);
	if (!ok) fprintf(stderr, "[ufomap_amd] the per-XCD atomics self-test failed (%llu additions counted): the volume path is off, scans take the general path\n", (unsigned long long)sum);
	return s;
}

// A depth-0 scan of a plain map whose ray box the steady-state path cannot take (more than 1022 cells per axis, or a bit grid
// beyond 8 MiB) and whose brick grids fit the scratch limit.
bool volPlan(const ufomap_map* m, const Grid& gr, unsigned depth, int simple, unsigned early_stopping, const uint8_t* d_rgb, VolPlan* vp)
{
	if (!m->opt_vol || m->keys_mode || 0 != depth || simple || early_stopping || ((d_rgb || m->g.color) && !m->opt_vol_color) || m->chg_enabled || m->g.L < 6) return false;
	const bool packed = 2 * gr.nb[0] < 1023 && 2 * gr.nb[1] < 1023 && 2 * gr.nb[2] < 1023;
	const u64 bytes1 = (u64)(gridRowBits(gr) >> 3) * (2ull * (u64)gr.nb[1]) * (2ull * (u64)gr.nb[2]);
	if (m->opt_vol < 2 && packed && bytes1 <= (8ull << 20)) return false;  // (the fast path's sizes; option vol = 2: tests run small scans here)
	for (int a = 0; a < 3; ++a)
		if (gr.nb[a] > (1 << 20)) return false;
	FastGeo fg{};
	fg.gr = gr;
	u64 nt = 1;
	for (int a = 0; a < 3; ++a) {
		fg.tbase[a] = gr.base[a] >> 3;
		fg.nt[a] = (u32)(((gr.base[a] + 2 * gr.nb[a] - 1) >> 3) - fg.tbase[a] + 1);
		nt *= fg.nt[a];
	}
	if (nt >= (1ull << 28)) return false;  // (32-bit word indices: 8 words per tile)
	fg.ntiles = (u32)nt;
	fg.tl = 3;
	// eight copies of M (one per XCD), the merged M, H (64 bytes per tile each), the tile bitmaps, the list, the records
	if (nt * (8 * 64 + 64 + 64 + 4 + 1 + 1 + 40) > m->scratch_limit) return false;
	vp->n = 0;
	vp->lv[vp->n++] = fg;
	vp->rec_total = nt;
	UpperGeo ug;
	for (;;) {
		const FastGeo& top = vp->lv[vp->n - 1];
		if (top.tl >= m->g.L) return false;
		if (vp->n > 1 && top.ntiles <= UFO_FAST_MAX_TILES && makeUpperGeo(top, m->g.L, &ug) <= UFO_UPPER_MAX) break;
		if (vp->n >= 20 || top.tl + 1u >= m->g.L) return false;
		vp->lv[vp->n] = upGeoOf(top);
		vp->rec_total += vp->lv[vp->n].ntiles;
		++vp->n;
	}
	for (int a = 0; a < 3; ++a) {
		vp->vg.cbase[a] = fg.tbase[a] * 8;
		vp->vg.nt[a] = fg.nt[a];
	}
	vp->vg.ntiles = fg.ntiles;
	if (volSelfTest(const_cast<ufomap_map*>(m)) < 0) return false;
	return true;
}

// Scan half of the volume path, after k_classify / k_select / k_reduce_boxes and the read-back of the boxes (scanPhase): hit
// voxels and ray cells into the brick grids, the list of active tiles. Returns 1 when the scan has to take the general path
// after all (a ray clipped at the map cube): nothing but scratch has been touched.
int volScan(ufomap_map* m, const D3& sensor, const VolPlan& vp, u32 n_hits, u32 n_rays)
{
	const u64 nt = vp.vg.ntiles;
	const size_t tbw = (size_t)volTbWords((u32)nt);
	const size_t cm = m->b_vM.cap, ch = m->b_vH.cap, cr = m->b_vrec.cap, ct = m->b_vtb.cap;
	HIP_TRY(m->b_vM.reserve(volCopyWords((u32)nt) * 8 * 8));  // (eight copies, each in whole 256-byte pieces)
	HIP_TRY(m->b_vMm.reserve(nt * 64));
	HIP_TRY(m->b_vH.reserve(nt * 64));
	HIP_TRY(m->b_vtb.reserve(tbw * 4 * 8));
	HIP_TRY(m->b_vlist.reserve(nt * 4));
	HIP_TRY(m->b_vcopies.reserve(nt));
	HIP_TRY(m->b_vslots.reserve(nt * 4));
	HIP_TRY(m->b_vrec.reserve(vp.rec_total * sizeof(TileRec)));
	HIP_TRY(m->b_vaux.reserve(UFO_VAUX_BYTES));
	HIP_TRY(m->b_vupbits.reserve(UFO_FAST_MAX_TILES / 8));
	if (cm != m->b_vM.cap || ch != m->b_vH.cap || ct != m->b_vtb.cap) m->vol_dirty = true;
	// (a different tile grid: what an aborted walk may have left marked lies elsewhere -- and a clean walk leaves nothing)
	if (cr != m->b_vrec.cap) HIP_TRY(hipMemsetAsync(m->b_vrec.p, 0, m->b_vrec.cap, m->cs));  // (a record counts if it carries the walk's number)
	if (m->vol_dirty) {
		// (steady state: k_vlist leaves the tile bitmaps clean, k_tile the copies of M and H)
		HIP_TRY(hipMemsetAsync(m->b_vM.p, 0, m->b_vM.cap, m->cs));
		HIP_TRY(hipMemsetAsync(m->b_vH.p, 0, m->b_vH.cap, m->cs));
		HIP_TRY(hipMemsetAsync(m->b_vtb.p, 0, m->b_vtb.cap, m->cs));
	}
	m->vol_dirty = true;  // (until the tree update has left the grids clean)
	HIP_TRY(hipMemsetAsync(m->b_vaux.p, 0, UFO_VAUX_BYTES, m->cs));
	HIP_TRY(hipMemsetAsync(m->b_vupbits.p, 0, m->b_vupbits.cap, m->cs));
	ScanCtl* ctl = m->b_ctl.as<ScanCtl>();
	u32* aux = m->b_vaux.as<u32>();  // [0..63] the reserve's counters, [64] tiles listed, [65] tiles done, [66..67] blocks the listed tiles touch
	if (n_hits) {
		ProfScope ps(m, "k_vhits");
		hipLaunchKernelGGL(k_vhits, gridFor(n_hits), dim3(256), 0, m->cs, m->g, vp.vg, m->b_hit_code.as<u64>(), ctl, m->b_vH.as<u64>(), ctl);
	}
	const u32* order = nullptr;
	if (0 == (m->opt_vol_mode & 4) && n_rays >= 4096u) {
		// the rays bundled by direction (a counting sort over a cube map of directions): a wave's rays share bricks all the way
		ProfScope ps(m, "k_vbin");
		HIP_TRY(m->b_vbin.reserve(((size_t)UFO_VBINS + 128u + 2u * (size_t)n_rays) * 4));
		u32* hist = m->b_vbin.as<u32>();
		u32* tot = hist + UFO_VBINS;  // (sums of k_vbin_scan1's workgroups)
		u32* bin_of = tot + 128u;
		u32* ord = bin_of + n_rays;
		HIP_TRY(hipMemsetAsync(hist, 0, (size_t)UFO_VBINS * 4, m->cs));
		const dim3 gb((n_rays + 255u) / 256u);
		hipLaunchKernelGGL(k_vbin_count, gb, dim3(256), 0, m->cs, sensor, m->b_ray_end.as<D3>(), ctl, bin_of, hist);
		hipLaunchKernelGGL(k_vbin_scan1, dim3(UFO_VBINS / 1024u), dim3(1024), 0, m->cs, hist, tot);
		hipLaunchKernelGGL(k_vbin_scatter, gb, dim3(256), 0, m->cs, ctl, bin_of, hist, ord, (const u32*)tot);
		order = ord;
	}
	// the rays cut into segments of ~K cells that lanes walk one each (vol_kernels.h, round 5) -- unless the segment list would not
	// fit the scratch limit, or option vol_mode bit 4 asks for the one-lane-per-ray kernel (kept as the cross-check)
	const u32 K = (u32)std::max(5, m->opt_vol_seg);
	const u32 per = (((n_rays + 7u) / 8u) + 255u) & ~255u;  // bundled rays per region (an eighth of the scan, whole blocks)
	const u64 ext = 2ull * ((u64)m->gridM.nb[0] + (u64)m->gridM.nb[1] + (u64)m->gridM.nb[2]);
	const u64 seg_cap = (u64)per * (ext / (K - 3u) + 2ull);  // (a ray of l1 cells: at most l1 / (K - 3) + 1 segments)
	const bool segmented = 0 == (m->opt_vol_mode & 16) && 8ull * seg_cap < (1ull << 31) &&
	                       8ull * seg_cap * UFO_VSEG_BYTES + (u64)n_rays * sizeof(VRay) <= m->scratch_limit;
	if (segmented) {
		HIP_TRY(m->b_vrays.reserve((size_t)n_rays * sizeof(VRay)));
		HIP_TRY(m->b_vsegs.reserve((size_t)(8ull * seg_cap) * UFO_VSEG_BYTES));
		HIP_TRY(m->b_vsegcnt.reserve(8 * UFO_VSEG_CNT_STRIDE * 4));
		HIP_TRY(hipMemsetAsync(m->b_vsegcnt.p, 0, 8 * UFO_VSEG_CNT_STRIDE * 4, m->cs));
		const VSegs sg = volSegViews(m->b_vsegs.p, (size_t)(8ull * seg_cap));
		{
			ProfScope ps(m, "k_vcut");
			hipLaunchKernelGGL(k_vcut, dim3((n_rays + 255u) / 256u), dim3(256), 0, m->cs, m->g, sensor, m->gridM, vp.vg, m->b_vM.as<u64>(), m->b_vtb.as<u32>(), m->b_ray_end.as<D3>(),
			                   ctl, ctl, order, K, per, (u32)seg_cap, m->b_vrays.as<VRay>(), sg, m->b_vsegcnt.as<u32>());
		}
		{
			ProfScope ps(m, "k_vdda");  // (the walk itself keeps the name the bench's per-kernel table knows)
			const u32 G = (u32)std::max(1, m->opt_vol_walk_blocks);
			hipLaunchKernelGGL(k_vwalk, dim3(8u * G), dim3(256), (size_t)m->opt_vol_walk_lds, m->cs, m->g, vp.vg, m->b_vM.as<u64>(), m->b_vtb.as<u32>(), m->b_vrays.as<VRay>(), sg,
			                   m->b_vsegcnt.as<u32>(), (u32)seg_cap, ctl, ctl, (u32)m->opt_vol_mode, reinterpret_cast<unsigned long long*>(aux + UFO_VAUX_STEPS));
		}
	} else {
		ProfScope ps(m, "k_vdda");
		const u32 nblk = ((n_rays + 255u) / 256u + 7u) & ~7u;  // (a multiple of 8: an eighth of the cloud per XCD)
		hipLaunchKernelGGL(k_vdda, dim3(nblk), dim3(256), 0, m->cs, m->g, sensor, m->gridM, vp.vg, m->b_vM.as<u64>(), m->b_vtb.as<u32>(), m->b_ray_end.as<D3>(), ctl, ctl, (u32)m->opt_vol_mode,
		                   order);
	}
	{
		ProfScope ps(m, "k_vlist");
		hipLaunchKernelGGL(k_vlist, gridFor(tbw, 256, 2048), dim3(256), 0, m->cs, m->b_vtb.as<u32>(), (u32)nt, m->b_vlist.as<u32>(), m->b_vcopies.as<uint8_t>(), aux + 64, m->b_vrec.as<TileRec>(),
		                   m->b_vslots.as<u32>(), reinterpret_cast<unsigned long long*>(aux + UFO_VAUX_STEPS), ctl);
	}
	HIP_TRY(hipMemcpyAsync(m->h_ctl, m->b_ctl.p, sizeof(ScanCtl), hipMemcpyDeviceToHost, m->cs));
	HIP_TRY(hipMemcpyAsync(&m->vol_count, aux + 64, 4, hipMemcpyDeviceToHost, m->cs));
	HIP_TRY(hipStreamSynchronize(m->cs));
	if (m->h_ctl->err & ERR_VOL) {
		// the general path takes the scan: the flag and the step count of the abandoned walk go, the head kernels' results stay
		hipLaunchKernelGGL(k_ctl_clear, dim3(1), dim3(1), 0, m->cs, ctl, (u32)(ERR_VOL | ERR_RUNAWAY));
		HIP_TRY(hipMemsetAsync(&ctl->n_steps, 0, 8, m->cs));
		++m->n_vol_fallback;
		return 1;
	}
	const int erc = ctlError(m);
	if (erc) return erc;
	if (0 == m->vol_count) {
		// (no ray cell was marked -- cannot happen with n_rays > 0, but a walk over no tiles is not a launch: the general path)
		++m->n_vol_fallback;
		HIP_TRY(hipMemsetAsync(&ctl->n_steps, 0, 8, m->cs));
		return 1;
	}
	m->vplan = vp;
	m->vol = true;
	++m->n_vol;
	return UFOMAP_OK;
}

// One pass of the walk over the listed tiles: k_tile<VOL> (tiles whose records carry the walk's number are done), k_up level
// after level, k_ftail -- which stores the finished control block and the walk's done word to pinned memory.
int volWalkEnqueue(ufomap_map* m, bool retry = false)
{
	const VolPlan& vp = m->vplan;
	const u32 T = m->vol_count;
	ScanCtl* ctl = m->b_ctl.as<ScanCtl>();
	Pipe* pipe = m->b_bpipe.as<Pipe>();
	u32* aux = m->b_vaux.as<u32>();
	const float miss = (float)m->g.miss_log;  // insert depth 0 (OMB:311)
	const FastGeo& fg = vp.lv[0];
	TileRec* recs = m->b_vrec.as<TileRec>();
	const u64 lim_total = (u64)m->t.nG * 9 / 10 > m->used_g ? (u64)m->t.nG * 9 / 10 - m->used_g : 0;  // (new tile groups: the directory at most 90 % full)
	m->h_res->err = ERR_NOT_STORED;
	*reinterpret_cast<volatile unsigned long long*>(m->h_res + 1) = 0ull;
	m->done_by_flag = true;
	hipLaunchKernelGGL(k_vreset, dim3(1), dim3(64), 0, m->stream, aux + UFO_VAUX_RESV, ctl, (u32)(ERR_GROW | ERR_PREV));
	TileVol va{};
	va.Mx = m->b_vM.as<u64>();
	va.Mm = m->opt_vol_keep ? m->b_vMm.as<u64>() : nullptr;
	va.H = m->b_vH.as<u64>();
	va.list = m->b_vlist.as<u32>();
	va.copies = m->b_vcopies.as<uint8_t>();
	va.slots = m->b_vslots.as<u32>();
	va.retry = retry ? 1u : 0u;
	va.count = T;
	va.resv = aux + UFO_VAUX_RESV;
	va.resv_lim = (u32)std::min<u64>(lim_total / 64, 0x7FFFFFFFull);
	// colour maps (OccupancyMapColor, occupancy_map_color.h:177-287): the colour instances of the same three kernels -- a voxel that
	// receives a hit takes its first point's colour (found through the scan's hit hash), the summaries carry colours upwards
	const bool color = m->g.color;
	if (color) {
		const u32 hcap = m->hh_mask + 1u;
		va.hh = HitHash{m->b_hh_keys.as<u64>(), reinterpret_cast<u32*>(m->b_hh_keys.as<u64>() + hcap), m->hh_mask};
		va.rgb = m->vol_rgb;
	}
	{
		ProfScope ps(m, "k_tile");
		if (color)
			hipLaunchKernelGGL((k_tile<true, true>), dim3((T + 3) / 4), dim3(256), 0, m->stream, m->t, m->g, fg, pipe, 0ull, recs, m->g.hit, miss, m->vol_scan_id,
			                   (const u32*)nullptr, ChangeLog{nullptr, 0u, m->g.L}, va);
		else
			hipLaunchKernelGGL((k_tile<false, true>), dim3((T + 3) / 4), dim3(256), 0, m->stream, m->t, m->g, fg, pipe, 0ull, recs, m->g.hit, miss, m->vol_scan_id,
			                   (const u32*)nullptr, ChangeLog{nullptr, 0u, m->g.L}, va);
	}
	TileRec* below = recs;
	for (int k = 1; k < vp.n; ++k) {
		TileRec* above = below + vp.lv[k - 1].ntiles;
		ProfScope ps(m, "k_up");
		const dim3 gu((u32)(((u64)vp.lv[k].ntiles * 8u + 255u) / 256u));
		u32* ub = (k + 1 == vp.n) ? m->b_vupbits.as<u32>() : (u32*)nullptr;
		if (color)
			hipLaunchKernelGGL(k_up<true>, gu, dim3(256), 0, m->stream, m->t, m->g, vp.lv[k - 1], pipe, 0ull, below, above, ub, m->vol_scan_id, (const u32*)nullptr, aux + UFO_VAUX_UPCNT);
		else
			hipLaunchKernelGGL(k_up<false>, gu, dim3(256), 0, m->stream, m->t, m->g, vp.lv[k - 1], pipe, 0ull, below, above, ub, m->vol_scan_id, (const u32*)nullptr, aux + UFO_VAUX_UPCNT);
		below = above;
	}
	{
		ProfScope ps(m, "k_ftail");
		if (color)
			hipLaunchKernelGGL(k_ftail<true>, dim3(1), dim3(UFO_FTAIL_THREADS), 0, m->stream, m->t, m->g, vp.lv[vp.n - 1], pipe, 0ull, below, m->vol_scan_id, (const u32*)nullptr,
			                   m->b_ctl_init.as<ScanCtl>(), m->b_vupbits.as<u32>(), 0u, aux + UFO_VAUX_UPCNT);
		else
			hipLaunchKernelGGL(k_ftail<false>, dim3(1), dim3(UFO_FTAIL_THREADS), 0, m->stream, m->t, m->g, vp.lv[vp.n - 1], pipe, 0ull, below, m->vol_scan_id, (const u32*)nullptr,
			                   m->b_ctl_init.as<ScanCtl>(), m->b_vupbits.as<u32>(), 0u, aux + UFO_VAUX_UPCNT);
	}
	HIP_TRY(hipGetLastError());
	return UFOMAP_OK;
}

// The walk that was enqueued is awaited. Blocks are created against a reserve; when it ran out (ERR_GROW) the table is exchanged
// for a larger one and the tiles that stood back are run -- the others' records carry the walk's number.
int volWalkFinish(ufomap_map* m)
{
	const u32 T = m->vol_count;
	const FastGeo& fg = m->vplan.lv[0];
	TileRec* recs = m->b_vrec.as<TileRec>();
	u32* aux = m->b_vaux.as<u32>();
	m->vol_walk = false;
	for (int attempt = 0;; ++attempt) {
		if (attempt > 8) return fail(UFOMAP_ERR_CAPACITY, "the node table kept running out of room during one update (internal error)");
		HIP_TRY(hipStreamSynchronize(m->stream));
		if (!(m->h_res->err & ERR_GROW) || (m->h_res->err & (ERR_NOT_STORED | ERR_TABLE_FULL))) break;
		// ---- the reserve ran out: a larger table, then the tiles that stood back ----
		++m->n_vol_grow;
		HIP_TRY(hipMemsetAsync(aux + 65, 0, 4, m->stream));
		hipLaunchKernelGGL(k_vfix, dim3((T + 255u) / 256u), dim3(256), 0, m->stream, m->t, m->g, fg, m->b_vlist.as<u32>(), T, recs, m->vol_scan_id, aux + 65);
		u32 n_done = 0;
		HIP_TRY(hipMemcpyAsync(&n_done, aux + 65, 4, hipMemcpyDeviceToHost, m->stream));
		HIP_TRY(hipStreamSynchronize(m->stream));
		// (the re-hash counts the groups and blocks it copies: what the tiles that are done have created included)
		{
			Need need;
			need.groups = (u64)(T - std::min(T, n_done)) + (u64)std::min(T, n_done);  // (the groups of the tiles that are done are not in used_g yet)
			const u64 nG = std::max<u64>((u64)m->t.nG + m->t.nG / 2, (m->used_g + need.groups) * 5 / 4 + 64);
			if ((u64)m->t.capU + UFO_GROUP * nG > (1ull << 31)) return fail(UFOMAP_ERR_CAPACITY, "node table would exceed 2^31 blocks");
			const hipStream_t keep = m->cs;
			m->cs = m->stream;
			const int rc = growTable(m, (u32)((nG + 63) & ~63ull), m->t.capU);
			m->cs = keep;
			if (rc) return rc;
		}
		hipLaunchKernelGGL(k_vfix, dim3((T + 255u) / 256u), dim3(256), 0, m->stream, m->t, m->g, fg, m->b_vlist.as<u32>(), T, recs, m->vol_scan_id, aux + 65);
		HIP_TRY(hipMemsetAsync(m->b_vupbits.p, 0, m->b_vupbits.cap, m->stream));
		const int rc = volWalkEnqueue(m, true);
		if (rc) return rc;
	}
	if (0 == m->h_res->err) m->vol_dirty = false;
	return UFOMAP_OK;
}

// Tree update of the volume path on the map stream: the node table sized for what the walk can add, the walk enqueued; awaited
// here unless the call is asynchronous (then by whatever joins the integration: finishPending).
int volMapPhase(ufomap_map* m, bool leave_enqueued = false)
{
	const VolPlan& vp = m->vplan;
	const u32 T = m->vol_count;
	m->cs = m->stream;
	m->hit_grid = false;
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
	{
		const size_t pc = m->b_bpipe.cap;
		HIP_TRY(m->b_bpipe.reserve(sizeof(Pipe)));
		if (pc != m->b_bpipe.cap) HIP_TRY(hipMemsetAsync(m->b_bpipe.p, 0, sizeof(Pipe), m->stream));
	}
	Pipe* pipe = m->b_bpipe.as<Pipe>();
	{
		DescPack pk{};
		ScanDesc& d = pk.d[0];
		d.ctl = m->b_ctl.as<ScanCtl>();
		d.host_result = m->h_res;
		d.done_value = (unsigned long long)m->seq;
		d.tile_bits = m->b_vupbits.as<u32>();  // (never read: k_ftail clears nwords3 = 0 words of it)
		d.geo = 0;
		hipLaunchKernelGGL(k_batch_descs, dim3(1), dim3(64), 0, m->stream, pipe, pk, 1u);
	}
	// Node table. The first region takes what the walk can create above the tiles (bounded per level by the listed tiles and by
	// the level's cells: a few per cent of the table); tile groups: a map that holds little of what the scan touches gets room
	// for every listed tile at once instead of finding out half-way, else the walk goes ahead on what there is -- new groups
	// come out of a reserve, and a tile that finds it used up stands back.
	{
		Need need;
		for (int k = 1; k < vp.n; ++k) need.upper += std::min<u64>(T, vp.lv[k].ntiles);
		need.upper += 2048;
		need.blocks = need.upper;
		if (m->opt_vol_pregrow && m->used_g * 2 < T) need.groups = T;
		if (!tableTakes(m, need)) {
			// (the first region for twice the bound: what this scan creates there counts as fill when the next scan comes with the
			// same bound on top -- the first warm scan of round 4's first bench run re-hashed 6.8 GB for 0.1 GB of first region)
			need.upper *= 2;
			const int rc = growFor(m, need);
			if (rc) return rc;
		}
	}
	m->scan_id += 1;
	m->vol_scan_id = m->scan_id;
	m->scan_new_bound = Need{};
	const int erc = volWalkEnqueue(m);
	if (erc) return erc;
	m->fast = true;  // (finishPending: the finished control block is in pinned memory, k_ftail left the device copy clean)
	m->pending = true;
	m->vol_walk = true;
	if (leave_enqueued) return UFOMAP_OK;
	return volWalkFinish(m);
}
