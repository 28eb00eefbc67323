// the map as the reference's byte stream, both ways: LZ4, write / writeData (serialiseNodes*), read / readData. Included by
// ufomap_hip.hip inside its extern "C" block.
// liblz4, loaded at run time (the reference links it for its I/O only: octree.h:1430-1486)
extern "C++" {
namespace
{
struct Lz4 {
	int (*bound)(int) = nullptr;
	int (*fast)(const char*, char*, int, int, int) = nullptr;
	int (*hc)(const char*, char*, int, int, int) = nullptr;
	int (*safe)(const char*, char*, int, int) = nullptr;
	bool ok = false;
};
const Lz4& lz4()
{
	static Lz4 z = [] {
		Lz4 r;
		void* h = nullptr;
		for (const char* name : {"liblz4.so.1", "liblz4.so", "/usr/lib/x86_64-linux-gnu/liblz4.so.1"}) {
			h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
			if (h) break;
		}
		if (h) {
			r.bound = reinterpret_cast<int (*)(int)>(dlsym(h, "LZ4_compressBound"));
			r.fast = reinterpret_cast<int (*)(const char*, char*, int, int, int)>(dlsym(h, "LZ4_compress_fast"));
			r.hc = reinterpret_cast<int (*)(const char*, char*, int, int, int)>(dlsym(h, "LZ4_compress_HC"));
			r.safe = reinterpret_cast<int (*)(const char*, char*, int, int)>(dlsym(h, "LZ4_decompress_safe"));
			r.ok = r.bound && r.fast && r.hc && r.safe;
		}
		return r;
	}();
	return z;
}

// b_ser[0], 32-bit words: [0..31] live blocks per level, [32..63] first list entry per level, [96..97] the stream's length, [100..163]
// the SerLevels the device-side kernels read, from UFO_SER_BLK_WORD on one row of 32 counts per workgroup of the listing kernels
#define UFO_SER_BLK_WORD 256u
#define UFO_SER_CNT_BYTES ((UFO_SER_BLK_WORD + 32u * UFO_SER_NB_MAX) * 4u)

// The node stream without the host in the middle (map_kernels.h: k_ser_prefix ... k_ser_copy_out): one synchronisation.
// Returns 1 when the long way has to be taken (a map too large for the bound, no live root block).
// (in_place: the caller reads the stream where the device left it, in the handle's pinned buffer -- *in_place = its length -- instead of
// from a copy in `data`: a server's per-scan publish of 300 KB pays for every copy)
int serialiseNodesShort(ufomap_map* m, const SerArgs& sa, std::vector<uint8_t>& data, size_t* in_place = nullptr)
{
	const u32 D = m->g.color ? 7u : 4u;
	const u32 L = m->g.L;
	if (!m->opt_ser_short || 0 == m->used_est || L <= sa.min_depth) return 1;
	const u64 bound = 1ull + (u64)m->used_est * (1ull + 8ull * D) + 64ull;  // every block: a mask byte and eight leaf payloads at most
	if (bound > (32ull << 20)) return 1;
	DevBuf &b_cnt = m->b_ser[0], &b_list = m->b_ser[1], &b_size = m->b_ser[2], &b_off = m->b_ser[3], &b_out = m->b_ser[4];
	const size_t ncap = (size_t)m->t.mask + 1;
	HIP_TRY(b_cnt.reserve(UFO_SER_CNT_BYTES));
	HIP_TRY(b_list.reserve(((size_t)m->used_est + 8) * 4));
	HIP_TRY(b_size.reserve(ncap * 8));
	HIP_TRY(b_off.reserve(ncap * 8));
	HIP_TRY(b_out.reserve((bound + 15) & ~15ull));
	if (m->h_out_cap < bound) {
		if (m->h_out) (void)hipHostFree(m->h_out);
		m->h_out = nullptr;
		m->h_out_cap = 0;
		const size_t want = (size_t)((bound + bound / 2 + 4095) & ~4095ull);
		HIP_TRY(hipHostMalloc((void**)&m->h_out, want));
		m->h_out_cap = want;
	}
	if (!m->h_ser) HIP_TRY(hipHostMalloc((void**)&m->h_ser, 512));
	volatile unsigned long long* h_total = reinterpret_cast<volatile unsigned long long*>(m->h_ser + 256);
	u32* d_cnt = b_cnt.as<u32>();
	unsigned long long* d_total = reinterpret_cast<unsigned long long*>(d_cnt + 96);
	SerLevels* d_lv = reinterpret_cast<SerLevels*>(d_cnt + 100);
	hipStream_t st = m->stream;
	static const bool trace = nullptr != getenv("UFOMAP_TRACE_SER");
	const auto t0 = std::chrono::steady_clock::now();
	const u32 nsb = serBlocks(ncap);  // (the listing kernels: a contiguous share of the slots per workgroup, no atomics on device memory)
	u32* d_blk = d_cnt + UFO_SER_BLK_WORD;
	hipLaunchKernelGGL(k_ser_count, dim3(nsb), dim3(256), 0, st, m->t, m->g, d_blk);
	const u32 list_cap = (u32)std::min<u64>(m->used_est + 8, 0xFFFFFFFFull);
	hipLaunchKernelGGL(k_ser_prefix, dim3(1), dim3(1024), 0, st, d_cnt, d_lv, list_cap, d_blk, nsb);
	hipLaunchKernelGGL(k_ser_collect, dim3(nsb), dim3(256), 0, st, m->t, m->g, d_cnt + 32, d_blk, b_list.as<u32>(), list_cap, b_size.as<unsigned long long>(),
	                   b_off.as<unsigned long long>());
	const u32 first = std::max<u32>(1u, sa.min_depth + 1);  // blocks of nodes above min_depth
	const u32 l_tail = std::min<u32>(first + 2u, L);        // the two widest levels: a launch each; the rest: one workgroup
	for (u32 l = first; l < l_tail; ++l)
		hipLaunchKernelGGL(k_ser_sizes_dev, dim3(1024), dim3(256), 0, st, m->t, m->g, sa, b_list.as<u32>(), d_lv, l, D, b_size.as<u64>());
	const unsigned long long cap = bound;
	// (the narrow levels: both passes in one launch when they hold few enough blocks for its LDS -- as the previous
	// serialisation of this map found; should the map have outgrown that since, the kernel says so and the long way is taken)
	if (m->ser_tail_blocks <= UFO_SER_TAIL_MAX - 64u && m->ser_tail_first == l_tail && ncap <= (1ull << 31)) {
		// (what the one workgroup needs to know about the children of these levels' blocks is looked up by the whole chip first)
		DevBuf& b_tail = m->b_ser[5];
		HIP_TRY(b_tail.reserve((size_t)UFO_SER_TAIL_MAX * 8u * 4u * 2u));
		u32* gcw = b_tail.as<u32>();
		u32* gcs = gcw + (size_t)UFO_SER_TAIL_MAX * 8u;
		hipLaunchKernelGGL(k_ser_tail_prep, dim3(UFO_SER_TAIL_MAX * 8u / 256u), dim3(256), 0, st, m->t, m->g, sa, b_list.as<u32>(), d_lv, l_tail, L, D, b_size.as<u64>(), gcw, gcs);
		hipLaunchKernelGGL(k_ser_tail_dev, dim3(1), dim3(1024), 0, st, m->t, m->g, b_list.as<u32>(), d_lv, l_tail, L, D, gcw, gcs, b_off.as<u64>(), b_out.as<uint8_t>(), d_total, cap);
	} else {
		hipLaunchKernelGGL(k_ser_sizes_tail_dev, dim3(1), dim3(1024), 0, st, m->t, m->g, sa, b_list.as<u32>(), d_lv, l_tail, L, D, b_size.as<u64>(), d_total);
		hipLaunchKernelGGL(k_ser_write_tail_dev, dim3(1), dim3(1024), 0, st, m->t, m->g, sa, b_list.as<u32>(), d_lv, L, l_tail, D, b_size.as<u64>(), b_off.as<u64>(),
		                   b_out.as<uint8_t>(), d_total, cap);
	}
	for (u32 l = l_tail; l-- > first;)
		hipLaunchKernelGGL(k_ser_write_dev, dim3(1024), dim3(256), 0, st, m->t, m->g, sa, b_list.as<u32>(), d_lv, l, D, b_size.as<u64>(), b_off.as<u64>(),
		                   b_out.as<uint8_t>(), d_total, cap);
	hipLaunchKernelGGL(k_ser_copy_out, dim3(128), dim3(256), 0, st, b_out.as<uint4>(), d_total, cap, reinterpret_cast<uint4*>(m->h_out),
	                   const_cast<unsigned long long*>(h_total), d_lv, l_tail, L);
	HIP_TRY(hipGetLastError());
	const auto t1 = std::chrono::steady_clock::now();
	HIP_TRY(hipStreamSynchronize(st));
	const auto t2 = std::chrono::steady_clock::now();
	const u64 total = *h_total;
	m->ser_tail_blocks = (u32)std::min<unsigned long long>((unsigned long long)h_total[1], 0xFFFFFFFFull);
	m->ser_tail_first = l_tail;
	if (trace) {
		auto us = [](auto a, auto b) { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() * 1e-3; };
		fprintf(stderr, "[ufomap] serialise: enqueue %.1f us, wait %.1f us, %llu bytes, table %llu slots, %llu used, %u blocks in the narrow levels (from level %u)\n", us(t0, t1), us(t1, t2),
		        (unsigned long long)total, (unsigned long long)m->t.mask + 1, (unsigned long long)m->used_est, m->ser_tail_blocks, m->ser_tail_first);
	}
	if (0 == total || total > cap) return 1;  // the root is a leaf / the narrow levels outgrew the one-launch form / more than the bound: the long way
	if (total > 0x7FFFFFFFull) return fail(UFOMAP_ERR_CAPACITY, "map byte stream exceeds 2^31 bytes (the reference's size field is an int)");
	if (in_place) *in_place = (size_t)total;
	else data.assign(m->h_out, m->h_out + total);
	return UFOMAP_OK;
}

// the node stream of writeNodes (occupancy_map_base.h:1457-1533) for the whole map or the part inside a bounding volume
int serialiseNodes(ufomap_map* m, const SerArgs& sa, std::vector<uint8_t>& data, size_t* in_place = nullptr)
{
	data.clear();
	if (in_place) *in_place = 0;
	m->cs = m->stream;
	const u32 D = m->g.color ? 7u : 4u;
	const u32 L = m->g.L;
	if (sa.has_bv) {
		// the root's box against the volume (OMB:1461-1467): nothing is written, not even the root byte
		const double h = m->g.hs[L];
		for (int k = 0; k < 3; ++k) {
			const double min1 = sa.vc[k] - sa.vh[k], max1 = sa.vc[k] + sa.vh[k], min2 = 0.0 - h, max2 = 0.0 + h;
			if (!(min1 <= max2) || !(min2 <= max1)) return UFOMAP_OK;
		}
	}
	{
		const int src = serialiseNodesShort(m, sa, data, in_place);
		if (src <= 0) return src;
		data.clear();
		if (in_place) *in_place = 0;
	}
	// (scratch kept with the map: a publish per scan must not pay five allocations)
	DevBuf &b_cnt = m->b_ser[0], &b_list = m->b_ser[1], &b_size = m->b_ser[2], &b_off = m->b_ser[3], &b_out = m->b_ser[4];
	if (!m->h_ser) HIP_TRY(hipHostMalloc((void**)&m->h_ser, 512));
	u32* h_cnt = reinterpret_cast<u32*>(m->h_ser);                                    // [32] live blocks per level
	MapRoot* h_root = reinterpret_cast<MapRoot*>(m->h_ser + 128);
	unsigned long long* h_total = reinterpret_cast<unsigned long long*>(m->h_ser + 256);
	u32 h_off[32] = {0};
	HIP_TRY(b_cnt.reserve(UFO_SER_CNT_BYTES));
	HIP_TRY(hipMemsetAsync(b_cnt.p, 0, 3 * 32 * 4 + 16, m->stream));
	u32* d_cnt = b_cnt.as<u32>();
	unsigned long long* d_total = reinterpret_cast<unsigned long long*>(d_cnt + 96);
	const u32 nsb = serBlocks((u64)m->t.mask + 1);
	u32* d_blk = d_cnt + UFO_SER_BLK_WORD;
	hipLaunchKernelGGL(k_ser_count, dim3(nsb), dim3(256), 0, m->stream, m->t, m->g, d_blk);
	// (the level counts for the host, the workgroups' places in the list for k_ser_collect; the list is sized from the counts: it fits)
	hipLaunchKernelGGL(k_ser_prefix, dim3(1), dim3(1024), 0, m->stream, d_cnt, reinterpret_cast<SerLevels*>(d_cnt + 100), 0xFFFFFFFFu, d_blk, nsb);
	HIP_TRY(hipMemcpyAsync(h_cnt, d_cnt, 32 * 4, hipMemcpyDeviceToHost, m->stream));
	HIP_TRY(hipMemcpyAsync(h_root, m->b_root.p, sizeof(MapRoot), hipMemcpyDeviceToHost, m->stream));
	HIP_TRY(hipStreamSynchronize(m->stream));
	u64 n_live = 0;
	for (u32 l = 0; l < 32; ++l) {
		h_off[l] = (u32)n_live;
		n_live += h_cnt[l];
	}
	const MapRoot root = *h_root;
	if (0 == h_cnt[L] || L <= sa.min_depth) {
		// the root is (written as) a leaf: children byte 0, then the root's payload (occupancy_map_base.h:1469-1478)
		data.resize(1 + D);
		data[0] = 0;
		memcpy(&data[1], &root.occ, 4);
		if (D > 4) {
			data[5] = (uint8_t)root.rgb;
			data[6] = (uint8_t)(root.rgb >> 8);
			data[7] = (uint8_t)(root.rgb >> 16);
		}
		return UFOMAP_OK;
	}
	const u32 first = std::max<u32>(1u, sa.min_depth + 1);  // blocks of nodes above min_depth
	const size_t ncap = (size_t)m->t.mask + 1;
	HIP_TRY(b_list.reserve(std::max<u64>(n_live, 1) * 4));
	HIP_TRY(b_size.reserve(ncap * 8));
	HIP_TRY(b_off.reserve(ncap * 8));
	SerLevels lv{};
	for (u32 l = 0; l < 32; ++l) {
		lv.off[l] = h_off[l];
		lv.cnt[l] = h_cnt[l];
	}
	HIP_TRY(hipMemcpyAsync(d_cnt + 32, h_off, 32 * 4, hipMemcpyHostToDevice, m->stream));  // (h_off: read by the copy before this function returns -- it synchronises below)
	HIP_TRY(hipMemsetAsync(b_off.p, 0xFF, ncap * 8, m->stream));
	hipLaunchKernelGGL(k_ser_collect, dim3(nsb), dim3(256), 0, m->stream, m->t, m->g, d_cnt + 32, d_blk, b_list.as<u32>(),
	                   (u32)std::min<u64>(std::max<u64>(n_live, 1), 0xFFFFFFFFull));
	// wide levels: a launch each; from the first level of at most 2048 blocks up to the root: ONE workgroup, a barrier per level
	u32 l_tail = L;
	while (l_tail > first && h_cnt[l_tail - 1] <= 2048u) --l_tail;
	for (u32 l = first; l < l_tail; ++l)
		if (h_cnt[l])
			hipLaunchKernelGGL(k_ser_sizes, gridFor((u64)h_cnt[l] * 8u), dim3(256), 0, m->stream, m->t, m->g, sa, b_list.as<u32>() + h_off[l], h_cnt[l], l, D,
			                   b_size.as<u64>());
	hipLaunchKernelGGL(k_ser_sizes_tail, dim3(1), dim3(1024), 0, m->stream, m->t, m->g, sa, b_list.as<u32>(), lv, l_tail, L, D, b_size.as<u64>(), d_total);
	HIP_TRY(hipMemcpyAsync(h_total, d_total, 8, hipMemcpyDeviceToHost, m->stream));
	HIP_TRY(hipStreamSynchronize(m->stream));
	const u64 total = *h_total;  // 0xFF byte + subtree of the root block
	if (total > 0x7FFFFFFFull) return fail(UFOMAP_ERR_CAPACITY, "map byte stream exceeds 2^31 bytes (the reference's size field is an int)");
	HIP_TRY(b_out.reserve(total));
	hipLaunchKernelGGL(k_ser_write_tail, dim3(1), dim3(1024), 0, m->stream, m->t, m->g, sa, b_list.as<u32>(), lv, L, l_tail, D, b_size.as<u64>(), b_off.as<u64>(),
	                   b_out.as<uint8_t>());
	for (u32 l = l_tail; l-- > first;)
		if (h_cnt[l])
			hipLaunchKernelGGL(k_ser_write, gridFor((u64)h_cnt[l] * 8u), dim3(256), 0, m->stream, m->t, m->g, sa, b_list.as<u32>() + h_off[l], h_cnt[l], l, D,
			                   b_size.as<u64>(), b_off.as<u64>(), b_out.as<uint8_t>());
	data.resize(total);
	HIP_TRY(hipMemcpyAsync(data.data(), b_out.p, total, hipMemcpyDeviceToHost, m->stream));
	HIP_TRY(hipStreamSynchronize(m->stream));
	return UFOMAP_OK;
}
}  // namespace
}  // extern "C++"

size_t ufomap_map_write_ex(ufomap_map* m, const double* aabb_center, const double* aabb_half, int compress, unsigned min_depth,
                           int compression_acceleration_level, int compression_level, int header, uint8_t* buf, size_t cap,
                           long long* uncompressed_size)
{
	if (!m || ((nullptr == aabb_center) != (nullptr == aabb_half))) {
		fail(UFOMAP_ERR_INVALID, "null map / half a bounding volume");
		return (size_t)-1;
	}
	if (ufomap_map_wait(m) < 0) return (size_t)-1;
	SerArgs sa{};
	sa.has_bv = aabb_center ? 1u : 0u;
	for (int k = 0; k < 3 && aabb_center; ++k) {
		sa.vc[k] = aabb_center[k];
		sa.vh[k] = aabb_half[k];
	}
	sa.min_depth = min_depth;
	std::vector<uint8_t> data;
	size_t pinned_n = 0;
	if (serialiseNodes(m, sa, data, &pinned_n)) return (size_t)-1;
	// (the stream: in the handle's pinned buffer when the device-side serialiser produced it, else in `data`)
	const uint8_t* src = pinned_n ? m->h_out : data.data();
	size_t src_n = pinned_n ? pinned_n : data.size();
	const long long usize = (long long)src_n;
	if (uncompressed_size) *uncompressed_size = usize;
	if (compress) {
		// compressData (octree.h:1430-1458)
		const Lz4& z = lz4();
		if (!z.ok) {
			fail(UFOMAP_ERR_UNSUPPORTED, "liblz4 could not be loaded: compressed output is not available");
			return (size_t)-1;
		}
		const int bound = z.bound((int)src_n);
		std::vector<uint8_t> comp((size_t)std::max(bound, 1));
		const int n = 0 >= compression_level
		                  ? z.fast(reinterpret_cast<const char*>(src), reinterpret_cast<char*>(comp.data()), (int)src_n, bound, compression_acceleration_level)
		                  : z.hc(reinterpret_cast<const char*>(src), reinterpret_cast<char*>(comp.data()), (int)src_n, bound, compression_level);
		if (n < 0) {
			fail(UFOMAP_ERR_DEVICE, "LZ4 compression failed");
			return (size_t)-1;
		}
		comp.resize((size_t)n);
		data.swap(comp);
		src = data.data();
		src_n = data.size();
	}
	std::string h;
	if (header) {
		// text header exactly as Octree::write prints it (octree.h:850-861)
		std::ostringstream hd;
		hd << "# UFOMap file";
		hd << "\n# (feel free to add / change comments, but leave the first line as it is!)\n#\n";
		hd << "version " << "1.0.0" << std::endl;
		hd << "id " << (m->g.color ? "occupancy_map_color" : "occupancy_map") << std::endl;
		hd << "resolution " << m->g.res << std::endl;
		hd << "depth_levels " << m->g.L << std::endl;
		hd << "compressed " << (compress ? true : false) << std::endl;
		hd << "uncompressed_data_size " << (int)usize << std::endl;
		hd << "data" << std::endl;
		h = hd.str();
	}
	const size_t total = h.size() + src_n;
	if (buf && cap >= total) {
		memcpy(buf, h.data(), h.size());
		if (src_n) memcpy(buf + h.size(), src, src_n);
	}
	return total;
}

extern "C++" {
namespace
{
// one pass over a node stream (readNodesRecurs, occupancy_map_base.h:1405-1455): a ReadRec per node with children
struct StreamParser {
	const uint8_t* p;
	size_t n, pos = 0;
	bool bad = false;
	u32 D;
	const MapGeom* g;
	bool has_bv;
	double vc[3], vh[3];
	std::vector<ReadRec> recs[24];  // by level (= depth of the node)
	size_t n_recs = 0;              // a record (~100 bytes) per node with children: bounded while parsing, not afterwards
	static constexpr size_t kMaxRecs = 1u << 26;

	bool inside(const double c[3], double h) const
	{
		if (!has_bv) return true;
		for (int k = 0; k < 3; ++k) {
			const double min1 = vc[k] - vh[k], max1 = vc[k] + vh[k], min2 = c[k] - h, max2 = c[k] + h;
			if (!(min1 <= max2) || !(min2 <= max1)) return false;
		}
		return true;
	}
	void leaf(float* v, u32* rgb)
	{
		if (pos + D > n) {
			bad = true;
			*v = 0;
			*rgb = 0;
			return;
		}
		memcpy(v, p + pos, 4);
		*rgb = D > 4 ? ((u32)p[pos + 4] | ((u32)p[pos + 5] << 8) | ((u32)p[pos + 6] << 16)) : 0u;
		pos += D;
	}
	void node(u64 lk, u32 cd, const double c[3], u32 parent)
	{
		if (bad || pos >= n) {
			bad = true;
			return;
		}
		const uint8_t children = p[pos++];
		if (++n_recs > kMaxRecs) {
			bad = true;  // (more inner nodes than a map this library can hold: not a stream it wrote)
			return;
		}
		const u32 mine = (u32)recs[cd].size();
		ReadRec r{};
		r.lk = lk;
		r.parent = parent;
		r.slot = NONE;
		recs[cd].push_back(r);
		const double chs = g->hs[cd - 1];
		u32 set_mask = 0, inner_mask = 0;
		for (u32 i = 0; i < 8 && !bad; ++i) {
			const double cc[3] = {c[0] + ((i & 1) ? chs : -chs), c[1] + ((i & 2) ? chs : -chs), c[2] + ((i & 4) ? chs : -chs)};
			if (!inside(cc, chs)) continue;
			if ((children >> i) & 1u) {
				inner_mask |= 1u << i;
				if (2 == cd) {
					// a depth-1 child: its 8 voxels follow without a mask byte (OMB:1427-1438)
					ReadRec cr{};
					cr.lk = (lk << 3) | (u64)i;
					cr.parent = mine;
					cr.slot = NONE;
					const double ghs = g->hs[0];
					for (u32 j = 0; j < 8 && !bad; ++j) {
						const double gc[3] = {cc[0] + ((j & 1) ? ghs : -ghs), cc[1] + ((j & 2) ? ghs : -ghs), cc[2] + ((j & 4) ? ghs : -ghs)};
						if (!inside(gc, ghs)) continue;
						cr.set_mask |= 1u << j;
						leaf(&cr.val[j], &cr.rgb[j]);
					}
					if (++n_recs > kMaxRecs) bad = true;
					recs[1].push_back(cr);
				} else {
					node((lk << 3) | (u64)i, cd - 1, cc, mine);
				}
			} else {
				float v;
				u32 col;
				leaf(&v, &col);
				set_mask |= 1u << i;
				recs[cd][mine].val[i] = v;
				recs[cd][mine].rgb[i] = col;
			}
		}
		recs[cd][mine].set_mask = set_mask;
		recs[cd][mine].inner_mask = inner_mask;
	}
};

// readNodes (occupancy_map_base.h:1379-1403) on an uncompressed node stream
int readNodes(ufomap_map* m, const uint8_t* data, size_t n, const double* aabb_center, const double* aabb_half)
{
	const u32 L = m->g.L;
	const u32 D = m->g.color ? 7u : 4u;
	if (aabb_center) {
		const double h = m->g.hs[L];
		for (int k = 0; k < 3; ++k) {
			const double min1 = aabb_center[k] - aabb_half[k], max1 = aabb_center[k] + aabb_half[k], min2 = 0.0 - h, max2 = 0.0 + h;
			if (!(min1 <= max2) || !(min2 <= max1)) return UFOMAP_OK;  // no node intersects
		}
	}
	if (n < 1) return fail(UFOMAP_ERR_INVALID, "empty node stream");
	m->cs = m->stream;
	m->args = ScanArgs{};
	ScanCtl init;
	memset(&init, 0, sizeof(init));
	for (int k = 0; k < 3; ++k) {
		init.aabb_min[k] = ~0ull;
		init.aabb_max[k] = 0ull;
	}
	*m->h_ctl = init;
	HIP_TRY(hipMemcpyAsync(m->b_ctl.p, m->h_ctl, sizeof(ScanCtl), hipMemcpyHostToDevice, m->stream));
	m->ctl_clean = false;  // (the set's device control block no longer holds the fast path's start state)
	ScanCtl* ctl = m->b_ctl.as<ScanCtl>();
	m->scan_id += 1;
	if (0 == data[0]) {
		// the stream's root is a leaf: deleteChildren(root), readData, updateNode (occupancy_map_base.h:1394-1399)
		if (n < 1 + (size_t)D) return fail(UFOMAP_ERR_INVALID, "truncated node stream");
		float v;
		memcpy(&v, data + 1, 4);
		const u32 col = D > 4 ? ((u32)data[5] | ((u32)data[6] << 8) | ((u32)data[7] << 16)) : 0u;
		hipLaunchKernelGGL(k_vol_root, gridFor((u64)m->t.mask + 1), dim3(256), 0, m->cs, m->t, m->g, v);
		HIP_TRY(hipMemcpyAsync(reinterpret_cast<char*>(m->b_root.p) + offsetof(MapRoot, rgb), &col, 4, hipMemcpyHostToDevice, m->stream));
		HIP_TRY(hipStreamSynchronize(m->stream));
		return UFOMAP_OK;
	}
	StreamParser sp;
	sp.p = data;
	sp.n = n;
	sp.pos = 1;  // behind the root's children byte
	sp.D = D;
	sp.g = &m->g;
	sp.has_bv = nullptr != aabb_center;
	for (int k = 0; k < 3 && aabb_center; ++k) {
		sp.vc[k] = aabb_center[k];
		sp.vh[k] = aabb_half[k];
	}
	const double c0[3] = {0.0, 0.0, 0.0};
	sp.node(1, L, c0, NONE);
	if (sp.bad) return fail(UFOMAP_ERR_INVALID, "truncated node stream");
	// records level by level, root first; parents are indices into the level above
	u32 off[24] = {0};
	u64 total = 0;
	for (u32 l = L; l >= 1; --l) {
		off[l] = (u32)total;
		total += sp.recs[l].size();
	}
	if (total > 0x7FFFFFF0ull) return fail(UFOMAP_ERR_CAPACITY, "node stream too large");
	std::vector<ReadRec> all;
	all.reserve((size_t)total);
	for (u32 l = L; l >= 1; --l)
		for (ReadRec r : sp.recs[l]) {
			if (r.parent != NONE) r.parent += off[l + 1];
			all.push_back(r);
		}
	// every record may create a block
	{
		// (a record of level l is a node block of level l: tile groups for the level-3 records, the first region above)
		Need need;
		need.blocks = total;
		for (u32 l = L; l >= 1; --l) {
			if (L < 4 || l >= 4) need.upper += sp.recs[l].size();
			else if (3 == l) need.groups += sp.recs[l].size();
		}
		if (!tableTakes(m, need)) {
			int rc = growFor(m, need);
			if (rc) return rc;
		}
	}
	const u32 kcap = (u32)std::min<u64>(m->used_est + 8, 0x7FFFFFFFull);
	HIP_TRY(m->b_crec.reserve((size_t)total * sizeof(ReadRec)));
	HIP_TRY(m->b_dlist.reserve((size_t)kcap * 4));
	HIP_TRY(hipMemcpyAsync(m->b_crec.p, all.data(), (size_t)total * sizeof(ReadRec), hipMemcpyHostToDevice, m->stream));
	ReadRec* rec = m->b_crec.as<ReadRec>();
	u32* kill = m->b_dlist.as<u32>();
	for (u32 l = L; l >= 1; --l) {
		const u32 cnt = (u32)sp.recs[l].size();
		if (cnt) hipLaunchKernelGGL(k_read_down, gridFor(cnt, 256, 4096), dim3(256), 0, m->cs, m->t, m->g, rec, off[l], off[l] + cnt, l, kill, kcap, m->scan_id, ctl);
	}
	hipLaunchKernelGGL(k_vol_kill_mark, dim3(1), dim3(1), 0, m->cs, ctl, 0u);
	for (u32 l = 0; l + 1 < L; ++l) {
		hipLaunchKernelGGL(k_vol_kill, gridFor(std::max<u64>(kcap, 256), 256, 4096), dim3(256), 0, m->cs, m->t, kill, kcap, ctl);
		hipLaunchKernelGGL(k_vol_kill_mark, dim3(1), dim3(1), 0, m->cs, ctl, 1u);
	}
	for (u32 l = 1; l <= L; ++l) {
		const u32 cnt = (u32)sp.recs[l].size();
		if (cnt) hipLaunchKernelGGL(k_read_up, gridFor(cnt, 256, 4096), dim3(256), 0, m->cs, m->t, m->g, rec, off[l], off[l] + cnt, l, ctl);
	}
	HIP_TRY(hipGetLastError());
	m->pending = true;
	HIP_TRY(hipStreamSynchronize(m->stream));  // (`all` is pageable: the upload has completed before it goes out of scope)
	return finishPending(m);
}
}  // namespace
}  // extern "C++"

int ufomap_map_read_data(ufomap_map* m, const uint8_t* data, size_t n, const double* aabb_center, const double* aabb_half,
                         double resolution, unsigned depth_levels, int uncompressed_data_size, int compressed)
{
	if (!m || (n && !data) || ((nullptr == aabb_center) != (nullptr == aabb_half))) return fail(UFOMAP_ERR_INVALID, "null argument");
	HIP_TRY(hipSetDevice(m->device));
	int rc = ufomap_map_wait(m);
	if (rc) return rc;
	if (m->g.res != resolution || m->g.L != depth_levels) {  // readData (octree.h:760-762)
		rc = ufomap_map_clear_to(m, resolution, depth_levels);
		if (rc) return rc;
	}
	{
		// used_est may be stale after a clear
		const int frc = refreshFill(m);
		if (frc) return frc;
	}
	if (compressed) {
		// decompressData (octree.h:1460-1486)
		const Lz4& z = lz4();
		if (!z.ok) return fail(UFOMAP_ERR_UNSUPPORTED, "liblz4 could not be loaded: compressed input is not available");
		if (uncompressed_data_size < 0) return fail(UFOMAP_ERR_INVALID, "negative uncompressed_data_size");
		if (n > 0x7FFFFFFFull) return fail(UFOMAP_ERR_INVALID, "compressed stream longer than 2^31 bytes (LZ4's size arguments are ints)");
		// (LZ4 cannot expand a block by more than a factor of 255: a header that claims more is not believed -- it would
		// only size an allocation)
		if ((u64)uncompressed_data_size > 255ull * (u64)n + 64ull) return fail(UFOMAP_ERR_INVALID, "uncompressed_data_size is impossible for a compressed stream of this length");
		std::vector<uint8_t> raw((size_t)std::max(uncompressed_data_size, 1));
		const int got = z.safe(reinterpret_cast<const char*>(data), reinterpret_cast<char*>(raw.data()), (int)n, uncompressed_data_size);
		if (got < 0) return fail(UFOMAP_ERR_INVALID, "LZ4 decompression failed");
		return readNodes(m, raw.data(), (size_t)got, aabb_center, aabb_half);
	}
	return readNodes(m, data, n, aabb_center, aabb_half);
}

int ufomap_map_read(ufomap_map* m, const uint8_t* buf, size_t n, double* resolution, unsigned* depth_levels)
{
	if (!m || !buf) return fail(UFOMAP_ERR_INVALID, "null argument");
	// Octree::read / readHeader (octree.h:701-735, 640-688): first line, then "token value" lines up to "data"
	static const char kHeader[] = "# UFOMap file";
	if (n < sizeof(kHeader) - 1 || 0 != memcmp(buf, kHeader, sizeof(kHeader) - 1)) return fail(UFOMAP_ERR_INVALID, "not a UFOMap file");
	size_t pos = 0;
	auto line = [&](std::string* out) {
		if (pos >= n) return false;
		size_t e = pos;
		while (e < n && buf[e] != '\n') ++e;
		out->assign(reinterpret_cast<const char*>(buf) + pos, e - pos);
		pos = std::min(n, e + 1);
		return true;
	};
	std::string ln, id;
	double res = 0;
	unsigned levels = 0;
	int compressed = 0, usize = 0;
	bool got_data = false;
	(void)line(&ln);  // the file header line
	while (line(&ln)) {
		std::istringstream is(ln);
		std::string tok;
		if (!(is >> tok)) continue;
		if ("data" == tok) {
			got_data = true;
			break;
		}
		if ('#' == tok[0]) continue;
		if ("id" == tok) is >> id;
		else if ("resolution" == tok) is >> res;
		else if ("depth_levels" == tok) is >> levels;
		else if ("compressed" == tok) is >> compressed;
		else if ("uncompressed_data_size" == tok) is >> usize;
	}
	if (!got_data || !(res > 0) || levels < 2 || levels > 21) return fail(UFOMAP_ERR_INVALID, "malformed UFOMap header");
	if (id != (m->g.color ? "occupancy_map_color" : "occupancy_map"))
		return fail(UFOMAP_ERR_INVALID, "file holds a '" + id + "', the map is a '" + (m->g.color ? "occupancy_map_color" : "occupancy_map") + "'");
	int rc = ufomap_map_read_data(m, buf + pos, n - pos, nullptr, nullptr, res, levels, usize, compressed);
	if (rc) return rc;
	if (resolution) *resolution = res;
	if (depth_levels) *depth_levels = levels;
	return UFOMAP_OK;
}
