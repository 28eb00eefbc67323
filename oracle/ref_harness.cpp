/*
 * ref_harness.cpp -- C-ABI shim around the UNMODIFIED reference implementation.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_abi.h).  This file is the only source of ours that goes
 * into oracle/_ref/libufo_ref.so; everything else is compiled where it lies under
 * /root/reference/ufomap (oracle/Makefile, target `ref`).  No reference source is copied here:
 * the harness only *calls* the reference's public API
 *   ufo::map::OccupancyMap / OccupancyMapColor            (occupancy_map.h:55, occupancy_map_color.h:56)
 *   insertPointCloud / insertPointCloudDiscrete           (occupancy_map_base.h:270, 340;
 *                                                          occupancy_map_color.h:93, 177)
 * and walks the resulting tree through the protected accessors a subclass may use
 *   getRoot / isLeaf / getChild                           (octree.h:948-952, 1089-1127)
 * so that the dump sees exactly what the reference's own leaf iterator sees (is_leaf flag).
 *
 * Do NOT read point queries (getOccupancy(code) ...) from the reference: they are off by one level
 * (SURVEY.md section 4).
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <sstream>
#include <tuple>
#include <vector>

#include <ufo/map/occupancy_map.h>
#include <ufo/map/occupancy_map_color.h>

#include "oracle_abi.h"

namespace
{
struct Rec {
	uint64_t code;  // shifted: code >> 3*depth
	uint8_t depth;
	float occ;
	uint8_t flags;
	uint8_t rgb[3];
};

inline bool recLess(Rec const& a, Rec const& b)
{
	return a.depth != b.depth ? a.depth < b.depth : a.code < b.code;
}

template <class MAP, bool COLOR>
struct Probe : MAP {
	using MAP::MAP;
	using Inner = typename MAP::INNER_NODE;
	using Leaf = typename MAP::LEAF_NODE;

	void walk(Leaf const& node, unsigned depth, uint64_t prefix, bool include_unknown,
	          std::vector<Rec>* leaves, std::vector<Rec>* inner) const
	{
		bool leaf = (0 == depth) || MAP::isLeaf(static_cast<Inner const&>(node));
		Rec r;
		r.code = prefix;
		r.depth = static_cast<uint8_t>(depth);
		r.occ = node.value.occupancy;
		r.flags = 0;
		r.rgb[0] = r.rgb[1] = r.rgb[2] = 0;
		if constexpr (COLOR) {
			r.rgb[0] = node.value.color.r;
			r.rgb[1] = node.value.color.g;
			r.rgb[2] = node.value.color.b;
		}
		if (leaf) {
			if (leaves) {
				// unknown <=> free_thres <= v <= occupied_thres (occupancy_map_base.h:932-936)
				bool unknown = MAP::isUnknown(node);
				if (include_unknown || !unknown) {
					leaves->push_back(r);
				}
			}
			return;
		}
		Inner const& in = static_cast<Inner const&>(node);
		if (inner) {
			r.flags = (in.contains_free ? 1 : 0) | (in.contains_unknown ? 2 : 0);
			inner->push_back(r);
		}
		for (unsigned i = 0; i < 8; ++i) {
			walk(MAP::getChild(in, depth - 1, i), depth - 1, (prefix << 3) | i, include_unknown,
			     leaves, inner);
		}
	}

	// the reference's own query functions on one coordinate (occupancy_map_base.h:599-728) + the raw logit of the
	// node Octree::getNode returns
	void query(double x, double y, double z, unsigned depth, float* logodds, uint8_t* state) const
	{
		ufo::map::Point3 p(x, y, z);
		auto code = MAP::toCode(p, depth);
		*logodds = MAP::getNode(code).first->value.occupancy;
		uint8_t st = 0;
		switch (MAP::getState(p, depth)) {
			case ufo::map::OccupancyState::occupied: st = 1; break;
			case ufo::map::OccupancyState::free: st = 2; break;
			default: st = 4; break;
		}
		if (MAP::containsFree(p, depth)) st |= 8;
		if (MAP::containsUnknown(p, depth)) st |= 16;
		*state = st;
	}

	void dump(bool include_unknown, std::vector<Rec>* leaves, std::vector<Rec>* inner) const
	{
		walk(MAP::getRoot(), MAP::getTreeDepthLevels(), 0, include_unknown, leaves, inner);
		if (leaves) std::sort(leaves->begin(), leaves->end(), recLess);
		if (inner) std::sort(inner->begin(), inner->end(), recLess);
	}
};

using ProbeOcc = Probe<ufo::map::OccupancyMap, false>;
using ProbeCol = Probe<ufo::map::OccupancyMapColor, true>;
}  // namespace

struct ufo_oracle_map {
	std::unique_ptr<ProbeOcc> occ;
	std::unique_ptr<ProbeCol> col;
};

extern "C" {

ufo_oracle_map* ufo_oracle_create(double resolution, unsigned depth_levels, int automatic_pruning,
                                  double occupied_thres, double free_thres, double prob_hit,
                                  double prob_miss, double clamp_min, double clamp_max, int color)
{
	try {
		auto* m = new ufo_oracle_map;
		if (color) {
			m->col.reset(new ProbeCol(resolution, depth_levels, 0 != automatic_pruning,
			                          occupied_thres, free_thres, prob_hit, prob_miss, clamp_min,
			                          clamp_max));
			m->col->enableMinMaxChangeDetection(true);
		} else {
			m->occ.reset(new ProbeOcc(resolution, depth_levels, 0 != automatic_pruning,
			                          occupied_thres, free_thres, prob_hit, prob_miss, clamp_min,
			                          clamp_max));
			m->occ->enableMinMaxChangeDetection(true);
		}
		return m;
	} catch (...) {
		return nullptr;
	}
}

void ufo_oracle_destroy(ufo_oracle_map* m) { delete m; }

int ufo_oracle_insert(ufo_oracle_map* m, const double origin[3], const double* xyz,
                      const uint8_t* rgb, size_t n, double max_range, unsigned depth, int discrete,
                      int simple_ray_casting, unsigned early_stopping)
{
	ufo::map::Point3 o(origin[0], origin[1], origin[2]);
	if (rgb) {
		if (!m->col) return -1;
		if (!discrete) return -2;  // OccupancyMapColor::insertPointCloud<PointCloudColor> does not compile
		ufo::map::PointCloudColor cloud;
		cloud.reserve(n);
		for (size_t i = 0; i < n; ++i) {
			cloud.push_back(ufo::map::Point3Color(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2],
			                                      rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]));
		}
		m->col->insertPointCloudDiscrete(o, cloud, max_range, depth, 0 != simple_ray_casting,
		                                 early_stopping, false);
		return 0;
	}
	ufo::map::PointCloud cloud;
	cloud.reserve(n);
	for (size_t i = 0; i < n; ++i) {
		cloud.push_back(ufo::map::Point3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
	}
	if (m->col) {
		if (discrete) {
			m->col->insertPointCloudDiscrete(o, cloud, max_range, depth, 0 != simple_ray_casting,
			                                 early_stopping, false);
		} else {
			m->col->insertPointCloud(o, cloud, max_range, depth, 0 != simple_ray_casting,
			                         early_stopping, false);
		}
	} else {
		if (discrete) {
			m->occ->insertPointCloudDiscrete(o, cloud, max_range, depth, 0 != simple_ray_casting,
			                                 early_stopping, false);
		} else {
			m->occ->insertPointCloud(o, cloud, max_range, depth, 0 != simple_ray_casting,
			                         early_stopping, false);
		}
	}
	return 0;
}

static size_t copyOut(std::vector<Rec> const& v, uint64_t* codes, uint8_t* depths, float* logodds,
                      uint8_t* flags, uint8_t* rgb, size_t cap)
{
	size_t n = std::min(v.size(), cap);
	for (size_t i = 0; i < n; ++i) {
		if (codes) codes[i] = v[i].code;
		if (depths) depths[i] = v[i].depth;
		if (logodds) logodds[i] = v[i].occ;
		if (flags) flags[i] = v[i].flags;
		if (rgb) std::memcpy(rgb + 3 * i, v[i].rgb, 3);
	}
	return v.size();
}

size_t ufo_oracle_export_leaves(const ufo_oracle_map* m, int include_unknown, uint64_t* codes,
                                uint8_t* depths, float* logodds, uint8_t* rgb, size_t cap)
{
	std::vector<Rec> leaves;
	if (m->col) {
		m->col->dump(0 != include_unknown, &leaves, nullptr);
	} else {
		m->occ->dump(0 != include_unknown, &leaves, nullptr);
	}
	return copyOut(leaves, codes, depths, logodds, nullptr, rgb, cap);
}

size_t ufo_oracle_export_inner(const ufo_oracle_map* m, uint64_t* codes, uint8_t* depths,
                               float* logodds, uint8_t* flags, uint8_t* rgb, size_t cap)
{
	std::vector<Rec> inner;
	if (m->col) {
		m->col->dump(true, nullptr, &inner);
	} else {
		m->occ->dump(true, nullptr, &inner);
	}
	return copyOut(inner, codes, depths, logodds, flags, rgb, cap);
}

size_t ufo_oracle_write(const ufo_oracle_map* m, uint8_t* buf, size_t cap)
{
	std::stringstream ss(std::ios_base::in | std::ios_base::out | std::ios_base::binary);
	bool ok = m->col ? m->col->write(ss, false) : m->occ->write(ss, false);
	if (!ok) return (size_t)-1;
	std::string const bytes = ss.str();
	if (buf && cap >= bytes.size()) std::memcpy(buf, bytes.data(), bytes.size());
	return bytes.size();
}

int ufo_oracle_minmax_change(const ufo_oracle_map* m, double mn[3], double mx[3])
{
	ufo::map::Point3 a = m->col ? m->col->minChange() : m->occ->minChange();
	ufo::map::Point3 b = m->col ? m->col->maxChange() : m->occ->maxChange();
	for (int i = 0; i < 3; ++i) {
		mn[i] = a[i];
		mx[i] = b[i];
	}
	return 0;
}

size_t ufo_oracle_last_hits(const ufo_oracle_map*, uint64_t*, size_t) { return (size_t)-1; }
size_t ufo_oracle_last_rays(const ufo_oracle_map*, double*, size_t) { return (size_t)-1; }
size_t ufo_oracle_last_misses(const ufo_oracle_map*, uint64_t*, size_t) { return (size_t)-1; }
uint64_t ufo_oracle_last_steps(const ufo_oracle_map*) { return (uint64_t)-1; }
uint64_t ufo_oracle_last_oob(const ufo_oracle_map*) { return (uint64_t)-1; }

void ufo_oracle_query(const ufo_oracle_map* m, const double* xyz, size_t n, unsigned depth, float* logodds, uint8_t* state)
{
	for (size_t q = 0; q < n; ++q) {
		if (m->col)
			m->col->query(xyz[3 * q], xyz[3 * q + 1], xyz[3 * q + 2], depth, logodds + q, state + q);
		else
			m->occ->query(xyz[3 * q], xyz[3 * q + 1], xyz[3 * q + 2], depth, logodds + q, state + q);
	}
}

int ufo_oracle_set_value_volume(ufo_oracle_map* m, const double mn[3], const double mx[3], double occupancy_value, unsigned min_depth)
{
	// exactly what the server does (ufomap_mapping/src/server.cpp:152-155): AABB(min, max), then setValueVolume
	ufo::geometry::AABB aabb(ufo::geometry::Point(mn[0], mn[1], mn[2]), ufo::geometry::Point(mx[0], mx[1], mx[2]));
	if (m->col)
		m->col->setValueVolume(aabb, occupancy_value, min_depth);
	else
		m->occ->setValueVolume(aabb, occupancy_value, min_depth);
	return 0;
}
void ufo_oracle_clamping_thres(const ufo_oracle_map* m, double* thres_min, double* thres_max)
{
	*thres_min = m->col ? m->col->getClampingThresMin() : m->occ->getClampingThresMin();
	*thres_max = m->col ? m->col->getClampingThresMax() : m->occ->getClampingThresMax();
}

// The conversion loop of ufomap_ros (conversions.cpp:98-138) needs ROS message types and is restated here; the
// transform is the reference's own Pose6::transform on its own Point3Color.
size_t ufo_oracle_ingest(const uint8_t* data, size_t n, uint32_t step, int off_x, int off_y, int off_z, int off_r, int off_g,
                         int off_b, const double q[4], const double t[3], double* xyz_out, uint8_t* rgb_out)
{
	const ufo::math::Pose6 pose(ufo::math::Vector3(t[0], t[1], t[2]), ufo::math::Quaternion(q[0], q[1], q[2], q[3]));
	ufo::map::PointCloudColor cloud;
	for (size_t i = 0; i < n; ++i) {
		const uint8_t* rec = data + i * (size_t)step;
		float fx, fy, fz;
		memcpy(&fx, rec + off_x, 4);
		memcpy(&fy, rec + off_y, 4);
		memcpy(&fz, rec + off_z, 4);
		if (!std::isnan(fx) && !std::isnan(fy) && !std::isnan(fz)) {
			if (off_r >= 0)
				cloud.push_back(ufo::map::Point3Color(fx, fy, fz, rec[off_r], rec[off_g], rec[off_b]));
			else
				cloud.push_back(ufo::map::Point3Color(fx, fy, fz));
		}
	}
	cloud.transform(pose, false);
	size_t k = 0;
	for (auto const& p : cloud) {
		xyz_out[3 * k] = p.x();
		xyz_out[3 * k + 1] = p.y();
		xyz_out[3 * k + 2] = p.z();
		if (rgb_out) {
			rgb_out[3 * k] = p.getColor().r;
			rgb_out[3 * k + 1] = p.getColor().g;
			rgb_out[3 * k + 2] = p.getColor().b;
		}
		++k;
	}
	return k;
}

const char* ufo_oracle_kind(void) { return "reference"; }

}  // extern "C"

// ---- round 2: iterators, change detection, write / read with all arguments, sensor-model accessors --------------
namespace
{
template <class IT, class MAP>
size_t runIterator(MAP const& map, IT it, IT end, bool color, uint64_t* codes, uint8_t* depths, float* logodds, uint8_t* rgb, uint8_t* flags,
                   size_t cap)
{
	size_t n = 0;
	for (; it != end; ++it, ++n) {
		if (n >= cap) continue;
		const unsigned d = it.getDepth();
		if (codes) codes[n] = it.getCode().getCode() >> (3 * d);
		if (depths) depths[n] = (uint8_t)d;
		if (logodds) logodds[n] = it->occupancy;
		if (flags) flags[n] = (it.containsFree() ? 1 : 0) | (it.containsUnknown() ? 2 : 0) | (it.isLeaf() ? 4 : 0);
		if (rgb) {
			rgb[3 * n] = rgb[3 * n + 1] = rgb[3 * n + 2] = 0;
			if constexpr (std::is_same_v<MAP, ProbeCol>) {
				rgb[3 * n] = it->color.r;
				rgb[3 * n + 1] = it->color.g;
				rgb[3 * n + 2] = it->color.b;
			}
		}
	}
	(void)map;
	(void)color;
	return n;
}
template <class MAP>
size_t iterateMap(MAP const& map, const double* c, const double* h, bool o, bool f, bool u, bool contains, unsigned min_depth, bool only_leaves,
                  uint64_t* codes, uint8_t* depths, float* logodds, uint8_t* rgb, uint8_t* flags, size_t cap)
{
	ufo::geometry::BoundingVolume bv;
	if (c) {
		ufo::geometry::AABB a;
		a.center = ufo::geometry::Point(c[0], c[1], c[2]);
		a.half_size = ufo::geometry::Point(h[0], h[1], h[2]);
		bv.add(a);
	}
	if (only_leaves) return runIterator(map, map.beginLeaves(bv, o, f, u, contains, min_depth), map.endLeaves(), false, codes, depths, logodds, rgb, flags, cap);
	return runIterator(map, map.beginTree(bv, o, f, u, contains, min_depth), map.endTree(), false, codes, depths, logodds, rgb, flags, cap);
}
ufo::geometry::BoundingVolume makeBv(const double* c, const double* h)
{
	ufo::geometry::BoundingVolume bv;
	if (c) {
		ufo::geometry::AABB a;
		a.center = ufo::geometry::Point(c[0], c[1], c[2]);
		a.half_size = ufo::geometry::Point(h[0], h[1], h[2]);
		bv.add(a);
	}
	return bv;
}
}  // namespace

extern "C" {
size_t ufo_oracle_iterate(const ufo_oracle_map* m, const double* c, const double* h, int o, int f, int u, int contains, unsigned min_depth,
                          int only_leaves, uint64_t* codes, uint8_t* depths, float* logodds, uint8_t* rgb, uint8_t* flags, size_t cap)
{
	if (m->col) return iterateMap(*m->col, c, h, o, f, u, contains, min_depth, only_leaves, codes, depths, logodds, rgb, flags, cap);
	return iterateMap(*m->occ, c, h, o, f, u, contains, min_depth, only_leaves, codes, depths, logodds, rgb, flags, cap);
}
int ufo_oracle_enable_change_detection(ufo_oracle_map* m, int enable)
{
	if (m->col) m->col->enableChangeDetection(0 != enable);
	else m->occ->enableChangeDetection(0 != enable);
	return 0;
}
int ufo_oracle_reset_change_detection(ufo_oracle_map* m)
{
	if (m->col) m->col->resetChangeDetection();
	else m->occ->resetChangeDetection();
	return 0;
}
size_t ufo_oracle_changes(const ufo_oracle_map* m, uint64_t* codes, uint8_t* depths, size_t cap)
{
	std::vector<std::pair<uint8_t, uint64_t>> v;
	auto collect = [&](auto const& map) {
		for (auto it = map.changesBegin(); it != map.changesEnd(); ++it) {
			ufo::map::Code const c = *it;
			v.emplace_back((uint8_t)c.getDepth(), c.getCode() >> (3 * c.getDepth()));
		}
	};
	if (m->col) collect(*m->col);
	else collect(*m->occ);
	std::sort(v.begin(), v.end());
	for (size_t i = 0; i < v.size() && i < cap; ++i) {
		if (codes) codes[i] = v[i].second;
		if (depths) depths[i] = v[i].first;
	}
	return v.size();
}
int ufo_oracle_enable_minmax_change_detection(ufo_oracle_map* m, int enable)
{
	if (m->col) m->col->enableMinMaxChangeDetection(0 != enable);
	else m->occ->enableMinMaxChangeDetection(0 != enable);
	return 0;
}
size_t ufo_oracle_write_ex(const ufo_oracle_map* m, const double* c, const double* h, int compress, unsigned min_depth, int accel, int level,
                           int header, uint8_t* buf, size_t cap, long long* uncompressed_size)
{
	std::stringstream ss(std::ios_base::in | std::ios_base::out | std::ios_base::binary);
	const ufo::geometry::BoundingVolume bv = makeBv(c, h);
	long long us = -1;
	if (header) {
		const bool ok = m->col ? m->col->write(ss, bv, 0 != compress, min_depth, accel, level) : m->occ->write(ss, bv, 0 != compress, min_depth, accel, level);
		if (!ok) return (size_t)-1;
	} else {
		us = m->col ? m->col->writeData(ss, bv, 0 != compress, min_depth, accel, level) : m->occ->writeData(ss, bv, 0 != compress, min_depth, accel, level);
		if (us < 0) return (size_t)-1;
	}
	if (uncompressed_size) *uncompressed_size = us;
	std::string const bytes = ss.str();
	if (buf && cap >= bytes.size()) std::memcpy(buf, bytes.data(), bytes.size());
	return bytes.size();
}
int ufo_oracle_read(ufo_oracle_map* m, const uint8_t* buf, size_t n)
{
	std::stringstream ss(std::ios_base::in | std::ios_base::out | std::ios_base::binary);
	ss.write(reinterpret_cast<const char*>(buf), (std::streamsize)n);
	const bool ok = m->col ? m->col->read(ss) : m->occ->read(ss);
	return ok ? 0 : -1;
}
int ufo_oracle_read_data(ufo_oracle_map* m, const uint8_t* data, size_t n, const double* c, const double* h, double resolution,
                         unsigned depth_levels, int uncompressed_data_size, int compressed)
{
	std::stringstream ss(std::ios_base::in | std::ios_base::out | std::ios_base::binary);
	ss.write(reinterpret_cast<const char*>(data), (std::streamsize)n);
	const ufo::geometry::BoundingVolume bv = makeBv(c, h);
	const bool ok = m->col ? m->col->readData(ss, bv, resolution, depth_levels, uncompressed_data_size, 0 != compressed)
	                       : m->occ->readData(ss, bv, resolution, depth_levels, uncompressed_data_size, 0 != compressed);
	return ok ? 0 : -1;
}
int ufo_oracle_get_sensor_model(const ufo_oracle_map* m, double out[6])
{
	auto get = [&](auto const& map) {
		out[0] = map.getOccupiedThres();
		out[1] = map.getFreeThres();
		out[2] = map.getProbHit();
		out[3] = map.getProbMiss();
		out[4] = map.getClampingThresMin();
		out[5] = map.getClampingThresMax();
	};
	if (m->col) get(*m->col);
	else get(*m->occ);
	return 0;
}
int ufo_oracle_set_model_value(ufo_oracle_map* m, int which, double p)
{
	auto set = [&](auto& map) {
		switch (which) {
			case 2: map.setProbHit(p); break;
			case 3: map.setProbMiss(p); break;
			case 4: map.setClampingThresMin(p); break;
			case 5: map.setClampingThresMax(p); break;
			default: return -1;
		}
		return 0;
	};
	return m->col ? set(*m->col) : set(*m->occ);
}
int ufo_oracle_set_occupied_free_thres(ufo_oracle_map* m, double occupied_thres, double free_thres)
{
	if (m->col) m->col->setOccupiedFreeThres(occupied_thres, free_thres);
	else m->occ->setOccupiedFreeThres(occupied_thres, free_thres);
	return 0;
}
int ufo_oracle_clear_to(ufo_oracle_map* m, double resolution, unsigned depth_levels)
{
	if (m->col) m->col->clear(resolution, depth_levels);
	else m->occ->clear(resolution, depth_levels);
	return 0;
}
int ufo_oracle_set_value_volume_ch(ufo_oracle_map* m, const double c[3], const double h[3], double occupancy_value, unsigned min_depth)
{
	ufo::geometry::AABB a;
	a.center = ufo::geometry::Point(c[0], c[1], c[2]);
	a.half_size = ufo::geometry::Point(h[0], h[1], h[2]);
	if (m->col) m->col->setValueVolume(a, occupancy_value, min_depth);
	else m->occ->setValueVolume(a, occupancy_value, min_depth);
	return 0;
}
}  // extern "C"
