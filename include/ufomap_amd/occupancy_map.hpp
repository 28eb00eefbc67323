// occupancy_map.hpp -- C++ host-side mirror of the reference's map API for the integration path.
//
// Same namespace, class names, member signatures, defaults and error behaviour as the reference
// (UnknownFreeOccupied/ufomap v1, paths relative to ufomap/include/ufo/):
//   ufo::map::OccupancyMap / OccupancyMapColor            map/occupancy_map.h:55-85, map/occupancy_map_color.h:56-98
//   insertPointCloud(origin, cloud, max_range=-1, depth=0, simple_ray_casting=false,
//                    early_stopping=0, async=false)        map/occupancy_map_base.h:270-273
//   insertPointCloudDiscrete(...same tail...)              map/occupancy_map_base.h:340-344, map/occupancy_map_color.h:177-181
//   insertPointCloudDone() / insertPointCloudWait()        map/occupancy_map_base.h:430-443
//   sensor-model setters                                   map/occupancy_map_base.h:746-773
//   minChange()/maxChange()/resetMinMaxChangeDetection()   map/occupancy_map_base.h:793-822
//   std::invalid_argument for depth_levels outside [2,21]  map/octree.h:931-935
// Everything forwards to the C ABI of include/ufomap_hip.h; the tree lives in HBM (linear-hashed octree),
// there is no CPU fallback. Header-only; link with ufomap_amd/csrc/libufomap_hip.so.
//
// What is NOT mirrored (out of the hot path's scope): iterators, point queries, castRay,
// setValueVolume, read/write, Pose6 overloads (transform the cloud on the host first, as
// occupancy_map_base.h:329-338 does).
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <utility>
#include <stdexcept>
#include <string>
#include <vector>

#include "../ufomap_hip.h"

namespace ufo::map
{
using DepthType = unsigned int;  // map/types.h:56

struct Point3 {  // math/vector3.h: three doubles
	double x = 0, y = 0, z = 0;
	Point3() = default;
	Point3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
};

struct Color {  // map/color.h:56-86
	uint8_t r = 0, g = 0, b = 0;
	Color() = default;
	Color(uint8_t r_, uint8_t g_, uint8_t b_) : r(r_), g(g_), b(b_) {}
	bool isSet() const { return 0 != r || 0 != g || 0 != b; }
};

struct Point3Color : Point3 {  // map/types.h:60-102
	Color color;
	Point3Color() = default;
	Point3Color(double x_, double y_, double z_, uint8_t r = 0, uint8_t g = 0, uint8_t b = 0) : Point3(x_, y_, z_), color(r, g, b) {}
	Color const& getColor() const { return color; }
};

// map/point_cloud.h:62-281 -- std::vector wrapper
template <typename T>
class PointCloudT
{
 public:
	void push_back(T const& p) { cloud_.push_back(p); }
	void reserve(size_t n) { cloud_.reserve(n); }
	size_t size() const { return cloud_.size(); }
	bool empty() const { return cloud_.empty(); }
	T const& operator[](size_t i) const { return cloud_[i]; }
	T& operator[](size_t i) { return cloud_[i]; }
	auto begin() const { return cloud_.begin(); }
	auto end() const { return cloud_.end(); }
	void clear() { cloud_.clear(); }

 private:
	std::vector<T> cloud_;
};
using PointCloud = PointCloudT<Point3>;            // map/point_cloud.h:277
using PointCloudColor = PointCloudT<Point3Color>;  // map/point_cloud.h:278

class DeviceError : public std::runtime_error
{
 public:
	DeviceError(int code, std::string const& what) : std::runtime_error(what), code_(code) {}
	int code() const { return code_; }

 private:
	int code_;
};

// map/occupancy_map_base.h:75 -- the part of OccupancyMapBase that is the integration path
class OccupancyMapBase
{
 public:
	OccupancyMapBase(OccupancyMapBase const&) = delete;
	OccupancyMapBase& operator=(OccupancyMapBase const&) = delete;
	virtual ~OccupancyMapBase() { ufomap_map_destroy(map_); }

	virtual std::string getTreeType() const noexcept = 0;

	// ---- integration (occupancy_map_base.h:270-443) ------------------------------------------------
	void insertPointCloud(Point3 const& sensor_origin, PointCloud const& cloud, double max_range = -1, DepthType depth = 0,
	                      bool simple_ray_casting = false, unsigned int early_stopping = 0, bool async = false)
	{
		insert(sensor_origin, cloud, max_range, depth, false, simple_ray_casting, early_stopping, async);
	}
	void insertPointCloudDiscrete(Point3 const& sensor_origin, PointCloud const& cloud, double max_range = -1, DepthType depth = 0,
	                              bool simple_ray_casting = false, unsigned int early_stopping = 0, bool async = false)
	{
		insert(sensor_origin, cloud, max_range, depth, true, simple_ray_casting, early_stopping, async);
	}
	// The server's rosToUfo + cloud.transform(pose) + insertPointCloudDiscrete(pose.translation(), cloud, ...)
	// (ufomap_mapping/src/server.cpp:114-120) on the raw records of a sensor_msgs/PointCloud2 (msg.data,
	// msg.point_step, field offsets; r/g/b offsets -1 without colour): conversion, NaN filter and transform run
	// inside the first kernel of the scan. No counterpart of this name in the reference.
	void insertPointCloud2(double const translation[3], double const rotation_wxyz[4], void const* data, std::size_t n_points,
	                       unsigned point_step, int off_x, int off_y, int off_z, int off_r = -1, int off_g = -1, int off_b = -1,
	                       double max_range = -1, DepthType depth = 0, bool simple_ray_casting = false, unsigned int early_stopping = 0,
	                       bool async = false, bool data_on_device = false)
	{
		check(ufomap_map_insert_pointcloud2(map_, translation, rotation_wxyz, data, data_on_device, n_points, point_step, off_x, off_y, off_z,
		                                    off_r, off_g, off_b, max_range, depth, 1, simple_ray_casting, early_stopping, async));
	}
	bool insertPointCloudDone() const { return check(ufomap_map_done(map_)) != 0; }
	void insertPointCloudWait() const { check(ufomap_map_wait(map_)); }

	// ---- point queries (occupancy_map_base.h:599-728), answered by ufomap_map_query -------------------
	std::pair<float, uint8_t> query(Point3 const& coord, DepthType depth) const
	{
		double const p[3] = {coord.x, coord.y, coord.z};
		float lo = 0;
		uint8_t st = 0;
		check(ufomap_map_query(map_, p, 0, 1, depth, &lo, &st));
		return {lo, st};
	}
	enum class OccupancyState { unknown, free, occupied };  // map/types.h
	OccupancyState getState(Point3 const& coord, DepthType depth = 0) const
	{
		uint8_t st = query(coord, depth).second;
		return (st & 1) ? OccupancyState::occupied : ((st & 2) ? OccupancyState::free : OccupancyState::unknown);
	}
	bool isOccupied(Point3 const& coord, DepthType depth = 0) const { return 0 != (query(coord, depth).second & 1); }
	bool isFree(Point3 const& coord, DepthType depth = 0) const { return 0 != (query(coord, depth).second & 2); }
	bool isUnknown(Point3 const& coord, DepthType depth = 0) const { return 0 != (query(coord, depth).second & 4); }
	bool containsOccupied(Point3 const& coord, DepthType depth = 0) const { return isOccupied(coord, depth); }
	bool containsFree(Point3 const& coord, DepthType depth = 0) const { return 0 != (query(coord, depth).second & 8); }
	bool containsUnknown(Point3 const& coord, DepthType depth = 0) const { return 0 != (query(coord, depth).second & 16); }
	// getOccupancy = toProb(LogitType) with LogitType = float (occupancy_map_base.h:599-602, 911)
	double getOccupancy(Point3 const& coord, DepthType depth = 0) const { return 1.0 / (1.0 + std::exp(-query(coord, depth).first)); }

	// ---- robot clearing (occupancy_map_base.h:492-518; ufomap_mapping/src/server.cpp:152-155) --------
	// setValueVolume(ufo::geometry::AABB(min, max), occupancy_value, min_depth): the AABB is passed as the two
	// corners the server constructs it from.
	void setValueVolume(Point3 const& aabb_min, Point3 const& aabb_max, double occupancy_value, DepthType min_depth = 0)
	{
		double const mn[3] = {aabb_min.x, aabb_min.y, aabb_min.z}, mx[3] = {aabb_max.x, aabb_max.y, aabb_max.z};
		check(ufomap_map_set_value_volume(map_, mn, mx, occupancy_value, min_depth));
	}
	double getClampingThresMin() const
	{
		double a = 0, b = 0;
		check(ufomap_map_clamping_thres(map_, &a, &b));
		return a;
	}
	double getClampingThresMax() const
	{
		double a = 0, b = 0;
		check(ufomap_map_clamping_thres(map_, &a, &b));
		return b;
	}

	// ---- sensor model (occupancy_map_base.h:746-773) -------------------------------------------------
	void setSensorModel(double occupied_thres, double free_thres, double prob_hit, double prob_miss, double clamping_thres_min,
	                    double clamping_thres_max)
	{
		check(ufomap_map_set_sensor_model(map_, occupied_thres, free_thres, prob_hit, prob_miss, clamping_thres_min, clamping_thres_max));
	}

	// ---- change detection (occupancy_map_base.h:793-822) ---------------------------------------------
	Point3 minChange() const
	{
		double mn[3], mx[3];
		check(ufomap_map_minmax_change(map_, mn, mx));
		return Point3(mn[0], mn[1], mn[2]);
	}
	Point3 maxChange() const
	{
		double mn[3], mx[3];
		check(ufomap_map_minmax_change(map_, mn, mx));
		return Point3(mx[0], mx[1], mx[2]);
	}
	void resetMinMaxChangeDetection() { check(ufomap_map_reset_minmax_change(map_)); }

	void clear() { check(ufomap_map_clear(map_)); }  // octree.h:541

	// ---- read-back in the canonical dump format (what beginLeaves() yields, occupancy_map_base.h:130-137) ----
	struct Leaf {
		uint64_t code;  // Code >> 3*depth
		uint8_t depth;
		float logodds;
		Color color;
	};
	std::vector<Leaf> leaves(bool include_unknown = false) const
	{
		size_t n = ufomap_map_export_leaves(map_, include_unknown, nullptr, nullptr, nullptr, nullptr, 0);
		if (n == (size_t)-1) check(UFOMAP_ERR_DEVICE);
		std::vector<uint64_t> c(n);
		std::vector<uint8_t> d(n), rgb(3 * n);
		std::vector<float> v(n);
		ufomap_map_export_leaves(map_, include_unknown, c.data(), d.data(), v.data(), rgb.data(), n);
		std::vector<Leaf> out(n);
		for (size_t i = 0; i < n; ++i) out[i] = Leaf{c[i], d[i], v[i], Color(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2])};
		return out;
	}

	ufomap_map* handle() const { return map_; }

 protected:
	OccupancyMapBase(double resolution, DepthType depth_levels, bool automatic_pruning, double occupied_thres, double free_thres,
	                 double prob_hit, double prob_miss, double clamping_thres_min, double clamping_thres_max, bool color, int device)
	{
		if (depth_levels < 2 || depth_levels > 21) {  // octree.h:931-935
			throw std::invalid_argument("depth_levels has to be [2, 21]");
		}
		map_ = ufomap_map_create(resolution, depth_levels, automatic_pruning, occupied_thres, free_thres, prob_hit, prob_miss,
		                         clamping_thres_min, clamping_thres_max, color ? 1 : 0, device);
		if (!map_) throw DeviceError(UFOMAP_ERR_DEVICE, ufomap_last_error());
	}

	static int check(int rc)
	{
		if (rc < 0) throw DeviceError(rc, ufomap_last_error());
		return rc;
	}

	void insert(Point3 const& o, PointCloud const& cloud, double max_range, DepthType depth, bool discrete, bool simple,
	            unsigned int early_stopping, bool async)
	{
		static_assert(sizeof(Point3) == 24, "Point3 must be three packed doubles");
		const double origin[3] = {o.x, o.y, o.z};
		const double* xyz = cloud.empty() ? nullptr : &cloud[0].x;
		check(ufomap_map_insert(map_, origin, xyz, nullptr, cloud.size(), max_range, depth, discrete, simple, early_stopping, async));
	}

	ufomap_map* map_ = nullptr;
};

// map/occupancy_map.h:55-85
class OccupancyMap : public OccupancyMapBase
{
 public:
	OccupancyMap(double resolution, DepthType depth_levels = 16, bool automatic_pruning = true, double occupied_thres = 0.5,
	             double free_thres = 0.5, double prob_hit = 0.7, double prob_miss = 0.4, double clamping_thres_min = 0.1192,
	             double clamping_thres_max = 0.971, int device = 0)
	    : OccupancyMapBase(resolution, depth_levels, automatic_pruning, occupied_thres, free_thres, prob_hit, prob_miss,
	                       clamping_thres_min, clamping_thres_max, false, device)
	{
	}
	std::string getTreeType() const noexcept override { return "occupancy_map"; }
};

// map/occupancy_map_color.h:56-98
class OccupancyMapColor : public OccupancyMapBase
{
 public:
	OccupancyMapColor(double resolution, DepthType depth_levels = 16, bool automatic_pruning = true, double occupied_thres = 0.5,
	                  double free_thres = 0.5, double prob_hit = 0.7, double prob_miss = 0.4, double clamping_thres_min = 0.1192,
	                  double clamping_thres_max = 0.971, int device = 0)
	    : OccupancyMapBase(resolution, depth_levels, automatic_pruning, occupied_thres, free_thres, prob_hit, prob_miss,
	                       clamping_thres_min, clamping_thres_max, true, device)
	{
	}
	std::string getTreeType() const noexcept override { return "occupancy_map_color"; }

	using OccupancyMapBase::insertPointCloud;
	using OccupancyMapBase::insertPointCloudDiscrete;

	// occupancy_map_color.h:177-267: the colour-carrying overload (the only one the ROS server calls, server.cpp:118-120)
	void insertPointCloudDiscrete(Point3 const& sensor_origin, PointCloudColor const& cloud, double max_range = -1, DepthType depth = 0,
	                              bool simple_ray_casting = false, unsigned int early_stopping = 0, bool async = false)
	{
		// Point3Color is not a packed xyz array: split into the two arrays the C ABI takes
		std::vector<double> xyz(3 * cloud.size());
		std::vector<uint8_t> rgb(3 * cloud.size());
		for (size_t i = 0; i < cloud.size(); ++i) {
			xyz[3 * i] = cloud[i].x;
			xyz[3 * i + 1] = cloud[i].y;
			xyz[3 * i + 2] = cloud[i].z;
			rgb[3 * i] = cloud[i].color.r;
			rgb[3 * i + 1] = cloud[i].color.g;
			rgb[3 * i + 2] = cloud[i].color.b;
		}
		const double origin[3] = {sensor_origin.x, sensor_origin.y, sensor_origin.z};
		check(ufomap_map_insert(map_, origin, xyz.data(), rgb.data(), cloud.size(), max_range, depth, 1, simple_ray_casting, early_stopping,
		                        async));
	}
};
}  // namespace ufo::map
