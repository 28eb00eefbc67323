// server_calls.cpp -- the map calls of the reference's caller, line by line, against the device mirror.
//
// ufomap_ros/ufomap_mapping/src/server.cpp is the only caller of the hot path in the reference. ROS is not installed
// here, so the translation unit cannot be compiled as a whole; this file reproduces every statement of it that touches
// the map (cited by line), with the ROS-side values replaced by locals of the same type, and is compiled (and, on a GPU,
// run) by tests/test_cpp_mirror.py. If a member the server uses were missing from include/ufomap_amd/occupancy_map.hpp
// or had another signature, this file would not compile.
#include <cstdio>
#include <sstream>
#include <string>
#include <variant>
#include <vector>

#include "ufomap_amd/occupancy_map.hpp"

namespace
{
struct Config {  // ufomap_mapping/cfg/Server.cfg
	double max_range = 7.0, prob_hit = 0.7, prob_miss = 0.4, clamping_thres_min = 0.1192, clamping_thres_max = 0.971;
	unsigned insert_depth = 0, early_stopping = 0, clearing_depth = 0, publish_depth = 4;
	bool simple_ray_casting = false, async = true, compress = false;
	double robot_radius = 0.5, robot_height = 1.0;
};

// ufomap_msgs::ufoToMsg (ufomap_msgs/include/ufomap_msgs/conversions.h:162-186) minus the ROS message type
struct UFOMapMsg {
	std::string version, id;
	double resolution = 0;
	unsigned depth_levels = 0;
	bool compressed = false;
	int uncompressed_data_size = 0;
	std::vector<int8_t> data;
};
template <typename TreeType>
bool ufoToMsg(TreeType const& tree, UFOMapMsg& msg, ufo::geometry::BoundingVolume const& bounding_volume, bool compress = false,
              unsigned int depth = 0, int compression_acceleration_level = 1, int compression_level = 0)
{
	msg.version = tree.getFileVersion();
	msg.id = tree.getTreeType();
	msg.resolution = tree.getResolution();
	msg.depth_levels = tree.getTreeDepthLevels();
	msg.compressed = compress;
	std::stringstream data_stream(std::ios_base::in | std::ios_base::out | std::ios_base::binary);
	msg.uncompressed_data_size = tree.writeData(data_stream, bounding_volume, compress, depth, compression_acceleration_level, compression_level);
	if (0 > msg.uncompressed_data_size) return false;
	std::string const& data_string = data_stream.str();
	msg.data = std::vector<int8_t>(data_string.begin(), data_string.end());
	return true;
}
template <typename TreeType, typename BoundingType>
bool ufoToMsg(TreeType const& tree, UFOMapMsg& msg, BoundingType const& bounding_volume, bool compress = false, unsigned int depth = 0)
{
	ufo::geometry::BoundingVolume bv;
	bv.add(bounding_volume);
	return ufoToMsg(tree, msg, bv, compress, depth);
}
// msgToUfo (conversions.h:122-135)
template <typename TreeType>
bool msgToUfo(UFOMapMsg const& msg, TreeType& tree)
{
	std::stringstream data_stream(std::ios_base::in | std::ios_base::out | std::ios_base::binary);
	if (!msg.data.empty()) {
		data_stream.write((char const*)&msg.data[0], msg.data.size());
		return tree.readData(data_stream, ufo::geometry::BoundingVolume(), msg.resolution, msg.depth_levels, msg.uncompressed_data_size,
		                     msg.compressed);
	}
	return false;
}

template <typename Map, typename Cloud>
int serverLoop(Map& map, Cloud cloud, Config const& config, char const* save_as)
{
	// server.cpp:74
	map.enableMinMaxChangeDetection(true);
	// server.cpp:468-471 (configCallback)
	map.setProbHit(config.prob_hit);
	map.setProbMiss(config.prob_miss);
	map.setClampingThresMin(config.clamping_thres_min);
	map.setClampingThresMax(config.clamping_thres_max);

	ufo::math::Pose6 transform(0.05, 0.05, 0.05, 1.0, 0.0, 0.0, 0.0);  // rosToUfo(tf lookup), server.cpp:98-102
	// server.cpp:114-120
	cloud.transform(transform, true);
	map.insertPointCloudDiscrete(transform.translation(), cloud, config.max_range, config.insert_depth, config.simple_ray_casting,
	                             config.early_stopping, config.async);
	// server.cpp:150-154 (clear robot)
	ufo::map::Point3 r(config.robot_radius, config.robot_radius, config.robot_height / 2.0);
	ufo::geometry::AABB aabb(transform.translation() - r, transform.translation() + r);
	map.setValueVolume(aabb, map.getClampingThresMin(), config.clearing_depth);
	// server.cpp:171-201 (publish update)
	int published = 0;
	if (map.validMinMaxChange()) {
		ufo::geometry::AABB changed(map.minChange(), map.maxChange());
		map.resetMinMaxChangeDetection();
		for (unsigned i = 0; i < 2; ++i) {
			UFOMapMsg msg;
			if (ufoToMsg(map, msg, changed, config.compress, i)) published += (int)msg.data.size();
		}
	}
	// server.cpp:300-310 (publish whole map), 327-336 (GetMap)
	UFOMapMsg whole;
	if (!ufoToMsg(map, whole, ufo::geometry::BoundingVolume(), true, config.publish_depth)) return -1;
	// server.cpp:350-356 (ClearVolume): every volume of a BoundingVolume
	ufo::geometry::BoundingVolume bv;
	bv.add(ufo::geometry::AABB(ufo::geometry::Point(2, 2, 0), 0.3));
	for (auto& b : bv) map.setValueVolume(b, map.getClampingThresMin(), 1);
	// server.cpp:386-391 (SaveMap)
	bool const saved = map.write(std::string(save_as), bv, true, 0, 1, 0);
	// a client: msgToUfo into another map (ufomap_rviz_plugins, conversions.h:122-135)
	Map other(0.1);
	if (!msgToUfo(whole, other)) return -2;
	// server.cpp:371 (Reset)
	map.clear(0.1, 17);
	map.insertPointCloudWait();
	return saved ? published : -3;
}
}  // namespace

int main(int argc, char**)
{
	try {
		Config config;
		ufo::map::PointCloudColor cloud;
		cloud.push_back(ufo::map::Point3Color(0.95, 0.0, 0.0, 200, 100, 50));
		cloud.push_back(ufo::map::Point3Color(0.0, 1.9, 0.3, 20, 30, 40));
		// server.cpp:62-70: automatic pruning off
		ufo::map::OccupancyMapColor color_map(0.16, 16, false);
		int const a = serverLoop(color_map, cloud, config, "/tmp/ufomap_amd_server_calls_color.ufo");
		ufo::map::OccupancyMap plain_map(0.16, 16, false);
		ufo::map::PointCloud plain;
		plain.push_back(ufo::map::Point3(0.95, 0.0, 0.0));
		int const b = serverLoop(plain_map, plain, config, "/tmp/ufomap_amd_server_calls.ufo");
		// iterators as the rviz plugin uses them (ufomap_rviz_plugins/src/ufomap_display.cpp:267)
		ufo::map::OccupancyMap m2(0.16);
		m2.insertPointCloudDiscrete(ufo::map::Point3(0.05, 0.05, 0.05), plain, 20.0);
		unsigned n_occ = 0, n_free = 0;
		ufo::geometry::AABB box(ufo::geometry::Point(0, 0, 0), 4.0);
		for (auto it = m2.beginLeaves(box, true, true, false, false, 0), end = m2.endLeaves(); it != end; ++it) {
			if (it.isOccupied()) ++n_occ;
			if (it.isFree()) ++n_free;
			(void)it.getX();
			(void)it.getSize();
			(void)it->occupancy;
		}
		std::printf("server loop ok: %d %d; leaves in box: %u occupied, %u free; %s\n", a, b, n_occ, n_free, m2.getTreeType().c_str());
		return (a >= 0 && b >= 0 && 1 == n_occ && n_free >= 4) ? 0 : 1;
	} catch (ufo::map::DeviceError const& e) {
		std::printf("device error %d: %s\n", e.code(), e.what());
		return 2;
	}
	(void)argc;
}
