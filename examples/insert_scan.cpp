// Minimal C++ user of the drop-in surface: the KAT of SURVEY.md 8c through ufo::map::OccupancyMap.
//   g++ -std=c++17 -Iinclude examples/insert_scan.cpp ufomap_amd/csrc/libufomap_hip.so -o /tmp/insert_scan
#include <cstdio>

#include "ufomap_amd/occupancy_map.hpp"

int main()
{
	try {
		ufo::map::OccupancyMap map(0.16);
		ufo::map::PointCloud cloud;
		cloud.push_back(ufo::map::Point3(1.0, 0.05, 0.05));
		map.insertPointCloud(ufo::map::Point3(0.05, 0.05, 0.05), cloud, 20.0);
		for (auto it = map.beginLeaves(), end = map.endLeaves(); it != end; ++it)
			std::printf("%llu %u %.6f\n", (unsigned long long)it.getCode().getCode(), it.getDepth(), it->occupancy);
	} catch (ufo::map::DeviceError const& e) {
		std::printf("device error %d: %s\n", e.code(), e.what());
		return 2;
	}
	return 0;
}
