// The headline loop of bench.py driven from C++ through the C ABI: what a C++ caller of the library (the reference's server is one)
// gets without the Python interpreter and ctypes between two calls. bench.py writes the clouds it uses to a file, runs this
// program and reports its ms per scan beside `value` (leg "host_cxx"); the final map's digest is compared with the Python legs'.
//   hipcc -O2 -std=c++17 -Iinclude examples/bench_loop.cpp ufomap_amd/csrc/libufomap_hip.so -o examples/bench_loop
// usage: bench_loop <clouds.bin> <warmup> <steps> <min_timed_seconds> <device>
//   clouds.bin: u64 n_poses, u64 n_points, then per pose: 3 doubles (sensor origin) + n_points * 3 doubles
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ufomap_hip.h"

#define CHECK(x)                                                                    \
	do {                                                                             \
		if ((x) != hipSuccess) {                                                     \
			std::fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__);        \
			return 2;                                                                \
		}                                                                            \
	} while (0)

int main(int argc, char** argv)
{
	if (argc < 6) {
		std::fprintf(stderr, "usage: %s clouds.bin warmup steps min_seconds device\n", argv[0]);
		return 1;
	}
	const int W = std::atoi(argv[2]), K = std::atoi(argv[3]), dev = std::atoi(argv[5]);
	const double min_s = std::atof(argv[4]);
	std::FILE* f = std::fopen(argv[1], "rb");
	if (!f) return 1;
	uint64_t hdr[2];
	if (std::fread(hdr, 8, 2, f) != 2) return 1;
	const size_t P = hdr[0], N = hdr[1];
	std::vector<std::vector<double>> origin(P, std::vector<double>(3));
	std::vector<double*> d_xyz(P, nullptr);
	CHECK(hipSetDevice(dev));
	{
		std::vector<double> h(N * 3);
		for (size_t p = 0; p < P; ++p) {
			if (std::fread(origin[p].data(), 8, 3, f) != 3 || std::fread(h.data(), 8, N * 3, f) != N * 3) return 1;
			CHECK(hipMalloc((void**)&d_xyz[p], N * 24));
			CHECK(hipMemcpy(d_xyz[p], h.data(), N * 24, hipMemcpyHostToDevice));  // resident in HBM before any timed region
		}
	}
	std::fclose(f);
	ufomap_map* m = ufomap_map_create(0.16, 16, 1, 0.5, 0.5, 0.7, 0.4, 0.1192, 0.971, 0, dev);
	if (!m) {
		std::fprintf(stderr, "%s\n", ufomap_last_error());
		return 2;
	}
	auto step = [&](int i) { return ufomap_map_insert_device(m, origin[i % P].data(), d_xyz[i % P], nullptr, N, 20.0, 0, 1, 0, 0, 1); };
	double total = 0;
	int reps = 0;
	std::vector<double> per_rep;
	while (total < min_s && reps < 4000) {
		if (ufomap_map_wait(m) || ufomap_map_clear(m)) return 3;
		for (int i = 0; i < W; ++i)
			if (step(i)) return 3;
		if (ufomap_map_wait(m)) return 3;
		CHECK(hipDeviceSynchronize());
		const auto t0 = std::chrono::steady_clock::now();
		for (int i = W; i < W + K; ++i)
			if (step(i)) return 3;
		if (ufomap_map_wait(m)) return 3;
		CHECK(hipDeviceSynchronize());
		const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		total += dt;
		per_rep.push_back(dt);
		++reps;
	}
	uint64_t dig[6] = {0, 0, 0, 0, 0, 0}, dbg[64] = {0};
	if (ufomap_map_digest(m, 1, dig)) return 3;
	ufomap_map_debug(m, dbg, 64);
	std::printf("{\"ms_per_step\": %.6f, \"repeats\": %d, \"timed_region_s\": %.4f, \"rays_per_s\": %.1f, \"digest\": [\"%llu\", \"%llu\", \"%llu\", \"%llu\", \"%llu\", \"%llu\"], "
	            "\"fast_path_scans\": %llu, \"tree_walks\": %llu, \"gate_timeouts\": %llu}\n",
	            total / ((double)K * reps) * 1e3, reps, total, (double)N * K * reps / total, (unsigned long long)dig[0], (unsigned long long)dig[1], (unsigned long long)dig[2],
	            (unsigned long long)dig[3], (unsigned long long)dig[4], (unsigned long long)dig[5], (unsigned long long)dbg[61], (unsigned long long)dbg[60], (unsigned long long)dbg[58]);
	ufomap_map_destroy(m);
	return 0;
}
