// How fast a 3.1 MB cloud crosses PCIe on this box (scripts/micro: design aids): hipMemcpyAsync from pinned memory against a kernel that
// reads the pinned buffer itself (zero copy), and what the host's memcpy of a pageable cloud into pinned staging costs beside them.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/pcie_read.hip -o scripts/micro/pcie_read && scripts/micro/pcie_read
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <thread>
__global__ void k_read(const double* __restrict__ src, double* __restrict__ dst, size_t n)
{
	// 24 bytes per thread, as k_fhits loads a point
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) {
		const double x = src[3 * i], y = src[3 * i + 1], z = src[3 * i + 2];
		dst[i] = x + y + z;
	}
}
__global__ void k_read16(const double2* __restrict__ src, double* __restrict__ dst, size_t n2)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n2) {
		const double2 v = src[i];
		dst[i] = v.x + v.y;
	}
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	const size_t n = 131072, bytes = n * 24;
	double *pinned, *dev, *out;
	(void)hipHostMalloc((void**)&pinned, bytes, hipHostMallocDefault);
	(void)hipMalloc((void**)&dev, bytes);
	(void)hipMalloc((void**)&out, n * 16);
	std::vector<double> pageable(3 * n, 1.0);
	memset(pinned, 0, bytes);
	hipStream_t st;
	(void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
	double* dpin = nullptr;
	(void)hipHostGetDevicePointer((void**)&dpin, pinned, 0);
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	auto dev_time = [&](auto&& f, int reps) {
		f();
		(void)hipStreamSynchronize(st);
		(void)hipEventRecord(e0, st);
		for (int r = 0; r < reps; ++r) f();
		(void)hipEventRecord(e1, st);
		(void)hipEventSynchronize(e1);
		float ms = 0;
		(void)hipEventElapsedTime(&ms, e0, e1);
		return ms * 1e3 / reps;
	};
	const double t_dma = dev_time([&] { (void)hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, st); }, 50);
	printf("hipMemcpyAsync H2D from pinned, 3.1 MB:        %7.1f us = %5.1f GB/s\n", t_dma, bytes / t_dma * 1e-3);
	const double t_k = dev_time([&] { hipLaunchKernelGGL(k_read, dim3((n + 255) / 256), dim3(256), 0, st, dpin, out, n); }, 50);
	printf("kernel reads the pinned buffer (24 B / thread): %7.1f us = %5.1f GB/s\n", t_k, bytes / t_k * 1e-3);
	const double t_k16 = dev_time([&] { hipLaunchKernelGGL(k_read16, dim3((bytes / 16 + 255) / 256), dim3(256), 0, st, reinterpret_cast<const double2*>(dpin), out, bytes / 16); }, 50);
	printf("kernel reads the pinned buffer (16 B / thread): %7.1f us = %5.1f GB/s\n", t_k16, bytes / t_k16 * 1e-3);
	const double t_kd = dev_time([&] { hipLaunchKernelGGL(k_read, dim3((n + 255) / 256), dim3(256), 0, st, dev, out, n); }, 50);
	printf("kernel reads the device copy:                   %7.1f us\n", t_kd);
	for (int threads : {1, 2, 4}) {
		double best = 1e30;
		for (int rep = 0; rep < 20; ++rep) {
			const double t0 = now();
			std::vector<std::thread> th;
			const size_t per = bytes / threads;
			for (int k = 1; k < threads; ++k) th.emplace_back([&, k] { memcpy((char*)pinned + k * per, (const char*)pageable.data() + k * per, per); });
			memcpy(pinned, pageable.data(), per);
			for (auto& t : th) t.join();
			best = std::min(best, now() - t0);
		}
		printf("host memcpy pageable -> pinned, %d thread(s):     %7.1f us = %5.1f GB/s\n", threads, best, bytes / best * 1e-3);
	}
	return 0;
}
