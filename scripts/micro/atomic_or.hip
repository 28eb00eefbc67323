// micro-benchmark: how fast can 256 workgroups OR ~8k words each into one shared 100 KB grid with device-scope atomics?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void k_or(unsigned* grid, unsigned nwords, unsigned density)
{
	for (unsigned j = threadIdx.x; j < nwords; j += blockDim.x) {
		unsigned h = (j * 2654435761u) ^ (blockIdx.x * 40503u);
		h ^= h >> 13;
		h *= 0x5bd1e995u;
		h ^= h >> 15;
		if (h % 100u < density) __hip_atomic_fetch_or(&grid[j], 1u << (h & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}
__global__ __launch_bounds__(512) void k_dense(uint4* out, unsigned n4)
{
	uint4* o = out + (size_t)blockIdx.x * n4;
	for (unsigned j = threadIdx.x; j < n4; j += blockDim.x) o[j] = make_uint4(j, 1, 2, 3);
}
int main()
{
	const unsigned nwords = 25600;
	unsigned* g;
	uint4* slabs;
	hipMalloc(&g, nwords * 4);
	hipMalloc(&slabs, (size_t)256 * nwords * 4);
	hipMemset(g, 0, nwords * 4);
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	for (unsigned density : {10u, 30u, 60u, 100u}) {
		for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_or, dim3(256), dim3(512), 0, 0, g, nwords, density);
		hipEventRecord(a);
		for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_or, dim3(256), dim3(512), 0, 0, g, nwords, density);
		hipEventRecord(b);
		hipEventSynchronize(b);
		float ms;
		hipEventElapsedTime(&ms, a, b);
		printf("atomic or: density %u%% (%.0f k atomics per launch): %.2f us per launch\n", density, 256.0 * nwords * density / 100 / 1000, ms * 1000 / 20);
	}
	for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_dense, dim3(256), dim3(512), 0, 0, slabs, nwords / 4);
	hipEventRecord(a);
	for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_dense, dim3(256), dim3(512), 0, 0, slabs, nwords / 4);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms;
	hipEventElapsedTime(&ms, a, b);
	printf("dense slabs (25 MB): %.2f us per launch\n", ms * 1000 / 20);
	return 0;
}
