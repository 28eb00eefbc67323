// micro-benchmark: hipStreamWaitValue32 on signal memory written by a kernel, against hipStreamWaitEvent (stream_wait.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void spin(unsigned long long ticks, unsigned long long* out, int slot, unsigned int* flag, unsigned int value)
{
	const unsigned long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks) {}
	if (0 == threadIdx.x && 0 == blockIdx.x) {
		out[2 * slot] = t0;
		out[2 * slot + 1] = wall_clock64();
		if (flag) {
			__threadfence_system();
			__hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
	hipStream_t a, b;
	CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
	CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
	unsigned long long* d;
	CK(hipMalloc(&d, 64 * 8));
	unsigned int* flag;
	CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
	CK(hipMemset(flag, 0, 8));
	unsigned long long h[64];
	const unsigned long long us = 100;
	std::vector<double> g1, g2;
	unsigned int seq = 0;
	for (int rep = 0; rep < 40; ++rep) {
		// 1: the value is there long before the waiting stream reaches the wait
		++seq;
		hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, a, 2 * us, d, 2, flag, seq);
		hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, b, 30 * us, d, 3, (unsigned int*)nullptr, 0u);
		CK(hipStreamWaitValue32(b, flag, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
		hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, b, 5 * us, d, 4, (unsigned int*)nullptr, 0u);
		CK(hipDeviceSynchronize());
		// 2: parked
		++seq;
		hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, a, 60 * us, d, 5, flag, seq);
		hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, b, 5 * us, d, 6, (unsigned int*)nullptr, 0u);
		CK(hipStreamWaitValue32(b, flag, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
		hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, b, 5 * us, d, 7, (unsigned int*)nullptr, 0u);
		CK(hipDeviceSynchronize());
		CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
		if (rep < 5) continue;
		g1.push_back((double)(h[8] - h[7]) / 100.0);
		g2.push_back((double)(h[14] - h[11]) / 100.0);
	}
	auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
	printf("wait-value, value long there:           gap %.2f us\n", med(g1));
	printf("wait-value, parked: value written -> next kernel %.2f us\n", med(g2));
	return 0;
}
