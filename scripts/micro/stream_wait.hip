// micro-benchmark: what does hipStreamWaitEvent cost a stream on this stack? (gaps between kernels from in-kernel clocks)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void spin(unsigned long long ticks, unsigned long long* out, int slot)
{
	const unsigned long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks) {}
	if (0 == threadIdx.x && 0 == blockIdx.x) {
		out[2 * slot] = t0;
		out[2 * slot + 1] = wall_clock64();
	}
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
	hipStream_t a, b;
	CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
	CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
	hipEvent_t ev;
	CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
	unsigned long long* d;
	CK(hipMalloc(&d, 64 * 8));
	unsigned long long h[64];
	const unsigned long long us = 100;  // wall_clock64 runs at 100 MHz
	std::vector<double> g0, g1, g2, g3;
	for (int rep = 0; rep < 40; ++rep) {
		// 0: same stream, back to back
		hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, b, 20 * us, d, 0);
		hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, b, 5 * us, d, 1);
		CK(hipDeviceSynchronize());
		// 1: wait on an event that completed long ago (the waiting stream reaches the wait after its own kernel)
		hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, a, 2 * us, d, 2);
		CK(hipEventRecord(ev, a));
		hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, b, 30 * us, d, 3);
		CK(hipStreamWaitEvent(b, ev, 0));
		hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, b, 5 * us, d, 4);
		CK(hipDeviceSynchronize());
		// 2: parked: the waiting stream sits at the wait when the event fires
		hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, a, 60 * us, d, 5);
		CK(hipEventRecord(ev, a));
		hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, b, 5 * us, d, 6);
		CK(hipStreamWaitEvent(b, ev, 0));
		hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, b, 5 * us, d, 7);
		CK(hipDeviceSynchronize());
		// 3: parked, nothing before the wait on the waiting stream
		hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, a, 60 * us, d, 8);
		CK(hipEventRecord(ev, a));
		CK(hipStreamWaitEvent(b, ev, 0));
		hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, b, 5 * us, d, 9);
		CK(hipDeviceSynchronize());
		CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
		if (rep < 5) continue;
		g0.push_back((double)(h[2] - h[1]) / 100.0);
		g1.push_back((double)(h[8] - h[7]) / 100.0);
		g2.push_back((double)(h[14] - h[11]) / 100.0);
		g3.push_back((double)(h[18] - h[17]) / 100.0);
	}
	auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
	printf("gap same stream back to back:                  %.2f us\n", med(g0));
	printf("gap across a wait on a long-complete event:    %.2f us\n", med(g1));
	printf("gap event fires -> parked waiter's next kernel: %.2f us (kernel before the wait)\n", med(g2));
	printf("gap event fires -> parked waiter's next kernel: %.2f us (nothing before the wait)\n", med(g3));
	return 0;
}
