// What a kernel launch costs the HOST on this box (scripts/micro: design aids, not product code): hipLaunchKernelGGL with no / small /
// 700-byte arguments, on one stream and alternating over three streams of different priorities, against hipModuleLaunchKernel-style
// launches through hipLaunchKernel with a pre-built argument array and through HIP_LAUNCH_PARAM_BUFFER_POINTER (one packed buffer).
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/launch_cost.hip -o scripts/micro/launch_cost && scripts/micro/launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
struct Big {
	double d[80];
	unsigned u[15];
};  // 700 bytes
__global__ void k_empty() {}
__global__ void k_small(unsigned* p, unsigned a, unsigned b)
{
	if (p && a == 0xFFFFFFFFu) *p = b;
}
__global__ void k_big(Big x, unsigned* p)
{
	if (p && x.u[0] == 0xFFFFFFFFu) *p = (unsigned)x.d[3];
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	hipStream_t s[3];
	int lo, hi;
	(void)hipDeviceGetStreamPriorityRange(&lo, &hi);
	for (int i = 0; i < 3; ++i) (void)hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, i == 0 ? lo : (i == 1 ? (lo + hi) / 2 : hi));
	unsigned* d;
	(void)hipMalloc(&d, 64);
	Big big;
	memset(&big, 0, sizeof(big));
	const int N = 2000;
	auto run = [&](const char* name, int nstreams, auto&& launch) {
		for (int w = 0; w < 200; ++w) launch(s[w % nstreams]);
		(void)hipDeviceSynchronize();
		const double t0 = now();
		for (int i = 0; i < N; ++i) launch(s[i % nstreams]);
		const double t1 = now();
		(void)hipDeviceSynchronize();
		printf("%-64s %d stream(s): %6.2f us per launch (host)\n", name, nstreams, (t1 - t0) / N);
	};
	for (int ns : {1, 3}) {
		run("hipLaunchKernelGGL, no arguments", ns, [&](hipStream_t st) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); });
		run("hipLaunchKernelGGL, 16 bytes of arguments", ns, [&](hipStream_t st) { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, d, 1u, 2u); });
		run("hipLaunchKernelGGL, 700 bytes of arguments", ns, [&](hipStream_t st) { hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, st, big, d); });
		run("hipLaunchKernel, argument array built once, 700 bytes", ns, [&](hipStream_t st) {
			static unsigned* dp = nullptr;
			dp = d;
			void* args[2] = {&big, &dp};
			(void)hipLaunchKernel(reinterpret_cast<const void*>(&k_big), dim3(1), dim3(64), args, 0, st);
		});
		hipFunction_t fn = nullptr;
		(void)hipGetFuncBySymbol(&fn, reinterpret_cast<const void*>(&k_big));
		if (fn) {
			run("hipModuleLaunchKernel, one packed 712-byte buffer", ns, [&](hipStream_t st) {
				struct {
					Big b;
					unsigned* p;
				} pk;
				pk.b = big;
				pk.p = d;
				size_t sz = sizeof(pk);
				void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &pk, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
				(void)hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, st, nullptr, cfg);
			});
		} else printf("hipGetFuncBySymbol not available\n");
	}
	return 0;
}
