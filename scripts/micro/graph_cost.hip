// What a pre-instantiated hipGraph costs the HOST per replay against the same kernels launched one by one (scripts/micro: design aids,
// not product code): 12 small kernels in three dependent chains (the shape of a scan's enqueue: prep, scan and map stream), with and
// without hipGraphExecKernelNodeSetParams on every node before the replay (a scan's pointers and counts change with every call).
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/graph_cost.hip -o scripts/micro/graph_cost && scripts/micro/graph_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
struct Args {
	double d[20];
	unsigned u[10];
};  // 200 bytes
__global__ void k_node(Args a, unsigned* p)
{
	if (p && a.u[0] == 0xFFFFFFFFu) *p = (unsigned)a.d[3];
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x)                                                                          \
	do {                                                                               \
		hipError_t e_ = (x);                                                           \
		if (e_ != hipSuccess) {                                                        \
			printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);         \
			return 1;                                                                  \
		}                                                                              \
	} while (0)
int main()
{
	const int NN = 12, N = 2000;
	hipStream_t s[3], launch_stream;
	for (int i = 0; i < 3; ++i) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
	CK(hipStreamCreateWithFlags(&launch_stream, hipStreamNonBlocking));
	hipEvent_t ev[2];
	for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
	unsigned* d;
	CK(hipMalloc(&d, 64));
	Args a;
	memset(&a, 0, sizeof(a));
	// ---- one by one: 4 kernels on each of three streams, two cross-stream events ----
	auto direct = [&]() {
		for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_node, dim3(1), dim3(64), 0, s[0], a, d);
		(void)hipEventRecord(ev[0], s[0]);
		(void)hipStreamWaitEvent(s[1], ev[0], 0);
		for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_node, dim3(1), dim3(64), 0, s[1], a, d);
		(void)hipEventRecord(ev[1], s[1]);
		(void)hipStreamWaitEvent(s[2], ev[1], 0);
		for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_node, dim3(1), dim3(64), 0, s[2], a, d);
	};
	for (int w = 0; w < 200; ++w) direct();
	CK(hipDeviceSynchronize());
	double t0 = now();
	for (int i = 0; i < N; ++i) direct();
	double t1 = now();
	CK(hipDeviceSynchronize());
	double t2 = now();
	printf("12 launches + 2 event pairs, one by one:            host %6.2f us per scan; device-complete %6.2f us per scan\n", (t1 - t0) / N, (t2 - t0) / N);
	// ---- the same as a graph built by hand: three chains of 4, chain c+1 depends on chain c's last node ----
	hipGraph_t g;
	CK(hipGraphCreate(&g, 0));
	std::vector<hipGraphNode_t> nodes(NN);
	hipKernelNodeParams kp{};
	void* kargs[2] = {&a, &d};
	kp.func = reinterpret_cast<void*>(&k_node);
	kp.gridDim = dim3(1);
	kp.blockDim = dim3(64);
	kp.sharedMemBytes = 0;
	kp.kernelParams = kargs;
	kp.extra = nullptr;
	for (int i = 0; i < NN; ++i) {
		hipGraphNode_t dep = i ? nodes[i - 1] : nullptr;
		CK(hipGraphAddKernelNode(&nodes[i], g, i ? &dep : nullptr, i ? 1 : 0, &kp));
	}
	hipGraphExec_t ge;
	CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
	for (int w = 0; w < 200; ++w) CK(hipGraphLaunch(ge, launch_stream));
	CK(hipDeviceSynchronize());
	t0 = now();
	for (int i = 0; i < N; ++i) (void)hipGraphLaunch(ge, launch_stream);
	t1 = now();
	CK(hipDeviceSynchronize());
	t2 = now();
	printf("graph of 12 kernel nodes (one chain), replay only:   host %6.2f us per scan; device-complete %6.2f us per scan\n", (t1 - t0) / N, (t2 - t0) / N);
	for (int upd : {4, 12}) {
		t0 = now();
		for (int i = 0; i < N; ++i) {
			a.u[1] = (unsigned)i;
			for (int k = 0; k < upd; ++k) (void)hipGraphExecKernelNodeSetParams(ge, nodes[k], &kp);
			(void)hipGraphLaunch(ge, launch_stream);
		}
		t1 = now();
		CK(hipDeviceSynchronize());
		t2 = now();
		printf("  ... with new parameters for %2d nodes per replay:   host %6.2f us per scan; device-complete %6.2f us per scan\n", upd, (t1 - t0) / N, (t2 - t0) / N);
	}
	// ---- a forked graph: three independent chains of 4 (what three streams give) ----
	hipGraph_t g2;
	CK(hipGraphCreate(&g2, 0));
	std::vector<hipGraphNode_t> n2(NN);
	for (int i = 0; i < NN; ++i) {
		hipGraphNode_t dep = (i % 4) ? n2[i - 1] : nullptr;
		CK(hipGraphAddKernelNode(&n2[i], g2, (i % 4) ? &dep : nullptr, (i % 4) ? 1 : 0, &kp));
	}
	hipGraphExec_t ge2;
	CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
	for (int w = 0; w < 200; ++w) CK(hipGraphLaunch(ge2, launch_stream));
	CK(hipDeviceSynchronize());
	t0 = now();
	for (int i = 0; i < N; ++i) (void)hipGraphLaunch(ge2, launch_stream);
	t1 = now();
	CK(hipDeviceSynchronize());
	t2 = now();
	printf("graph of 3 independent chains of 4, replay only:     host %6.2f us per scan; device-complete %6.2f us per scan\n", (t1 - t0) / N, (t2 - t0) / N);
	return 0;
}
