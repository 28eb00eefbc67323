// What a dependent instruction costs a lone wave on this part: chains of FP64 adds, of the DDA step's compare/select pattern, and of
// LDS ds_or marks, timed with the 100 MHz wall clock and the shader clock side by side (scripts/micro: design aids, not product code).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/micro/chain_latency.hip -o /tmp/chain && /tmp/chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_chain(double* out, unsigned long long* t, int n, double d, int mode, int waves_busy)
{
	__shared__ unsigned lds[4096];
	for (int j = threadIdx.x; j < 4096; j += blockDim.x) lds[j] = 0;
	__syncthreads();
	double a = out[threadIdx.x & 63], b = a * 0.5, c = a * 0.25;
	const double db = d * 1.5, dc = d * 2.5, dist = 1e30;
	unsigned lin = threadIdx.x * 7u;
	const unsigned long long w0 = wall_clock64(), c0 = clock64();
	if (0 == mode) {
		for (int i = 0; i < n; ++i) a = a + d;  // one dependent add per iteration
	} else if (1 == mode) {
		for (int i = 0; i < n; i += 4) {  // unrolled by four (the cut loops' form)
			const double a1 = a + d, a2 = a1 + d, a3 = a2 + d;
			a = a3 + d;
		}
	} else if (2 == mode || 3 == mode) {
		// the DDA step of k_fcast2's walk (fast_kernels.h), with (3) and without (2) the LDS mark
		int cnt = 0;
		bool go = true;
		while (go) {
			++cnt;
			if (3 == mode) atomicOr(&lds[(lin >> 5) & 4095u], 1u << (lin & 31u));
			const bool cxy = a <= b, cxz = a <= c, cyz = b <= c;
			const bool selx = cxy & cxz, sely = !cxy & cyz, selz = !(selx | sely);
			lin += (unsigned)(selx ? 1 : (sely ? 192 : 29952));
			const double nx = a + d, ny = b + db, nz = c + dc;
			a = selx ? nx : a;
			b = sely ? ny : b;
			c = selz ? nz : c;
			const bool more = (__double_as_longlong(a) <= __double_as_longlong(dist)) | (__double_as_longlong(b) <= __double_as_longlong(dist)) |
			                  (__double_as_longlong(c) <= __double_as_longlong(dist));
			go = (lin != 0xFFFFFFF0u) & more & (cnt < n);
		}
	}
	else if (4 == mode) {
		// the same step with the selected axis updated under its own mask: one addition, one stride, one range test per step
		int cnt = 0;
		bool go = true;
		bool okx = true, oky = true, okz = true;
		const long long idist = __double_as_longlong(dist);
		while (go) {
			++cnt;
			atomicOr(&lds[(lin >> 5) & 4095u], 1u << (lin & 31u));
			const bool cxy = a <= b, cxz = a <= c, cyz = b <= c;
			if (cxy & cxz) {
				a = a + d;
				lin += 1u;
				okx = __double_as_longlong(a) <= idist;
			} else if (cyz) {
				b = b + db;
				lin += 192u;
				oky = __double_as_longlong(b) <= idist;
			} else {
				c = c + dc;
				lin += 29952u;
				okz = __double_as_longlong(c) <= idist;
			}
			go = (lin != 0xFFFFFFF0u) & (okx | oky | okz) & (cnt < n);
		}
	} else if (5 == mode) {
		// two independent segments per lane, interleaved (ILP 2), branch-free selects
		int cnt = 0;
		double a2 = a * 1.1, b2 = b * 0.9, c2 = c * 1.3;
		unsigned lin2 = lin + 77u;
		bool go = true;
		while (go) {
			++cnt;
			atomicOr(&lds[(lin >> 5) & 4095u], 1u << (lin & 31u));
			atomicOr(&lds[(lin2 >> 5) & 4095u], 1u << (lin2 & 31u));
			{
				const bool cxy = a <= b, cxz = a <= c, cyz = b <= c;
				const bool selx = cxy & cxz, sely = !cxy & cyz, selz = !(selx | sely);
				lin += (unsigned)(selx ? 1 : (sely ? 192 : 29952));
				const double nx = a + d, ny = b + db, nz = c + dc;
				a = selx ? nx : a;
				b = sely ? ny : b;
				c = selz ? nz : c;
			}
			{
				const bool cxy = a2 <= b2, cxz = a2 <= c2, cyz = b2 <= c2;
				const bool selx = cxy & cxz, sely = !cxy & cyz, selz = !(selx | sely);
				lin2 += (unsigned)(selx ? 1 : (sely ? 192 : 29952));
				const double nx = a2 + d, ny = b2 + db, nz = c2 + dc;
				a2 = selx ? nx : a2;
				b2 = sely ? ny : b2;
				c2 = selz ? nz : c2;
			}
			const bool more = (__double_as_longlong(a) <= __double_as_longlong(dist)) | (__double_as_longlong(b) <= __double_as_longlong(dist)) |
			                  (__double_as_longlong(c) <= __double_as_longlong(dist)) | (__double_as_longlong(a2) <= __double_as_longlong(dist));
			go = (lin != 0xFFFFFFF0u) & (lin2 != 0xFFFFFFF1u) & more & (cnt < n);
		}
		a += a2 + b2 + c2 + lin2;
	}
	const unsigned long long w1 = wall_clock64(), c1 = clock64();
	out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + (double)lin + (double)lds[threadIdx.x];
	if (0 == threadIdx.x && 0 == blockIdx.x) {
		t[0] = w1 - w0;
		t[1] = c1 - c0;
	}
}
int main()
{
	double* out;
	unsigned long long* t;
	hipMalloc(&out, 1 << 24);
	hipMemset(out, 0, 1 << 24);
	hipMallocManaged(&t, 64);
	const char* names[] = {"dependent f64 add", "f64 add x4 unrolled", "DDA step (no mark)", "DDA step + ds_or", "DDA step, masked update", "2 DDA steps interleaved"};
	for (int warm = 0; warm < 2; ++warm)
		for (int mode = 2; mode < 6; ++mode)
			for (int threads : {64, 512, 1024}) {
				for (int blocks : {1, 256}) {
					const int n = 4096;
					hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(threads), 0, 0, out, t, n, 1e-3, mode, 0);
					hipDeviceSynchronize();
					if (warm)
						printf("%-22s threads %4d blocks %3d: %6.1f ns per iteration, %6.1f shader clocks (clock %4.0f MHz)\n", names[mode], threads, blocks, t[0] * 10.0 / n,
						       (double)t[1] / n, (double)t[1] / (t[0] * 10.0) * 1e3);
				}
			}
	return 0;
}
