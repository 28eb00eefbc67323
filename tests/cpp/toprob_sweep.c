// toprob_sweep.c -- TEST ONLY: ufoExpfRef (ufomap_amd/csrc/expf_ref.h: the lines the device runs in toProbF) against this
// box's libm expf -- what the reference's std::exp(float) calls (occupancy_map_base.h:911) -- for EVERY float32 in
// [lo, hi]; prints "n mismatches first_bad". Also the two functions as a shared library for the device test.
//   gcc -O2 -ffp-contract=off [-shared -fPIC] tests/cpp/toprob_sweep.c -lm
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../ufomap_amd/csrc/expf_ref.h"

void expf_ref_host(const float* x, float* out, size_t n)
{
	for (size_t i = 0; i < n; ++i) out[i] = ufoExpfRef(x[i]);
}
void expf_libm(const float* x, float* out, size_t n)
{
	for (size_t i = 0; i < n; ++i) out[i] = expf(x[i]);
}
// floats ordered as integers: bits -> monotone key
static int32_t mono(float f)
{
	int32_t i;
	memcpy(&i, &f, 4);
	return i < 0 ? (int32_t)(0x80000000u - (uint32_t)i) : i;
}
static float unmono(int32_t m)
{
	const int32_t bits = m < 0 ? (int32_t)(0x80000000u - (uint32_t)m) : m;
	float f;
	memcpy(&f, &bits, 4);
	return f;
}
#ifndef SWEEP_NO_MAIN
int main(int argc, char** argv)
{
	if (argc < 5) return 2;
	const float lo = (float)atof(argv[1]), hi = (float)atof(argv[2]);
	const long part = atol(argv[3]), parts = atol(argv[4]);
	const int64_t a = mono(lo), b = mono(hi), span = b - a + 1;
	const int64_t from = a + span * part / parts, to = a + span * (part + 1) / parts;
	unsigned long long n = 0, bad = 0;
	float first_bad = 0.f;
	for (int64_t m = from; m < to; ++m) {
		const float x = unmono((int32_t)m);
		const float e = expf(x), r = ufoExpfRef(x);
		++n;
		if (memcmp(&e, &r, 4) != 0) {
			if (!bad) first_bad = x;
			++bad;
		}
	}
	printf("%llu %llu %a\n", n, bad, (double)first_bad);
	return 0;
}
#endif
