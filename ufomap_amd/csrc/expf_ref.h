// expf_ref.h -- std::exp(float) as the reference's toProb evaluates it (occupancy_map_base.h:911 with LogitType = float),
// bit for bit: the algorithm of glibc's expf (sysdeps/ieee754/flt-32/e_expf.c since 2.27: exp(x) = 2^(k/32) * 2^(r/32),
// table of 32 values + a cubic, evaluated in double and rounded to float once). (float)exp((double)x) is NOT the same
// function: it differs in ~0.06 % of the arguments (double rounding of a result that is correct to 2^-53 vs the single
// rounding of one that is correct to 2^-34), which is what the device used until round 3. Plain C so that the host-side
// sweep (tests/test_toprob_sweep.py: every float32 a clamped log-odds value can take, against this box's libm) compiles
// the very same lines the device runs.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define UFO_HD __host__ __device__
#else
#define UFO_HD
#endif

UFO_HD static inline float ufoExpfRef(float x)
{
	// 2^(i/32) as IEEE doubles, minus i << 47 (so that adding k << 47 lands the exponent: k = 32 e + i)
	const uint64_t T[32] = {
	    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL, 0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL,
	    0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL, 0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
	    0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL, 0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL,
	    0x3feea11473eb0187ULL, 0x3feea589994cce13ULL, 0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
	    0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL, 0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL,
	    0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL};
	const double InvLn2N = 0x1.71547652b82fep+0 * 32.0;
	const double C0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0, C1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0, C2 = 0x1.62e42ff0c52d6p-1 / 32.0;
	const double SHIFT = 0x1.8p52;
	if (x != x) return x;
	if (x > 0x1.62e42ep6f) return __builtin_inff();  // overflow
	if (x < -0x1.9fe368p6f) return 0.0f;             // underflow
	const double xd = (double)x;
	double z = InvLn2N * xd;  // x * 32 / ln 2 = k + r, k integer, |r| <= 1/2
	double kd = z + SHIFT;
	uint64_t ki;
	memcpy(&ki, &kd, 8);
	kd -= SHIFT;
	const double r = z - kd;
	uint64_t t = T[ki % 32u];
	t += ki << (52 - 5);
	double s;
	memcpy(&s, &t, 8);
	z = C0 * r + C1;
	const double r2 = r * r;
	double y = C2 * r + 1.0;
	y = z * r2 + y;
	y = y * s;
	return (float)y;
}
