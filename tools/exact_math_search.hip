// tools/exact_math_search.hip — exhaustive search for shorter exactly-rounded sqrt / reciprocal / reciprocal-sqrt sequences.
// Every candidate is run over ALL float bit patterns inside the kernels' fast band (2^-60 <= x < 2^60) and compared with
// the compiler's correctly rounded expansion; the count of mismatching inputs is printed per candidate.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/exact_math_search tools/exact_math_search.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

__device__ __forceinline__ float sqrt_core(float x, float& h_out)
{
	const float r = __builtin_amdgcn_rsqf(x);
	float s = x * r, h = 0.5f * r;
	const float e = fma_(-h, s, 0.5f);
	h = fma_(h, e, h);
	s = fma_(s, e, s);
	const float d = fma_(-s, s, x);
	s = fma_(d, h, s);
	h_out = h;
	return s;
}

__device__ __forceinline__ float rcp3(float x, float seed)
{
	float r = seed;
	const float e0 = fma_(-x, r, 1.0f);
	r = fma_(e0, r, r);
	float q = r;
	const float e1 = fma_(-x, q, 1.0f);
	q = fma_(e1, r, q);
	const float e2 = fma_(-x, q, 1.0f);
	q = fma_(e2, r, q);
	return q;
}
__device__ __forceinline__ float rcp2(float x, float seed) // no refinement of r itself
{
	const float r = seed;
	float q = r;
	const float e1 = fma_(-x, q, 1.0f);
	q = fma_(e1, r, q);
	const float e2 = fma_(-x, q, 1.0f);
	q = fma_(e2, r, q);
	return q;
}
__device__ __forceinline__ float rcp2b(float x, float seed) // refine r, one correction
{
	float r = seed;
	const float e0 = fma_(-x, r, 1.0f);
	r = fma_(e0, r, r);
	float q = r;
	const float e1 = fma_(-x, q, 1.0f);
	q = fma_(e1, r, q);
	return q;
}
__device__ __forceinline__ float rcp1(float x, float seed)
{
	const float r = seed;
	const float e1 = fma_(-x, r, 1.0f);
	return fma_(e1, r, r);
}

constexpr int n_candidates = 20;

__device__ float candidate(int c, float x, float& reference)
{
	const float sq = __builtin_sqrtf(x);
	float h;
	switch (c)
	{
		// ---- 1 / sqrt(x), both roundings (the current contract) ----
		case 0: { reference = 1.0f / sq; const float s = sqrt_core(x, h); return rcp3(s, __builtin_amdgcn_rcpf(s)); }
		case 1: { reference = 1.0f / sq; const float s = sqrt_core(x, h); return rcp3(s, h + h); }
		case 2: { reference = 1.0f / sq; const float s = sqrt_core(x, h); return rcp2(s, h + h); }
		case 3: { reference = 1.0f / sq; const float s = sqrt_core(x, h); return rcp2b(s, h + h); }
		case 4: { reference = 1.0f / sq; const float s = sqrt_core(x, h); return rcp1(s, h + h); }
		case 5: { reference = 1.0f / sq; const float s = sqrt_core(x, h); return rcp2(s, __builtin_amdgcn_rcpf(s)); }
		case 6: { reference = 1.0f / sq; const float s = sqrt_core(x, h); return rcp2b(s, __builtin_amdgcn_rcpf(s)); }
		// ---- 1 / x ----
		case 7: reference = 1.0f / x; return rcp3(x, __builtin_amdgcn_rcpf(x));
		case 8: reference = 1.0f / x; return rcp2(x, __builtin_amdgcn_rcpf(x));
		case 9: reference = 1.0f / x; return rcp2b(x, __builtin_amdgcn_rcpf(x));
		case 10: reference = 1.0f / x; return rcp1(x, __builtin_amdgcn_rcpf(x));
		// ---- sqrt(x) ----
		case 11: reference = sq; return sqrt_core(x, h);
		case 12: // one residual correction, no Goldschmidt step
		{
			reference = sq;
			const float r = __builtin_amdgcn_rsqf(x);
			float s = x * r;
			const float hh = 0.5f * r;
			const float d = fma_(-s, s, x);
			return fma_(d, hh, s);
		}
		case 13: // two residual corrections
		{
			reference = sq;
			const float r = __builtin_amdgcn_rsqf(x);
			float s = x * r;
			const float hh = 0.5f * r;
			float d = fma_(-s, s, x);
			s = fma_(d, hh, s);
			d = fma_(-s, s, x);
			return fma_(d, hh, s);
		}
		case 14: // v_sqrt_f32 seed + one residual correction with rsq
		{
			reference = sq;
			float s = __builtin_amdgcn_sqrtf(x);
			const float hh = 0.5f * __builtin_amdgcn_rsqf(x);
			const float d = fma_(-s, s, x);
			return fma_(d, hh, s);
		}
		// ---- correctly rounded 1 / sqrt(x), ONE rounding (a candidate contract): reference through double ----
		case 15:
		{
			reference = static_cast<float>(1.0 / __builtin_sqrt(static_cast<double>(x)));
			const float y = __builtin_amdgcn_rsqf(x);
			const float t = x * y;
			const float e = fma_(-t, y, 1.0f);
			return fma_(0.5f * y, e, y);
		}
		case 16:
		{
			reference = static_cast<float>(1.0 / __builtin_sqrt(static_cast<double>(x)));
			const float y = __builtin_amdgcn_rsqf(x);
			const float t = x * y;
			const float dt = fma_(x, y, -t);
			const float e = fma_(-dt, y, fma_(-t, y, 1.0f));
			return fma_(0.5f * y, e, y);
		}
		case 17: // the Goldschmidt pair, then the reciprocal square root read off h
		{
			reference = static_cast<float>(1.0 / __builtin_sqrt(static_cast<double>(x)));
			(void)sqrt_core(x, h);
			return h + h;
		}
		case 18: // Goldschmidt + exact residual correction of y = 2h
		{
			reference = static_cast<float>(1.0 / __builtin_sqrt(static_cast<double>(x)));
			(void)sqrt_core(x, h);
			const float y = h + h;
			const float t = x * y;
			const float dt = fma_(x, y, -t);
			const float e = fma_(-dt, y, fma_(-t, y, 1.0f));
			return fma_(h, e, y);
		}
		default: // 19: raw v_rsq_f32 against the correctly rounded value (how often is the hardware estimate already right?)
			reference = static_cast<float>(1.0 / __builtin_sqrt(static_cast<double>(x)));
			return __builtin_amdgcn_rsqf(x);
	}
}

__global__ void search(unsigned long long* bad, unsigned int* first)
{
	const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x; // 2^22 threads x 2^10 patterns
	uint32_t mine[n_candidates] = {};
	for (uint32_t k = 0; k < 1024u; k++)
	{
		const uint32_t bits = (k << 22) | tid;
		if (!((bits - 0x21800000u) < (0x5D800000u - 0x21800000u)))
			continue; // outside the band
		const float x = __uint_as_float(bits);
#pragma unroll
		for (int c = 0; c < n_candidates; c++)
		{
			float reference;
			const float got = candidate(c, x, reference);
			if (__float_as_uint(got) != __float_as_uint(reference))
			{
				mine[c]++;
				atomicMin(&first[c], bits);
			}
		}
	}
#pragma unroll
	for (int c = 0; c < n_candidates; c++)
		if (mine[c])
			atomicAdd(&bad[c], static_cast<unsigned long long>(mine[c]));
}

int main()
{
	unsigned long long* bad;
	unsigned int* first;
	hipMalloc(&bad, sizeof(unsigned long long) * n_candidates);
	hipMalloc(&first, sizeof(unsigned int) * n_candidates);
	hipMemset(bad, 0, sizeof(unsigned long long) * n_candidates);
	hipMemset(first, 0xFF, sizeof(unsigned int) * n_candidates);
	hipLaunchKernelGGL(search, dim3((1u << 22) / 256), dim3(256), 0, 0, bad, first);
	if (hipDeviceSynchronize() != hipSuccess)
	{
		std::printf("kernel failed\n");
		return 1;
	}
	unsigned long long h_bad[n_candidates];
	unsigned int h_first[n_candidates];
	hipMemcpy(h_bad, bad, sizeof(h_bad), hipMemcpyDeviceToHost);
	hipMemcpy(h_first, first, sizeof(h_first), hipMemcpyDeviceToHost);
	const char* names[n_candidates] = { "1/sqrt 2-round: sqrt_core + rcp3(v_rcp)   [current]",
										"1/sqrt 2-round: sqrt_core + rcp3(2h)",
										"1/sqrt 2-round: sqrt_core + rcp2(2h)",
										"1/sqrt 2-round: sqrt_core + rcp2b(2h)",
										"1/sqrt 2-round: sqrt_core + rcp1(2h)",
										"1/sqrt 2-round: sqrt_core + rcp2(v_rcp)",
										"1/sqrt 2-round: sqrt_core + rcp2b(v_rcp)",
										"1/x: rcp3(v_rcp)   [current]",
										"1/x: rcp2(v_rcp)",
										"1/x: rcp2b(v_rcp)",
										"1/x: rcp1(v_rcp)",
										"sqrt: Goldschmidt + residual   [current]",
										"sqrt: rsq, one residual correction",
										"sqrt: rsq, two residual corrections",
										"sqrt: v_sqrt seed + residual correction",
										"rsqrt 1-round: rsq + Newton",
										"rsqrt 1-round: rsq + Newton with exact residual",
										"rsqrt 1-round: Goldschmidt, y = 2h",
										"rsqrt 1-round: Goldschmidt + exact residual correction",
										"rsqrt 1-round: raw v_rsq_f32" };
	for (int c = 0; c < n_candidates; c++)
		std::printf("%-60s mismatches %12llu  first 0x%08x\n", names[c], h_bad[c], h_first[c]);
	return 0;
}
