// tools/valu_rate_probe.hip — issue rate of single vector instructions on gfx950, many waves per SIMD (the render kernels'
// situation), relative to v_fma_f32:   hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate_probe tools/valu_rate_probe.hip
// Each kernel runs 8 independent chains of ONE instruction, 4096 x 8 instructions per wave, 8 waves per SIMD on every SIMD.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                           \
	do                                                                                                     \
	{                                                                                                      \
		const hipError_t e_ = (x);                                                                         \
		if (e_ != hipSuccess)                                                                              \
		{                                                                                                  \
			std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                   \
			std::exit(1);                                                                                  \
		}                                                                                                  \
	} while (0)

constexpr int trips = 16384;

#define PROBE(name, text)                                                                                  \
	__global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t seed)                              \
	{                                                                                                      \
		uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u + 1u, a2 = a0 * 5u + 2u, a3 = a0 * 7u + 3u;         \
		uint32_t a4 = a0 * 11u + 4u, a5 = a0 * 13u + 5u, a6 = a0 * 17u + 6u, a7 = a0 * 19u + 7u;           \
		const uint32_t c = seed | 0x00abcdefu;                                                             \
		uint32_t d = c ^ (threadIdx.x << 3);                                                               \
		asm volatile("s_mov_b32 s4, 0x3f800347\ns_mov_b32 s6, 0x55555555\ns_mov_b32 s7, 0x33333333\nv_cmp_lt_u32 vcc, %0, %1\n" : : "v"(c), "v"(d) : "vcc", "s4", "s6", "s7"); \
		for (int i = 0; i < trips; i++)                                                                    \
		{                                                                                                  \
			asm volatile(text "\n" : "+v"(a0) : "v"(c), "v"(d) : "vcc", "s4", "s6", "s7");                                                   \
			asm volatile(text "\n" : "+v"(a1) : "v"(c), "v"(d) : "vcc", "s4", "s6", "s7");                                                   \
			asm volatile(text "\n" : "+v"(a2) : "v"(c), "v"(d) : "vcc", "s4", "s6", "s7");                                                   \
			asm volatile(text "\n" : "+v"(a3) : "v"(c), "v"(d) : "vcc", "s4", "s6", "s7");                                                   \
			asm volatile(text "\n" : "+v"(a4) : "v"(c), "v"(d) : "vcc", "s4", "s6", "s7");                                                   \
			asm volatile(text "\n" : "+v"(a5) : "v"(c), "v"(d) : "vcc", "s4", "s6", "s7");                                                   \
			asm volatile(text "\n" : "+v"(a6) : "v"(c), "v"(d) : "vcc", "s4", "s6", "s7");                                                   \
			asm volatile(text "\n" : "+v"(a7) : "v"(c), "v"(d) : "vcc", "s4", "s6", "s7");                                                   \
		}                                                                                                  \
		out[blockIdx.x * 256u + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                      \
	}

PROBE(fma_f32, "v_fma_f32 %0, %0, %1, %1")
PROBE(add_u32, "v_add_u32 %0, %0, %1")
PROBE(xor_b32, "v_xor_b32 %0, %0, %1")
PROBE(lshrrev_b32, "v_lshrrev_b32 %0, 15, %0")
PROBE(mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
PROBE(mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
PROBE(mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
PROBE(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %1")
PROBE(mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
PROBE(cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
PROBE(rsq_f32, "v_rsq_f32 %0, %0")
PROBE(sqrt_f32, "v_sqrt_f32 %0, %0")
PROBE(rcp_f32, "v_rcp_f32 %0, %0")
PROBE(xad_u32, "v_xad_u32 %0, %0, %1, %1")
PROBE(alignbit, "v_alignbit_b32 %0, %0, %0, 15")
PROBE(bfe_u32, "v_bfe_u32 %0, %0, 8, 24")
PROBE(lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
PROBE(add3, "v_add3_u32 %0, %0, %1, %1")
PROBE(perm, "v_perm_b32 %0, %0, %1, %1")

PROBE(fmac_f32, "v_fmac_f32 %0, %1, %1")
PROBE(mul_f32, "v_mul_f32 %0, %0, %1")
PROBE(add_f32, "v_add_f32 %0, %0, %1")
PROBE(fma_f32_sgpr, "v_fma_f32 %0, %0, s4, %1")
PROBE(fma_f32_2src, "v_fma_f32 %0, %0, %0, %1")
PROBE(mov_b32, "v_mov_b32 %0, %1")
PROBE(cndmask_vcc, "v_cndmask_b32 %0, %0, %1, vcc")
PROBE(cndmask_sgpr, "v_cndmask_b32 %0, %0, %1, s[6:7]")
PROBE(cmp_f32_vcc, "v_cmp_lt_f32 vcc, %0, %1")
PROBE(cmp_f32_sgpr, "v_cmp_lt_f32 s[6:7], %0, %1")
PROBE(xor_sdwa, "v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
PROBE(and_or, "v_and_or_b32 %0, %0, %1, %1")
PROBE(max_f32, "v_max_f32 %0, %0, %1")
PROBE(sub_f32, "v_sub_f32 %0, %0, %1")
PROBE(mul_f32_e64, "v_mul_f32_e64 %0, %0, |%1|")
PROBE(ldexp_f32, "v_ldexp_f32 %0, %0, %1")

PROBE(fma_f32_3src, "v_fma_f32 %0, %0, %1, %2")
PROBE(fma_f32_const, "v_fma_f32 %0, %0, 2.0, %1")
PROBE(fmac_f32_sgpr, "v_fmac_f32 %0, s4, %1")
PROBE(fmac_f32_3, "v_fmac_f32 %0, %1, %2")
PROBE(mul_f32_sgpr, "v_mul_f32 %0, s4, %0")
PROBE(add_f32_sgpr, "v_add_f32 %0, s4, %0")
PROBE(add_f32_2, "v_add_f32 %0, %1, %2")
PROBE(min_f32, "v_min_f32 %0, %0, %1")
PROBE(fmamk, "v_fmamk_f32 %0, %0, 0x3f800347, %1")
PROBE(fmaak, "v_fmaak_f32 %0, %0, %1, 0x3f800347")

// v_mad_u64_u32 writes a register pair
__global__ void __launch_bounds__(256) mad_u64_u32(uint32_t* out, uint32_t seed)
{
	unsigned long long a[8];
	for (int k = 0; k < 8; k++)
		a[k] = threadIdx.x * (2 * k + 3) + seed;
	const uint32_t c = seed | 0x00abcdefu;
	for (int i = 0; i < trips; i++)
		for (int k = 0; k < 8; k++)
			asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0\n" : "+v"(a[k]) : "v"(c) : "vcc");
	uint32_t x = 0;
	for (int k = 0; k < 8; k++)
		x ^= static_cast<uint32_t>(a[k]) ^ static_cast<uint32_t>(a[k] >> 32);
	out[blockIdx.x * 256u + threadIdx.x] = x;
}

// packed forms work on register pairs
#define PROBE64(name, text)                                                                                \
	__global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t seed)                              \
	{                                                                                                      \
		unsigned long long a[8];                                                                           \
		for (int k = 0; k < 8; k++)                                                                        \
			a[k] = (static_cast<unsigned long long>(0x3f800000u + threadIdx.x * (k + 1)) << 32) | (0x3f900000u + seed + k); \
		const unsigned long long c = 0x3f8000013f800002ull + seed, d = 0x3f8000033f800004ull + threadIdx.x; \
		for (int i = 0; i < trips; i++)                                                                    \
			for (int k = 0; k < 8; k++)                                                                    \
				asm volatile(text "\n" : "+v"(a[k]) : "v"(c), "v"(d));                                     \
		uint32_t x = 0;                                                                                    \
		for (int k = 0; k < 8; k++)                                                                        \
			x ^= static_cast<uint32_t>(a[k]) ^ static_cast<uint32_t>(a[k] >> 32);                          \
		out[blockIdx.x * 256u + threadIdx.x] = x;                                                          \
	}
PROBE64(pk_fma_3, "v_pk_fma_f32 %0, %0, %1, %2")
PROBE64(pk_fma_2, "v_pk_fma_f32 %0, %0, %0, %1")
PROBE64(pk_mul_2, "v_pk_mul_f32 %0, %0, %1")
PROBE64(pk_add_2, "v_pk_add_f32 %0, %1, %2")
PROBE64(pk_add_neg, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]")
PROBE64(pk_mul_bcast, "v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]")
PROBE64(pk_fma_bcast, "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]")
PROBE64(fma_f64, "v_fma_f64 %0, %0, %1, %2")

int main()
{
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	const int blocks = prop.multiProcessorCount * 8; // 8 blocks of 4 waves per CU = 8 waves per SIMD
	uint32_t* out = nullptr;
	CHECK(hipMalloc(&out, static_cast<size_t>(blocks) * 256 * 4));
	hipEvent_t t0, t1;
	CHECK(hipEventCreate(&t0));
	CHECK(hipEventCreate(&t1));
	struct probe
	{
		const char* name;
		void (*kernel)(uint32_t*, uint32_t);
	} probes[] = { { "v_fma_f32", fma_f32 }, { "v_add_u32", add_u32 }, { "v_xor_b32", xor_b32 }, { "v_lshrrev_b32", lshrrev_b32 }, { "v_mul_lo_u32", mul_lo_u32 }, { "v_mul_hi_u32", mul_hi_u32 },
				   { "v_mul_u32_u24", mul_u32_u24 }, { "v_mad_u32_u24", mad_u32_u24 }, { "v_mul_hi_u32_u24", mul_hi_u32_u24 }, { "v_mad_u64_u32", mad_u64_u32 }, { "v_cvt_f32_u32", cvt_f32_u32 },
				   { "v_rsq_f32", rsq_f32 }, { "v_sqrt_f32", sqrt_f32 }, { "v_rcp_f32", rcp_f32 }, { "v_xad_u32", xad_u32 }, { "v_alignbit_b32", alignbit }, { "v_bfe_u32", bfe_u32 },
				   { "v_lshl_add_u32", lshl_add }, { "v_add3_u32", add3 }, { "v_perm_b32", perm },
				   { "v_fmac_f32 (VOP2)", fmac_f32 }, { "v_mul_f32", mul_f32 }, { "v_add_f32", add_f32 }, { "v_sub_f32", sub_f32 }, { "v_max_f32", max_f32 }, { "v_fma_f32 v,s,v", fma_f32_sgpr },
				   { "v_fma_f32 a,a,a,c", fma_f32_2src }, { "v_mul_f32_e64 |.|", mul_f32_e64 }, { "v_ldexp_f32", ldexp_f32 }, { "v_mov_b32", mov_b32 }, { "v_cndmask vcc", cndmask_vcc }, { "v_cndmask s[6:7]", cndmask_sgpr },
				   { "v_fma_f32 a,a,c,d", fma_f32_3src }, { "v_fma_f32 a,a,2.0,c", fma_f32_const }, { "v_fmac_f32 a,s,c", fmac_f32_sgpr }, { "v_fmac_f32 a,c,d", fmac_f32_3 }, { "v_mul_f32 a,s,a", mul_f32_sgpr }, { "v_add_f32 a,s,a", add_f32_sgpr }, { "v_add_f32 a,c,d", add_f32_2 }, { "v_min_f32", min_f32 }, { "v_fmamk_f32", fmamk }, { "v_fmaak_f32", fmaak },
				   { "v_pk_fma_f32 a,a,c,d", pk_fma_3 }, { "v_pk_fma_f32 a,a,a,c", pk_fma_2 }, { "v_pk_mul_f32 a,a,c", pk_mul_2 }, { "v_pk_add_f32 a,c,d", pk_add_2 }, { "v_pk_add_f32 a,c,-d", pk_add_neg }, { "v_pk_mul_f32 a,a,c.lo", pk_mul_bcast }, { "v_pk_fma_f32 a,c.lo,d,a", pk_fma_bcast }, { "v_fma_f64 a,a,c,d", fma_f64 },
				   { "v_cmp_lt_f32 vcc", cmp_f32_vcc }, { "v_cmp_lt_f32 sgpr", cmp_f32_sgpr }, { "v_xor_b32_sdwa", xor_sdwa }, { "v_and_or_b32", and_or } };
	constexpr int n_probes = sizeof(probes) / sizeof(probes[0]);
	double best[n_probes];
	for (double& b : best)
		b = 1e30;
	for (int w = 0; w < 40; w++) // clocks up
		hipLaunchKernelGGL(fma_f32, dim3(blocks), dim3(256), 0, 0, out, 12345u);
	for (int pass = 0; pass < 4; pass++) // the probes interleaved; the best of four passes each
		for (int i = 0; i < n_probes; i++)
		{
			hipLaunchKernelGGL(probes[i].kernel, dim3(blocks), dim3(256), 0, 0, out, 12345u);
			CHECK(hipEventRecord(t0));
			for (int r = 0; r < 3; r++)
				hipLaunchKernelGGL(probes[i].kernel, dim3(blocks), dim3(256), 0, 0, out, 12345u);
			CHECK(hipEventRecord(t1));
			CHECK(hipEventSynchronize(t1));
			float ms = 0;
			CHECK(hipEventElapsedTime(&ms, t0, t1));
			best[i] = ms / 3 < best[i] ? ms / 3 : best[i];
		}
	const double insts_per_simd = 8.0 * trips * 8.0; // wave-instructions per SIMD: 8 waves x trips x 8 chains
	for (int i = 0; i < n_probes; i++)
		std::printf("%-22s %8.4f ms   %5.2f x v_add_f32   %5.2f cycles per wave-instruction if the clock is 2.4 GHz\n", probes[i].name, best[i], best[i] / best[22], best[i] * 1e-3 * 2.4e9 / insts_per_simd);
	return 0;
}
