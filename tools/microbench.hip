// tools/microbench.hip — per-instruction VALU throughput on gfx950, to price the renderer's arithmetic.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/microbench tools/microbench.hip
// Each kernel runs ITER iterations of 8 independent chains of one operation per lane, on enough waves to fill
// the chip; the result is lane-operations per second (Tops/s) and cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITER = 4096;

template <typename Op>
__global__ __launch_bounds__(256) void bench(float* out, float seed)
{
	float v[8];
#pragma unroll
	for (int i = 0; i < 8; i++)
		v[i] = seed + threadIdx.x * 0.001f + i;
	for (int it = 0; it < ITER; it++)
	{
#pragma unroll
		for (int i = 0; i < 8; i++)
			v[i] = Op::apply(v[i]);
	}
	float s = 0;
#pragma unroll
	for (int i = 0; i < 8; i++)
		s += v[i];
	if (s == 123.456f)
		out[0] = s;
}

struct op_fma { static __device__ float apply(float x) { return __builtin_fmaf(x, 1.0001f, 0.5f); } };
struct op_mul { static __device__ float apply(float x) { return x * 1.0001f; } };
struct op_add { static __device__ float apply(float x) { return x + 1.0001f; } };
struct op_rcp { static __device__ float apply(float x) { return __builtin_amdgcn_rcpf(x); } };
struct op_rsq { static __device__ float apply(float x) { return __builtin_amdgcn_rsqf(x); } };
struct op_sqrt_native { static __device__ float apply(float x) { return __builtin_amdgcn_sqrtf(x); } };
struct op_sqrt_ieee { static __device__ float apply(float x) { return __builtin_sqrtf(x) + 1.5f; } };
struct op_div_ieee { static __device__ float apply(float x) { return 3.0f / x + 1.5f; } };
struct op_normalize { static __device__ float apply(float x) { return 1.0f / __builtin_sqrtf(__builtin_fmaf(x, x, 1.0f)) + 1.5f; } };
struct op_mul_lo { static __device__ float apply(float x) { return __uint_as_float(__float_as_uint(x) * 0x7feb352du); } };
struct op_mad24 { static __device__ float apply(float x) { return __uint_as_float(((__float_as_uint(x) & 0xFFFFFFu) * 0x352du) + 77u); } };
struct op_xorshift { static __device__ float apply(float x) { uint32_t u = __float_as_uint(x); u ^= u >> 15; return __uint_as_float(u + 3u); } };
struct op_hash32 { static __device__ float apply(float x) { uint32_t u = __float_as_uint(x); u ^= u >> 16; u *= 0x7feb352du; u ^= u >> 15; u *= 0x846ca68bu; u ^= u >> 16; return __uint_as_float(u); } };
struct op_cndmask { static __device__ float apply(float x) { return x > 2.0f ? x - 1.0f : x + 1.5f; } };

typedef float float2v __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void bench_pk(float* out, float seed)
{
	float2v v[8];
#pragma unroll
	for (int i = 0; i < 8; i++)
		v[i] = float2v{ seed + threadIdx.x * 0.001f + i, seed + i * 0.5f };
	const float2v a = { 1.0001f, 0.9999f }, b = { 0.5f, 0.25f };
	for (int it = 0; it < ITER; it++)
	{
#pragma unroll
		for (int i = 0; i < 8; i++)
		{
			if (KIND == 0)
				v[i] = __builtin_elementwise_fma(v[i], a, b);
			else if (KIND == 1)
				v[i] = v[i] * a;
			else
				v[i] = v[i] + a;
		}
	}
	float s = 0;
#pragma unroll
	for (int i = 0; i < 8; i++)
		s += v[i].x + v[i].y;
	if (s == 123.456f)
		out[0] = s;
}

// Issue-port experiment: the same FMA work as plain (2-cycle) or packed (4-cycle) instructions, interleaved with S
// scalar instructions per 8 lane-FMAs.  If scalar and vector instructions compete for issue slots, the packed form
// (half as many vector instructions) should lose less to the scalar ones.
template <int SCALAR_OPS, bool PACKED>
__global__ __launch_bounds__(256) void bench_mixed(float* out, float seed)
{
	float2v v[4];
#pragma unroll
	for (int i = 0; i < 4; i++)
		v[i] = float2v{ seed + threadIdx.x * 0.001f + i, seed + i * 0.5f };
	const float2v a = { 1.0001f, 0.9999f }, b = { 0.5f, 0.25f };
	unsigned s0 = blockIdx.x, s1 = 3;
	for (int it = 0; it < ITER; it++)
	{
#pragma unroll
		for (int i = 0; i < 4; i++)
		{
			if (PACKED)
				v[i] = __builtin_elementwise_fma(v[i], a, b);
			else
			{
				v[i].x = __builtin_fmaf(v[i].x, a.x, b.x);
				asm volatile("" : "+v"(v[i].x)); // keep hipcc from re-packing the pair
				v[i].y = __builtin_fmaf(v[i].y, a.y, b.y);
				asm volatile("" : "+v"(v[i].y));
			}
			if (i < SCALAR_OPS)
				asm volatile("s_add_u32 %0, %0, %1\n\ts_xor_b32 %1, %1, %0" : "+s"(s0), "+s"(s1) : : "scc"); // both write SCC
		}
	}
	float s = 0;
#pragma unroll
	for (int i = 0; i < 4; i++)
		s += v[i].x + v[i].y;
	if (s == 123.456f || s0 == 0x12345u)
		out[0] = s + s1;
}

template <typename K>
int run(const char* name, K kernel, double ops_per_iter_per_lane, float* d_out)
{
	const int blocks = 256 * 8, threads = 256; // 8 blocks per CU -> 8 waves per SIMD
	hipEvent_t t0, t1;
	CHECK(hipEventCreate(&t0));
	CHECK(hipEventCreate(&t1));
	hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d_out, 1.0f);
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(t0));
	hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d_out, 1.0f);
	CHECK(hipEventRecord(t1));
	CHECK(hipEventSynchronize(t1));
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, t0, t1));
	const double lane_ops = double(blocks) * threads * ITER * 8.0 * ops_per_iter_per_lane;
	const double tops = lane_ops / (ms * 1e-3) / 1e12;
	// cycles per wave-instruction per SIMD at 2.4 GHz: 1024 SIMDs
	const double wave_instr = lane_ops / 64.0;
	const double cycles = (ms * 1e-3) * 2.4e9 * 1024.0 / wave_instr;
	std::printf("%-16s %8.3f ms  %8.3f T lane-ops/s  %6.2f cycles/wave-op/SIMD (at 2.4 GHz)\n", name, ms, tops, cycles);
	return 0;
}

int main()
{
	float* d_out;
	CHECK(hipMalloc(&d_out, 64));
	run("v_fma_f32", bench<op_fma>, 1, d_out);
	run("v_mul_f32", bench<op_mul>, 1, d_out);
	run("v_add_f32", bench<op_add>, 1, d_out);
	run("v_pk_fma_f32", bench_pk<0>, 2, d_out);
	run("v_pk_mul_f32", bench_pk<1>, 2, d_out);
	run("v_pk_add_f32", bench_pk<2>, 2, d_out);
	run("v_rcp_f32", bench<op_rcp>, 1, d_out);
	run("v_rsq_f32", bench<op_rsq>, 1, d_out);
	run("v_sqrt_f32", bench<op_sqrt_native>, 1, d_out);
	run("sqrt ieee(+add)", bench<op_sqrt_ieee>, 1, d_out);
	run("div ieee(+add)", bench<op_div_ieee>, 1, d_out);
	run("1/sqrt(fma)+add", bench<op_normalize>, 1, d_out);
	run("v_mul_lo_u32", bench<op_mul_lo>, 1, d_out);
	run("mul_u24+add", bench<op_mad24>, 1, d_out);
	run("xorshift+add", bench<op_xorshift>, 1, d_out);
	run("hash32", bench<op_hash32>, 1, d_out);
	run("cmp+cndmask+add", bench<op_cndmask>, 1, d_out);
	std::printf("-- 8 lane-FMAs per trip as 8 v_fma_f32 (plain) or 4 v_pk_fma_f32 (packed), plus 2*S scalar ALU instructions --\n");
	run("plain  S=0", bench_mixed<0, false>, 1, d_out);
	run("packed S=0", bench_mixed<0, true>, 1, d_out);
	run("plain  S=2", bench_mixed<2, false>, 1, d_out);
	run("packed S=2", bench_mixed<2, true>, 1, d_out);
	run("plain  S=4", bench_mixed<4, false>, 1, d_out);
	run("packed S=4", bench_mixed<4, true>, 1, d_out);
	return 0;
}
