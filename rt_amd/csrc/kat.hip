// rt_amd/csrc/kat.hip — librt_hip_kat.so, the TEST-ONLY companion of librt_hip.so (include/rt_hip_kat.h): device
// implementations of the path's leaf functions run on inputs of the tests' choosing — the random streams, the closest-hit
// scan (the render kernels' own code: scan.hpp), the square root and the division, and every binary32 bit pattern through
// the shortened exact-math sequences of contract.hpp.  The reference has no tests; these entry points are
// how tests/ pin the kernels' arithmetic function by function against oracle/.
//
// Not part of the drop-in surface: nothing in librt_hip.so refers to this file, and a deployment does not ship it.  It
// includes internal.hpp for the layout of rt_hip_ctx only (it reads a context's device and its resident scene), links
// librt_hip.so for nothing but the public rt_hip.h calls, and — like the product — hands the HIP runtime no caller
// memory: inputs and results travel through page-locked blocks of its own.
#include "../../include/rt_hip_kat.h"
#include "internal.hpp"
#include "scan.hpp"

#include <algorithm>

using namespace rt_hip;

namespace
{
	thread_local std::string g_kat_error;

	rt_hip_status kat_fail(rt_hip_status status, const char* format, ...) __attribute__((format(printf, 2, 3)));
	rt_hip_status kat_fail(rt_hip_status status, const char* format, ...)
	{
		char buffer[512];
		va_list args;
		va_start(args, format);
		std::vsnprintf(buffer, sizeof(buffer), format, args);
		va_end(args);
		g_kat_error = buffer;
		return status;
	}

#define RT_HIP_KAT_TRY(expr)                                                                                           \
	do                                                                                                                 \
	{                                                                                                                  \
		const hipError_t rt_hip_try_err = (expr);                                                                      \
		if (rt_hip_try_err != hipSuccess)                                                                              \
			return kat_fail(RT_HIP_RUNTIME_ERROR, "%s failed: %s", #expr, hipGetErrorString(rt_hip_try_err));          \
	}                                                                                                                  \
	while (false)

	// one call's scratch: a device block and a page-locked host block of the same size, freed when the call returns
	struct scratch
	{
		device_buffer device;
		pinned_buffer host;
		~scratch()
		{
			device.release();
			host.release();
		}
		hipError_t reserve(size_t bytes)
		{
			const hipError_t e = device.reserve(bytes);
			return e != hipSuccess ? e : host.reserve(bytes);
		}
	};
}

namespace rt_hip
{
namespace
{
	// ---- known-answer kernels -----------------------------------------------------------------------------------
	__global__ void kat_random(frame_keys frame, uint32_t pixel, uint32_t sample, uint32_t n, float* out)
	{
		if (blockIdx.x || threadIdx.x)
			return;
		stream_keys keys;
		keys.function_key = pixel_function_key(frame.a, pixel);
		keys.stride = pixel_stride(frame.b, keys.function_key);
		// the stream flattened, as oracle_random has it: the three draws of the first step, of the second, ...
		uint32_t counter = sample_counter(keys.stride, sample);
		for (uint32_t i = 0; i < n; i += 3u)
		{
			const uint32_t word = next_step_word(counter, keys);
			const uint32_t product[3] = { word * step_mul_a, word * step_mul_b, word * step_mul_c };
			for (uint32_t j = 0; j < 3u && i + j < n; j++)
				out[i + j] = step_numerator(product[j]) * random_scale;
		}
	}

	__global__ __launch_bounds__(block_threads) void kat_closest_hit(const device_scene s,
																	 uint32_t n,
																	 const float* __restrict__ origins,
																	 const float* __restrict__ directions,
																	 float* __restrict__ out_distance,
																	 uint32_t* __restrict__ out_kind,
																	 uint32_t* __restrict__ out_index,
																	 float* __restrict__ out_normal)
	{
		__shared__ float4 tile[tile_primitives];
		const uint32_t i = blockIdx.x * block_threads + threadIdx.x;
		const bool alive = i < n;
		vec3 o = { 0, 0, 0 }, d = { 0, 0, 1 };
		if (alive)
		{
			o = { origins[i * 3], origins[i * 3 + 1], origins[i * 3 + 2] };
			d = { directions[i * 3], directions[i * 3 + 1], directions[i * 3 + 2] };
		}
		candidate planes = { 0.0f, 0u, false };
		candidate spheres = { 0.0f, 0u, false };
		for (uint32_t first = 0; first < s.n_planes; first += tile_primitives)
		{
			const uint32_t count = min(tile_primitives, s.n_planes - first);
			__syncthreads();
			stage_planes(tile, s, first, count);
			__syncthreads();
			if (alive)
				scan_lds<false>(planes, o, d, tile, count, first);
		}
		for (uint32_t first = 0; first < s.n_spheres; first += tile_primitives)
		{
			const uint32_t count = min(tile_primitives, s.n_spheres - first);
			__syncthreads();
			stage_spheres(tile, s, first, count);
			__syncthreads();
			if (alive)
				scan_lds<true>(spheres, o, d, tile, count, first);
		}
		if (alive)
		{
			float distance;
			uint32_t index;
			const uint32_t kind = select_hit(spheres, planes, distance, index);
			vec3 normal;
			float4 shading;
			uint32_t scatter;
			fetch_hit<false>(s, o, d, kind, distance, index, normal, shading, scatter);
			if (!kind)
				normal = { 0.0f, 0.0f, 0.0f }; // hit_result{ -1 } of the reference: no normal
			out_distance[i] = distance;
			out_kind[i] = kind;
			out_index[i] = kind ? index : 0u;
			out_normal[i * 3 + 0] = normal.x;
			out_normal[i * 3 + 1] = normal.y;
			out_normal[i * 3 + 2] = normal.z;
		}
	}

	__global__ void kat_sqrt_div(uint32_t n, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out_sqrt, float* __restrict__ out_div)
	{
		const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
		if (i < n)
		{
			out_sqrt[i] = __builtin_sqrtf(a[i]);
			out_div[i] = a[i] / b[i];
		}
	}

	// every one of the 2^32 binary32 bit patterns through sqrt_rn / rcp_rn / inv_sqrt_rn against hipcc's general
	// correctly rounded expansions (inv_sqrt_rn: against the contract's definition, evaluated through binary64);
	// result[2k] = mismatches, result[2k+1] = smallest mismatching input + 1
	__device__ __forceinline__ bool same_float(float a, float b)
	{
		return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
	}

	__global__ __launch_bounds__(block_threads) void kat_exhaustive_math(unsigned long long* __restrict__ result)
	{
		const uint32_t tid = blockIdx.x * block_threads + threadIdx.x; // 2^22 threads x 2^10 patterns each
		uint32_t bad[3] = { 0, 0, 0 };
		unsigned long long first[3] = { ~0ull, ~0ull, ~0ull };
		for (uint32_t k = 0; k < 1024u; k++)
		{
			const uint32_t bits = (k << 22) | tid;
			const float x = __uint_as_float(bits);
			const bool ok[3] = { same_float(sqrt_rn(x), __builtin_sqrtf(x)),
								 // (the plane test's form too, over the arguments its caller may hand it: not below 2^-60, or NaN)
								 same_float(rcp_rn(x), 1.0f / x) && (__builtin_fabsf(x) < 0x1p-60f || same_float(rcp_rn_not_tiny_where(x, true), 1.0f / x)),
								 same_float(inv_sqrt_rn(x), inv_sqrt_definition(x)) };
#pragma unroll
			for (int f = 0; f < 3; f++)
				if (!ok[f])
				{
					bad[f]++;
					if (static_cast<unsigned long long>(bits) + 1ull < first[f])
						first[f] = static_cast<unsigned long long>(bits) + 1ull;
				}
		}
#pragma unroll
		for (int f = 0; f < 3; f++)
			if (bad[f])
			{
				atomicAdd(&result[2 * f], static_cast<unsigned long long>(bad[f]));
				atomicMin(&result[2 * f + 1], first[f]);
			}
	}


static void launch_kat_random(uint32_t frame_key_a, uint32_t frame_key_b, uint32_t pixel, uint32_t sample, uint32_t n, float* d_out, hipStream_t stream)
{
	hipLaunchKernelGGL(kat_random, dim3(1), dim3(64), 0, stream, frame_keys{ frame_key_a, frame_key_b }, pixel, sample, n, d_out);
}

static void launch_kat_closest_hit(const device_scene& scene,
							uint32_t n,
							const float* d_origins,
							const float* d_directions,
							float* d_distance,
							uint32_t* d_kind,
							uint32_t* d_index,
							float* d_normal,
							hipStream_t stream)
{
	const dim3 grid((n + block_threads - 1) / block_threads);
	hipLaunchKernelGGL(kat_closest_hit, grid, dim3(block_threads), 0, stream, scene, n, d_origins, d_directions, d_distance, d_kind, d_index, d_normal);
}

static void launch_kat_sqrt_div(uint32_t n, const float* d_a, const float* d_b, float* d_sqrt, float* d_div, hipStream_t stream)
{
	const dim3 grid((n + block_threads - 1) / block_threads);
	hipLaunchKernelGGL(kat_sqrt_div, grid, dim3(block_threads), 0, stream, n, d_a, d_b, d_sqrt, d_div);
}

static void launch_kat_exhaustive_math(unsigned long long* d_result, hipStream_t stream)
{
	hipLaunchKernelGGL(kat_exhaustive_math, dim3((1u << 22) / block_threads), dim3(block_threads), 0, stream, d_result);
}

}
}

extern "C" const char* rt_hip_kat_last_error(void)
{
	return g_kat_error.c_str();
}

extern "C" rt_hip_status rt_hip_kat_random(rt_hip_ctx* ctx, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, float* out)
{
	if (!ctx || !out || !n)
		return kat_fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_kat_random: invalid argument");
	RT_HIP_KAT_TRY(hipSetDevice(ctx->device));
	const size_t bytes = static_cast<size_t>(n) * sizeof(float);
	scratch s;
	RT_HIP_KAT_TRY(s.reserve(bytes));
	const frame_keys keys = make_frame_keys(seed);
	launch_kat_random(keys.a, keys.b, pixel, sample, n, s.device.as<float>(), nullptr);
	RT_HIP_KAT_TRY(hipGetLastError());
	RT_HIP_KAT_TRY(hipMemcpy(s.host.ptr, s.device.ptr, bytes, hipMemcpyDeviceToHost));
	std::memcpy(out, s.host.ptr, bytes);
	return RT_HIP_OK;
}

extern "C" rt_hip_status rt_hip_kat_closest_hit(rt_hip_ctx* ctx,
												uint32_t n,
												const float* origins,
												const float* directions,
												float* out_distance,
												uint32_t* out_kind,
												uint32_t* out_index,
												float* out_normal)
{
	if (!ctx || !n || !origins || !directions || !out_distance || !out_kind || !out_index || !out_normal)
		return kat_fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_kat_closest_hit: invalid argument");
	if (!ctx->have_scene)
		return kat_fail(RT_HIP_NO_SCENE, "rt_hip_kat_closest_hit: no scene uploaded");
	RT_HIP_KAT_TRY(hipSetDevice(ctx->device));
	const size_t vec_bytes = static_cast<size_t>(n) * 3 * sizeof(float);
	const size_t scalar_bytes = static_cast<size_t>(n) * sizeof(float);
	scratch in, out;
	RT_HIP_KAT_TRY(in.reserve(2 * vec_bytes));
	RT_HIP_KAT_TRY(out.reserve(vec_bytes + 3 * scalar_bytes));
	std::memcpy(in.host.as<unsigned char>(), origins, vec_bytes);
	std::memcpy(in.host.as<unsigned char>() + vec_bytes, directions, vec_bytes);
	RT_HIP_KAT_TRY(hipMemcpy(in.device.ptr, in.host.ptr, 2 * vec_bytes, hipMemcpyHostToDevice));
	unsigned char* const d_in = in.device.as<unsigned char>();
	unsigned char* const d_out = out.device.as<unsigned char>();
	float* d_distance = reinterpret_cast<float*>(d_out);
	uint32_t* d_kind = reinterpret_cast<uint32_t*>(d_out + scalar_bytes);
	uint32_t* d_index = reinterpret_cast<uint32_t*>(d_out + 2 * scalar_bytes);
	float* d_normal = reinterpret_cast<float*>(d_out + 3 * scalar_bytes);
	launch_kat_closest_hit(ctx->scene, n, reinterpret_cast<const float*>(d_in), reinterpret_cast<const float*>(d_in + vec_bytes), d_distance, d_kind, d_index, d_normal, nullptr);
	RT_HIP_KAT_TRY(hipGetLastError());
	RT_HIP_KAT_TRY(hipMemcpy(out.host.ptr, out.device.ptr, vec_bytes + 3 * scalar_bytes, hipMemcpyDeviceToHost));
	const unsigned char* const h_out = out.host.as<unsigned char>();
	std::memcpy(out_distance, h_out, scalar_bytes);
	std::memcpy(out_kind, h_out + scalar_bytes, scalar_bytes);
	std::memcpy(out_index, h_out + 2 * scalar_bytes, scalar_bytes);
	std::memcpy(out_normal, h_out + 3 * scalar_bytes, vec_bytes);
	return RT_HIP_OK;
}

extern "C" rt_hip_status rt_hip_kat_sqrt_div(rt_hip_ctx* ctx, uint32_t n, const float* a, const float* b, float* out_sqrt, float* out_div)
{
	if (!ctx || !n || !a || !b || !out_sqrt || !out_div)
		return kat_fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_kat_sqrt_div: invalid argument");
	RT_HIP_KAT_TRY(hipSetDevice(ctx->device));
	const size_t bytes = static_cast<size_t>(n) * sizeof(float);
	scratch in, out;
	RT_HIP_KAT_TRY(in.reserve(2 * bytes));
	RT_HIP_KAT_TRY(out.reserve(2 * bytes));
	std::memcpy(in.host.as<unsigned char>(), a, bytes);
	std::memcpy(in.host.as<unsigned char>() + bytes, b, bytes);
	RT_HIP_KAT_TRY(hipMemcpy(in.device.ptr, in.host.ptr, 2 * bytes, hipMemcpyHostToDevice));
	unsigned char* const d_in = in.device.as<unsigned char>();
	unsigned char* const d_out = out.device.as<unsigned char>();
	launch_kat_sqrt_div(n, reinterpret_cast<const float*>(d_in), reinterpret_cast<const float*>(d_in + bytes), reinterpret_cast<float*>(d_out), reinterpret_cast<float*>(d_out + bytes), nullptr);
	RT_HIP_KAT_TRY(hipGetLastError());
	RT_HIP_KAT_TRY(hipMemcpy(out.host.ptr, out.device.ptr, 2 * bytes, hipMemcpyDeviceToHost));
	std::memcpy(out_sqrt, out.host.ptr, bytes);
	std::memcpy(out_div, out.host.as<unsigned char>() + bytes, bytes);
	return RT_HIP_OK;
}

extern "C" rt_hip_status rt_hip_kat_exhaustive_math(rt_hip_ctx* ctx, uint64_t out_mismatches[3], uint32_t out_first[3])
{
	if (!ctx || !out_mismatches || !out_first)
		return kat_fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_kat_exhaustive_math: NULL argument");
	RT_HIP_KAT_TRY(hipSetDevice(ctx->device));
	scratch s;
	RT_HIP_KAT_TRY(s.reserve(6 * sizeof(unsigned long long)));
	unsigned long long* const host = s.host.as<unsigned long long>();
	const unsigned long long initial[6] = { 0, ~0ull, 0, ~0ull, 0, ~0ull };
	std::memcpy(host, initial, sizeof(initial));
	RT_HIP_KAT_TRY(hipMemcpy(s.device.ptr, host, sizeof(initial), hipMemcpyHostToDevice));
	launch_kat_exhaustive_math(s.device.as<unsigned long long>(), nullptr);
	RT_HIP_KAT_TRY(hipGetLastError());
	RT_HIP_KAT_TRY(hipMemcpy(host, s.device.ptr, sizeof(initial), hipMemcpyDeviceToHost));
	for (int f = 0; f < 3; f++)
	{
		out_mismatches[f] = host[2 * f];
		out_first[f] = host[2 * f] ? static_cast<uint32_t>(host[2 * f + 1] - 1ull) : 0u;
	}
	return RT_HIP_OK;
}

#ifdef RT_HIP_REGION_COUNTERS
// experiment variant only: out[0..12] = runs, out[13..25] = lanes of the most recent launch on this context
extern "C" __attribute__((visibility("default"))) rt_hip_status rt_hip_debug_region_counters(rt_hip_ctx* ctx, uint64_t* out)
{
	if (!ctx || !out)
		return kat_fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_debug_region_counters: NULL argument");
	RT_HIP_KAT_TRY(hipSetDevice(ctx->device));
	RT_HIP_KAT_TRY(hipEventSynchronize(ctx->counters_copied));
	for (unsigned i = 0; i < device_counters::regions; i++)
	{
		out[i] = ctx->counters_host->region_runs[i];
		out[device_counters::regions + i] = ctx->counters_host->region_lanes[i];
	}
	return RT_HIP_OK;
}
#endif

#ifdef RT_HIP_WAVE_CLOCKS
// experiment variant only: out[3 * w + {0, 1, 2}] = start / queue-dry / end tick (100 MHz) of wave w of the most recent
// persistent launch on this context, for w < *count (in: capacity of `out` in waves; out: waves the build records)
extern "C" __attribute__((visibility("default"))) rt_hip_status rt_hip_debug_wave_clocks(rt_hip_ctx* ctx, uint64_t* out, uint32_t* count)
{
	if (!ctx || !out || !count)
		return kat_fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_debug_wave_clocks: NULL argument");
	RT_HIP_KAT_TRY(hipSetDevice(ctx->device));
	RT_HIP_KAT_TRY(hipEventSynchronize(ctx->counters_copied));
	const uint32_t n = std::min<uint32_t>(*count, device_counters::clocked_waves);
	for (uint32_t w = 0; w < n; w++)
		for (int k = 0; k < 3; k++)
			out[3 * w + k] = ctx->counters_host->wave_clocks[w][k];
	*count = n;
	return RT_HIP_OK;
}
#endif
