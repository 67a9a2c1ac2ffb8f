// rt_amd/csrc/contract.hpp — device-side leaf functions of the path, written to "arithmetic contract v4"
// (v2 = v1's floating-point rules with per-pixel keyed random streams; v3 = v2 with normalize()'s reciprocal square
// root taken in one step instead of as a rounded square root followed by a rounded reciprocal; v4 = v3 with one
// generator step per random<T>() call and primary rays built from a per-pixel base: kernels.hpp, frame_params).
//
// Every function here is the gfx950 counterpart of a reference function on the mg_ray_tracer path and produces
// bit-identical binary32 results to the CPU oracle:
//   round-to-nearest-even, subnormals kept (hipcc's default f32 mode), no contraction (built with
//   -ffp-contract=off), fused multiply-adds only where __builtin_fmaf is written, sqrt and '/' correctly
//   rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt; checked on device by rt_hip_kat_sqrt_div).
// Comparisons are written in the same (negated) form as the reference so that NaNs take the same branches.
//
// Reference citations are relative to the reference root.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt_hip
{
	struct vec3
	{
		float x, y, z;
	};

	__device__ __forceinline__ vec3 operator+(vec3 a, vec3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
	__device__ __forceinline__ vec3 operator-(vec3 a, vec3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
	__device__ __forceinline__ vec3 operator*(vec3 a, vec3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
	__device__ __forceinline__ vec3 operator*(vec3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }

	__device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

	// dot(a,b) = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
	__device__ __forceinline__ float dot(vec3 a, vec3 b) { return fma(a.z, b.z, fma(a.y, b.y, a.x * b.x)); }

	// ---- correctly rounded sqrt and reciprocal, cheaper than hipcc's general expansions ------------------------
	// hipcc expands __builtin_sqrtf and '/' into sequences that are correct for every input (subnormals,
	// infinities, overflow of intermediates): ~22 and ~18 full-rate VALU slots each (profiles/r01/microbench_valu.txt).
	// Inside a band of exponents where no intermediate can overflow or underflow the core of those sequences is
	// enough.  The functions below take the core when the argument lies in the band and fall back to the general
	// expansion otherwise (a wave-uniform branch that is practically never taken), so their results are
	// BIT-IDENTICAL to __builtin_sqrtf(x), 1.0f / x and the contract's inv_sqrt(x) (below) for EVERY input; this is
	// checked on the device over all 2^32 bit patterns by rt_hip_kat_exhaustive_math (tests/test_gpu_parity.py).
#ifdef RT_HIP_FAST_BUILD
	// ---- contract "v2-fast" (RT_HIP_FLAG_FAST; this translation unit is built a second time with RT_HIP_FAST_BUILD and
	// -ffp-contract=fast) -------------------------------------------------------------------------------------------
	// The hardware's own approximations, about 1 ulp each, with no correction step and no range guard, and the compiler
	// free to fuse multiply-adds: what the reference's own build asks of its compiler (-ffast-math -ffp-contract=fast,
	// meson.build:153-160).  Results are NOT bit-identical to the oracle; tests hold them to a stated tolerance.
	__device__ __forceinline__ float sqrt_rn_where(float x, bool) { return __builtin_amdgcn_sqrtf(x); }
	__device__ __forceinline__ float sqrt_rn(float x) { return __builtin_amdgcn_sqrtf(x); }
	__device__ __forceinline__ float rcp_rn(float x) { return __builtin_amdgcn_rcpf(x); }
	__device__ __forceinline__ float rcp_in_band(float x) { return __builtin_amdgcn_rcpf(x); }
	__device__ __forceinline__ float rcp_rn_not_tiny_where(float x, bool) { return __builtin_amdgcn_rcpf(x); }
	__device__ __forceinline__ float inv_sqrt_rn(float x) { return __builtin_amdgcn_rsqf(x); }
	__device__ __forceinline__ float inv_sqrt_in_band(float x) { return __builtin_amdgcn_rsqf(x); }
	__device__ __forceinline__ float divide(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
#else
	__device__ __forceinline__ bool in_fast_band(float x) // 2^-60 <= x < 2^60 (positive, finite, normal)
	{
		return (__float_as_uint(x) - 0x21800000u) < (0x5D800000u - 0x21800000u);
	}

	__device__ __forceinline__ float sqrt_core(float x, float& half_rsq)
	{
		// x * rsq(x) with ONE residual correction.  gfx950's v_rsq_f32 is accurate enough that this is the correctly
		// rounded square root for every input in the band (the Goldschmidt step hipcc's expansion puts in front of the
		// correction never changes a result there): tools/exact_math_search.hip tried the shorter forms exhaustively,
		// rt_hip_kat_exhaustive_math re-checks the one in use on the device the tests run on.
		const float r = __builtin_amdgcn_rsqf(x);
		const float s = x * r;
		const float h = 0.5f * r;
		const float d = fma(-s, s, x);
		half_rsq = h;
		return fma(d, h, s);
	}

	__device__ __forceinline__ float rcp_core(float x, float seed)
	{
		// v_rcp_f32 with ONE residual correction: the correctly rounded reciprocal for every input in the band on
		// gfx950 (same exhaustive search; the two further steps of the IEEE division expansion never change a result)
		const float e = fma(-x, seed, 1.0f);
		return fma(e, seed, seed);
	}

	// The general expansions sit behind a branch that is practically never taken (hipcc skips it with one
	// s_cbranch_execz).  The empty asm statement keeps hipcc from if-converting that branch (it would otherwise
	// evaluate both sides on every call).
#define RT_HIP_RARE_PATH() asm volatile("; rare path: general IEEE expansion" ::: "memory")

	// == __builtin_sqrtf(x) in every lane where `wanted` holds (other lanes get an unspecified value, possibly NaN)
	__device__ __forceinline__ float sqrt_rn_where(float x, bool wanted)
	{
		const bool fast = in_fast_band(x);
		float h;
		float s = sqrt_core(x, h); // out of the band: some value (possibly NaN) that is replaced below or never read
		if (wanted && !fast) // 0, subnormal, huge, infinite, negative or NaN
		{
			RT_HIP_RARE_PATH();
			s = __builtin_sqrtf(x);
		}
		return s;
	}

	// == __builtin_sqrtf(x)
	__device__ __forceinline__ float sqrt_rn(float x) { return sqrt_rn_where(x, true); }

	// == 1.0f / x
	__device__ __forceinline__ float rcp_rn(float x)
	{
		const bool fast = in_fast_band(__builtin_fabsf(x));
		float q = rcp_core(x, __builtin_amdgcn_rcpf(x)); // out of the band: replaced below
		if (!fast)
		{
			RT_HIP_RARE_PATH();
			q = 1.0f / x;
		}
		return q;
	}

	// rcp_rn for an argument the CALLER has proved to lie in the band (|x| in 2^-60 .. 2^60): no check, no fallback
	__device__ __forceinline__ float rcp_in_band(float x) { return rcp_core(x, __builtin_amdgcn_rcpf(x)); }

	// == 1.0f / x in every lane where `wanted` holds, for a caller that has proved |x| >= 2^-60 or NaN there (other lanes get an
	// unspecified value).  The band's lower end is the caller's, so the guard is ONE comparison — is |x| huge, infinite or NaN —
	// where in_fast_band() takes a mask, an add and a compare.
	__device__ __forceinline__ float rcp_rn_not_tiny_where(float x, bool wanted)
	{
		const bool fast = __builtin_fabsf(x) < 0x1p60f;
		float q = rcp_core(x, __builtin_amdgcn_rcpf(x));
		if (wanted && !fast)
		{
			RT_HIP_RARE_PATH();
			q = 1.0f / x;
		}
		return q;
	}

	// ---- the reciprocal square root of normalize(): contract v3 -----------------------------------------------------
	// v2 defined normalize() through 1.0f / sqrtf(x) and kept BOTH roundings: a quarter-rate v_rsq AND a quarter-rate
	// v_rcp plus six multiply-adds per call.  v3 defines it as ONE Newton-Raphson step in binary32, with the residual
	// computed exactly, from the truncated reciprocal square root (oracle/cpu_ref.cpp, inv_sqrt):
	//     y  = 1/sqrt(x) rounded TOWARD ZERO to binary32                  (positive normal x; everything else below)
	//     t  = x * y                dt = fma(x, y, -t)                     (t + dt == x * y exactly)
	//     e  = fma(-dt, y, fma(-t, y, 1))                                 (the residual 1 - x*y*y, to 2^-24 of itself)
	//     r  = fma(0.5f * y, e, y)
	// r is the correctly rounded 1/sqrt(x) for every significand but one — x = 4^k * (1 - 2^-23), where the first-order
	// value lands exactly on a rounding midpoint and r is 2^-k (0.5000002 ulp); tests/test_oracle_kat.py checks that over
	// every significand.  The step squares the seed's error, so from any seed within an ulp of the truth it gives the
	// same r for all but a handful of significands; that gfx950's v_rsq_f32 estimate gives the definition's r for EVERY
	// input was established exhaustively (tools/rsqrt_search.hip over the band, profiles/r03/rsqrt_search.txt) and is
	// re-checked by rt_hip_kat_exhaustive_math on ALL 2^32 inputs on the device the tests run on.
	__device__ __forceinline__ float inv_sqrt_newton(float x, float y)
	{
		const float t = x * y;
		const float dt = fma(x, y, -t);
		const float e = fma(-dt, y, fma(-t, y, 1.0f));
		return fma(0.5f * y, e, y);
	}

	// the definition itself, for any argument (the rare path of inv_sqrt_rn and the reference of the exhaustive check)
	__device__ __forceinline__ float inv_sqrt_definition(float x)
	{
		const double exact = 1.0 / __builtin_sqrt(static_cast<double>(x));
		float y = static_cast<float>(exact);
		if (!((__float_as_uint(x) - 0x00800000u) < (0x7F800000u - 0x00800000u)))
			return y; // zero, subnormal, negative, infinite, NaN: the plain quotient (inf, NaN, 0 ...)
		if (static_cast<double>(y) > exact)
			y = __uint_as_float(__float_as_uint(y) - 1u); // toward zero
		return inv_sqrt_newton(x, y);
	}

	// == inv_sqrt_definition(x)
	__device__ __forceinline__ float inv_sqrt_rn(float x)
	{
		const bool fast = in_fast_band(x);
		float q = inv_sqrt_newton(x, __builtin_amdgcn_rsqf(x)); // out of the band: replaced below
		if (!fast)
		{
			RT_HIP_RARE_PATH();
			q = inv_sqrt_definition(x);
		}
		return q;
	}

	// inv_sqrt_rn for an argument the CALLER has proved to lie in the band: no check, no fallback
	__device__ __forceinline__ float inv_sqrt_in_band(float x) { return inv_sqrt_newton(x, __builtin_amdgcn_rsqf(x)); }

	// a / b, correctly rounded
	__device__ __forceinline__ float divide(float a, float b) { return a / b; }
#endif

	// normalize(v) = v * inv_sqrt(dot(v,v))
	__device__ __forceinline__ vec3 normalize(vec3 v)
	{
		const float inv = inv_sqrt_rn(dot(v, v));
		return v * inv;
	}

	// normalize() of a random_unit_vector() candidate (random.hpp:57-66): components are multiples of 2^-24 in [0, 1),
	// not all zero, so dot(v,v) >= 2^-48 (a sum of non-negative terms is never rounded below its largest term) and < 3:
	// always inside the band
	__device__ __forceinline__ vec3 normalize_unit_cube_draw(vec3 v)
	{
		const float inv = inv_sqrt_in_band(dot(v, v));
		return v * inv;
	}
	// The same from the draws' NUMERATORS k (v = k * 2^-24): scaling by a power of two commutes with every rounding
	// involved (products, fmas, the correctly rounded square root and reciprocal; nothing leaves the normal range:
	// 1 <= dot(k,k) < 2^50), so normalize(k * 2^-24) == k * inv_sqrt(dot(k,k)) bit for bit, without the three scalings.
	__device__ __forceinline__ vec3 normalize_unit_cube_numerators(vec3 k)
	{
		const float inv = inv_sqrt_in_band(dot(k, k));
		return k * inv;
	}

	// muu::ray::at
	__device__ __forceinline__ vec3 ray_at(vec3 o, vec3 d, float t) { return { fma(d.x, t, o.x), fma(d.y, t, o.y), fma(d.z, t, o.z) }; }

	// ---- random streams (replaces src/random.cpp:9-26) ------------------------------------
	// Contract v2: every pixel of a frame draws through its OWN keyed hash function, evaluated along its OWN arithmetic
	// progression of counters — 64 bits of key per (frame, pixel), as the reference's generators are independent of one
	// another (src/random.cpp:9-13):
	//   (fa, fb)  = the two halves of mix64(seed)                       (bijective: distinct seeds, distinct frame keys)
	//   k         = hash32(pixel ^ fa)            the pixel's FUNCTION key: a bijection of the pixel index, so no two
	//                                             pixels of a frame ever draw through the same function
	//   stride    = hash32(k ^ fb) | 1            the pixel's counter stride (odd: m -> stride * m is a bijection)
	//   counter   = stride * (sample * 4096)      before the first step of a sample: 4096 steps reserved per sample
	//   draw (v2) : counter += stride;  x = counter;  x ^= x >> 16;  x = x * 0x7feb352d + k;  x ^= x >> 15;
	//               x *= 0x846ca68b;  u = (x >> 8) * 2^-24               (v4: three draws per such step, see below)
	// i.e. the lowbias32 finaliser with the pixel's key added between its two rounds, walked with the pixel's stride,
	// and the top 24 bits of the last product taken as they are (lowbias32's closing x ^= x >> 16 only touches the low
	// half of the word: it would change the draw's lowest 8 bits and cost two instructions).  For a fixed key the map
	// counter -> x is a bijection of 32-bit words.  The 2^32 counter values are shared by all
	// pixels (a 4K x 256 spp frame makes 1.5e10 draws), but two pixels with different strides meet only at isolated
	// counters, never along a run of consecutive draws, and what they compute there goes through different functions.
	// (Contract v1 drew every pixel from ONE shared sequence at a hashed offset: at 1920x1080x256 samples most sample
	// windows overlapped another pixel's, and 3e7 samples repeated another sample's whole stream.)
	__device__ __host__ __forceinline__ uint32_t hash32(uint32_t x)
	{
		x ^= x >> 16;
		x *= 0x7feb352du;
		x ^= x >> 15;
		x *= 0x846ca68bu;
		x ^= x >> 16;
		return x;
	}

	struct frame_keys
	{
		uint32_t a, b;
	};

	__host__ inline frame_keys make_frame_keys(uint64_t seed) // the splitmix64 finaliser: a bijection of 64-bit words
	{
		uint64_t z = seed + 0x9E3779B97F4A7C15ull;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		z ^= z >> 31;
		return { static_cast<uint32_t>(z), static_cast<uint32_t>(z >> 32) };
	}

	__device__ __forceinline__ uint32_t pixel_function_key(uint32_t frame_a, uint32_t pixel_index) { return hash32(pixel_index ^ frame_a); }
	__device__ __forceinline__ uint32_t pixel_stride(uint32_t frame_b, uint32_t function_key) { return hash32(function_key ^ frame_b) | 1u; }
	// stream position before the first draw of sample `sample_index`: 4096 draws are reserved per sample
	constexpr uint32_t draws_per_sample_log2 = 12;
	__device__ __forceinline__ uint32_t sample_counter(uint32_t stride, uint32_t sample_index) { return stride * (sample_index << draws_per_sample_log2); }

	__device__ __forceinline__ uint32_t keyed_hash32(uint32_t x, uint32_t function_key) // the draw is the top 24 bits
	{
		x ^= x >> 16;
		x = x * 0x7feb352du + function_key;
		x ^= x >> 15;
		x *= 0x846ca68bu;
		return x;
	}

	// Contract v4: ONE generator step serves one call of random<T>() (src/random.hpp:12-46) and yields up to three draws:
	//   counter += stride;  x = counter;  x ^= x >> 16;  x = x * 0x7feb352d + k;  x ^= x >> 15;          the step's WORD
	//   a = x * M2,  b = x * M2 A,  c = x * M2 A^2  (mod 2^32);   u = (product >> 8) * 2^-24               its draws
	// with M2 = lowbias32's second multiplier and A = 0xadb4a92d: random<float>() takes u_a, random<vec2>() (u_a, u_b),
	// random<vec3>() (u_a, u_b, u_c).  The three products are a point of the rank-1 lattice (1, A, A^2) / 2^32 picked by a
	// hashed index (oracle/cpu_ref.cpp, "random streams", has the figures).  Per draw that is a multiply, a shift and a
	// conversion on top of a third (vec3) or a half (vec2) of a hash, where v2/v3 spent a whole hash on every component.
	constexpr uint32_t step_mul_a = 0x846ca68bu;
	constexpr uint32_t step_lattice = 0xadb4a92du;
	constexpr uint32_t step_mul_b = step_mul_a * step_lattice;				  // (mod 2^32)
	constexpr uint32_t step_mul_c = step_mul_a * step_lattice * step_lattice; // (mod 2^32)
	constexpr float random_scale = 0x1.0p-24f;
	struct stream_keys
	{
		uint32_t function_key, stride;
	};
	__device__ __forceinline__ uint32_t step_word_at(uint32_t counter, stream_keys keys);
	__device__ __forceinline__ uint32_t next_step_word(uint32_t& counter, stream_keys keys)
	{
		counter += keys.stride;
		return step_word_at(counter, keys);
	}
	__device__ __forceinline__ uint32_t step_word_at(uint32_t counter, stream_keys keys)
	{
		uint32_t x = counter;
		x ^= x >> 16;
		x = x * 0x7feb352du + keys.function_key;
		x ^= x >> 15;
		return x;
	}
	// a draw of the step as the integer k of u = k * 2^-24 (0 <= k < 2^24, exact as a float)
	__device__ __forceinline__ float step_numerator(uint32_t product) { return static_cast<float>(product >> 8); }

	// random_unit_vector(), src/random.hpp:57-66 (positive octant only)
	__device__ __forceinline__ vec3 random_unit_vector(uint32_t& counter, stream_keys keys)
	{
		uint32_t a, b, c;
		do
		{
			const uint32_t word = next_step_word(counter, keys);
			a = word * step_mul_a, b = word * step_mul_b, c = word * step_mul_c;
		}
		while (((a | b | c) >> 8) == 0u); // x == 0 && y == 0 && z == 0
		return normalize_unit_cube_numerators({ step_numerator(a), step_numerator(b), step_numerator(c) });
	}

	// ---- intersection (muu::ray::hits) --------------------------------------------------------
	constexpr float min_hit_dist = 0.001f; // mg_ray_tracer.cpp:20
	constexpr float approx_zero_epsilon = 1.0e-6f;

	// sphere given as (center, radius^2).  true = muu's optional holds a value; t is then the distance.
	__device__ __forceinline__ bool hits_sphere(vec3 o, vec3 d, vec3 center, float r2, float& t)
	{
		const vec3 e = center - o;
		const float a = dot(e, d);
		const float e2 = dot(e, e);
		const float disc = r2 - fma(-a, a, e2);
		if (disc < 0.0f)
			return false;
		const float f = sqrt_rn(disc);
		t = (e2 < r2) ? a + f : a - f;
		if (t < 0.0f)
			return false;
		return true;
	}

	__device__ __forceinline__ bool hits_plane(vec3 o, vec3 d, vec3 n, float pd, float& t)
	{
		const float den = dot(n, d);
		if (__builtin_fabsf(den) <= approx_zero_epsilon)
			return false;
		const float num = dot(n, o) + pd;
		t = (-num) * rcp_rn(den); // (contract v4: the correctly rounded reciprocal, then one product)
		if (t < 0.0f)
			return false;
		return true;
	}

	// Axis-aligned box given by its min and max corners (muu::bounding_box = center -/+ extents), slab method.  Drawn by
	// the preview only.  Selections are written as compare-and-select so that NaN (0 * inf: a ray parallel to a slab and
	// starting exactly on it) and signed zeros behave the same here and in the oracle; a NaN that survives makes
	// `tmax >= tmin` false = miss.  Origin inside the box: the exit distance, like hits_sphere's far root.
	__device__ __forceinline__ float select_min(float a, float b) { return a < b ? a : b; }
	__device__ __forceinline__ float select_max(float a, float b) { return a > b ? a : b; }
	__device__ __forceinline__ bool hits_box(vec3 o, vec3 d, vec3 lo, vec3 hi, float& t)
	{
		const vec3 inv = { rcp_rn(d.x), rcp_rn(d.y), rcp_rn(d.z) };
		const vec3 t1 = (lo - o) * inv;
		const vec3 t2 = (hi - o) * inv;
		const float tmin = select_max(select_max(select_min(t1.x, t2.x), select_min(t1.y, t2.y)), select_min(t1.z, t2.z));
		const float tmax = select_min(select_min(select_max(t1.x, t2.x), select_max(t1.y, t2.y)), select_max(t1.z, t2.z));
		if (!(tmax >= tmin) || tmax < 0.0f)
			return false;
		t = tmin >= 0.0f ? tmin : tmax;
		return true;
	}

	// ---- shading --------------------------------------------------------------------------------------------
	// mg_ray_tracer.cpp:164
	__device__ __forceinline__ vec3 sky(float dir_y)
	{
		const float t = 0.5f * (dir_y + 1.0f);
		return { fma(0.5f - 1.0f, t, 1.0f), fma(0.7f - 1.0f, t, 1.0f), fma(1.0f - 1.0f, t, 1.0f) };
	}

	__device__ __forceinline__ float clamp01(float x) { return x > 1.0f ? 1.0f : (x >= 0.0f ? x : 0.0f); }

	// rt::colour{vec3} -> uint32, src/colour.hpp:63-65,101-106
	__device__ __forceinline__ uint32_t pack_rgba8888(vec3 c)
	{
		const uint32_t r = static_cast<uint32_t>(clamp01(c.x) * 255.99999f);
		const uint32_t g = static_cast<uint32_t>(clamp01(c.y) * 255.99999f);
		const uint32_t b = static_cast<uint32_t>(clamp01(c.z) * 255.99999f);
		return (r << 24u) | (g << 16u) | (b << 8u) | 255u;
	}
}
