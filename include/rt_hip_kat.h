/*
 * rt_hip_kat.h — known-answer entry points of librt_hip_kat.so, the TEST-ONLY companion of librt_hip.so.
 *
 * Device implementations of the leaf functions of the mg_ray_tracer path (reference src/renderers/mg_ray_tracer.cpp:36-140,
 * src/random.hpp:12-66), run on inputs of the caller's choosing, for parity tests against oracle/.  The reference has no
 * tests and nothing in it corresponds.  NOT part of the drop-in surface of include/rt_hip.h: a deployment
 * ships librt_hip.so alone.  `ctx` is a context from rt_hip_create (librt_hip.so); failures leave their message in
 * rt_hip_kat_last_error().
 */
#ifndef RT_HIP_KAT_H
#define RT_HIP_KAT_H

#include "rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Message for the most recent failure of a call below on the calling thread ("" if none).  Never NULL. */
RT_HIP_API const char* rt_hip_kat_last_error(void);

/* out[i] = bits of the i-th draw: rt_hip random stream (seed, pixel, sample, draw k) for k in [0, n). */
RT_HIP_API rt_hip_status rt_hip_kat_random(rt_hip_ctx* ctx, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, float* out);

/* For n rays (origin/direction as 3 floats each, AoS) against the resident scene: closest-hit distance
 * (< 0 = miss), primitive kind (0 none, 1 sphere, 2 plane), primitive index, and hit normal (3 floats). */
RT_HIP_API rt_hip_status rt_hip_kat_closest_hit(rt_hip_ctx* ctx,
									 uint32_t n,
									 const float* origins,
									 const float* directions,
									 float* out_distance,
									 uint32_t* out_kind,
									 uint32_t* out_index,
									 float* out_normal);

/* out_sqrt[i] = sqrtf(a[i]), out_div[i] = a[i] / b[i] as the device computes them (must be correctly rounded). */
RT_HIP_API rt_hip_status rt_hip_kat_sqrt_div(rt_hip_ctx* ctx, uint32_t n, const float* a, const float* b, float* out_sqrt, float* out_div);

/* Runs ALL 2^32 binary32 bit patterns through the kernels' shortened sqrt / reciprocal / reciprocal-sqrt sequences and
 * compares each result, bit for bit, with the compiler's general correctly rounded expansion (sqrt, reciprocal) and with
 * the arithmetic contract's definition of normalize()'s reciprocal square root evaluated through binary64.
 * out_mismatches[k] = number of differing inputs, out_first[k] = smallest differing input's bits (valid if count > 0),
 * k = 0 sqrt, 1 reciprocal, 2 reciprocal of sqrt.  All three counts must be 0. */
RT_HIP_API rt_hip_status rt_hip_kat_exhaustive_math(rt_hip_ctx* ctx, uint64_t out_mismatches[3], uint32_t out_first[3]);

#ifdef __cplusplus
}
#endif

#endif /* RT_HIP_KAT_H */
