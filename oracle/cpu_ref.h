/*
 * oracle/cpu_ref.h — C entry points of the CPU oracle.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load liboracle.so; nothing under rt_amd/ links, imports or calls it.
 *
 * PARITY UNPINNED: the reference (marzer/rt) ships no tests, golden images or known-answer vectors,
 * cannot be built offline (its math library marzer/muu @ 06dbcecb6e6c8192d24c0d3dc98260b1144c2d70 is a
 * network-fetched meson wrap, subprojects/muu.wrap:1-8) and is non-deterministic by construction
 * (src/random.cpp:12-13).  This oracle is a restatement of src/renderers/mg_ray_tracer.cpp pinned only by
 * analytic known-answer tests derived from the reference's source (tests/test_oracle_kat.py).
 *
 * Scene and partition structs are the public ABI types of include/rt_hip.h so that the same bytes can be
 * handed to the oracle and to the HIP module.
 */
#ifndef RT_ORACLE_CPU_REF_H
#define RT_ORACLE_CPU_REF_H

#include "../include/rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_stats
{
	uint64_t primary_samples;
	uint64_t segments;
	uint64_t sphere_tests;
	uint64_t plane_tests;
	double seconds; /* wall time of the render loop */
} oracle_stats;

/* render mode bits */
enum
{
	ORACLE_TRACE_ITERATIVE = 0, /* the arithmetic contract shared with the GPU (forward throughput product) */
	ORACLE_TRACE_RECURSIVE = 1, /* literal recursion of mg_ray_tracer.cpp:155-174 (attenuation * trace(...)) */
	ORACLE_MATERIALS_SM	   = 2, /* scatter table of sm_ray_tracer.cpp:221-236: dielectric/air/vacuum/water/ice refract */
	ORACLE_PREVIEW		   = 4	/* src/renderers/rasterizer.cpp:24-85 instead of mg_ray_tracer: one ray per pixel (seed unused) */
};

/* Counter-RNG, strict IEEE render: the parity oracle.  rgba8 / rgb_f32 are local_rows x width (compact
 * stripes of `part`, see rt_hip_partition); part == NULL renders the whole frame.  n_threads <= 0 = all cores.
 * Returns 0 on success. */
int oracle_render(const rt_hip_scene* scene,
				  uint32_t width,
				  uint32_t height,
				  uint64_t seed,
				  int mode, /* ORACLE_* bits */
				  const rt_hip_partition* part,
				  uint32_t* rgba8,
				  float* rgb_f32, /* nullable */
				  int n_threads,
				  oracle_stats* stats /* nullable */);

/* Reference-faithful COST model of mg_ray_tracer (AoS primitives, recursion, std::optional, function-pointer
 * scatter table, thread_local std::mt19937 + uniform_real_distribution<float>), built with the reference's
 * release flags (-O3 -ffast-math -ffp-contract=fast -mfma -mavx2).  Not reproducible (random_device seed unless
 * fixed_seed != 0); used for the CPU baseline timing and for statistical checks only. */
int oracle_render_mt19937(const rt_hip_scene* scene,
						  uint32_t width,
						  uint32_t height,
						  uint32_t fixed_seed,
						  uint32_t* rgba8,
						  float* rgb_f32, /* nullable */
						  int n_threads,
						  oracle_stats* stats /* nullable */);

/* Leaf functions, for known-answer tests and device KATs. */
/* the first n numbers of the random stream of (pixel, sample): the three draws of its first generator step, of its second, ... */
void oracle_random(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, float* out);
/* state of the random stream of (pixel, sample) before its first draw: the pixel's function key and stride, and the counter */
void oracle_stream_keys(uint64_t seed, uint32_t n, const uint32_t* pixels, const uint32_t* samples, uint32_t* out_function_key, uint32_t* out_stride, uint32_t* out_counter);
void oracle_closest_hit(const rt_hip_scene* scene,
						uint32_t n,
						const float* origins,
						const float* directions,
						float* out_distance,
						uint32_t* out_kind,
						uint32_t* out_index,
						float* out_normal);
void oracle_sqrt_div(uint32_t n, const float* a, const float* b, float* out_sqrt, float* out_div);
void oracle_inv_sqrt(uint32_t n, const float* x, float* out); /* the contract's reciprocal square root (normalize), cpu_ref.cpp inv_sqrt */
void oracle_inv_sqrt_step(uint32_t n, const float* x, const float* estimate, float* out); /* its Newton-Raphson step alone, from a given estimate */
uint32_t oracle_pack(float r, float g, float b);			   /* rt::colour{vec3} -> uint32, colour.hpp:63-65,101-106 */
void oracle_sky(float dir_y, float* out_rgb);				   /* mg_ray_tracer.cpp:164 */
/* sm_ray_tracer.cpp:181-219: direction chosen by dielectric_scatter for the uniform number u; also the reflect probability */
void oracle_dielectric_direction(const float* dir, const float* normal, float reflectivity, float u, float* out_dir, float* out_reflect_prob);
/* ray vs axis-aligned box (center, half extents): 1 = hit and *out_t set.  The slab test of the preview. */
int oracle_hits_box(const float* origin, const float* dir, const float* center, const float* extents, float* out_t);
/* primary ray of pixel (x, y) for the jitter (ka, kb) * 2^-24 (numerators of one generator step; 2^23 = the centre);
 * returns 1 if the matrix was taken as a pinhole camera's, 2 as a perspective matrix with a finite eye (the eye form),
 * 0 if it went through the general homogeneous form */
int oracle_primary_ray(const rt_hip_scene* scene, uint32_t width, uint32_t height, uint32_t x, uint32_t y, float ka, float kb, float* out_origin, float* out_dir);

#ifdef __cplusplus
}
#endif

#endif
