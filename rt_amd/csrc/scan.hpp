// rt_amd/csrc/scan.hpp — the closest-hit scan of the path, as device functions: test_planes / test_spheres / select
// (reference src/renderers/mg_ray_tracer.cpp:36-102) for all lanes of a wave at once, with wave-uniform primitives.
// Shared by the render kernels (kernels.hip) and the known-answer kernels of the test-only library (kat.hip), which runs
// this very code on rays of the tests' choosing.
#pragma once

#include "contract.hpp"
#include "kernels.hpp"

namespace rt_hip
{
	constexpr uint32_t block_threads = 256;

	// best candidate of one linear scan (test_planes / test_spheres, mg_ray_tracer.cpp:36-87)
	struct candidate
	{
		float t;
		uint32_t index;
		bool have;
	};

	// One sphere of test_spheres (:70-79) for all lanes at once, in two halves: the part every lane needs
	// (discriminant) and the part only a possible hit needs (square root, distance, comparison with the best so far).
	struct sphere_probe
	{
		float a, e2, disc;
		bool pos; // hits_sphere did not return at `if (disc < 0)`
	};

	__device__ __forceinline__ sphere_probe probe_sphere(vec3 o, vec3 d, float4 s) // s = (center, radius^2), wave-uniform
	{
		const vec3 e = { s.x - o.x, s.y - o.y, s.z - o.z };
		sphere_probe p;
		p.a = dot(e, d);
		p.e2 = dot(e, e);
		p.disc = s.w - fma(-p.a, p.a, p.e2);
		p.pos = !(p.disc < 0.0f);
		return p;
	}

	// `lanes` = ballot of p.pos
	__device__ __forceinline__ void finish_sphere(candidate& best, const sphere_probe& p, float r2, uint32_t index, unsigned long long lanes)
	{
		if (lanes != 0) // no lane can hit: skip the square root for the whole wave
		{
			const float f = sqrt_rn_where(p.disc, p.pos); // lanes with disc < 0 never use f
			const float t = (p.e2 < r2) ? p.a + f : p.a - f;
			// hits() holds a value <=> pos && !(t < 0); the scan then drops t < min_hit_dist (which covers
			// t < 0) and anything not closer than the best so far: `hit_index && hit_dist <= *hit` (:74)
			const bool accept = p.pos && !(t < min_hit_dist) && !(best.have && best.t <= t);
			best.t = accept ? t : best.t;
			best.index = accept ? index : best.index;
			best.have = best.have || accept;
		}
	}

	__device__ __forceinline__ void test_sphere(candidate& best, vec3 o, vec3 d, float4 s, uint32_t index)
	{
		const sphere_probe p = probe_sphere(o, d, s);
		finish_sphere(best, p, s.w, index, __builtin_amdgcn_ballot_w64(p.pos));
	}

	// One plane of test_planes (:43-52).  `pl` = (normal, d), wave-uniform.
	// The distance t = -num * (1 / den) takes a correctly rounded reciprocal: v_rcp_f32 and one residual step, exact for every
	// |den| in the band 2^-60 .. 2^60.  A lane that crosses the plane has |den| > approx_zero_epsilon (or a NaN), so only the
	// band's upper end is left to check — one comparison — and lanes that do not cross compute some value nobody reads.
	// No votes: rounds 4 and 5 skipped the reciprocal for a wave whose lanes were all "hopeless" (num * den positive and finite:
	// t < 0 for certain) and skipped the test for a wave without a crossing lane; measured against the straight line below, the
	// votes cost more than they saved on every plane scene (profiles/r05/plane_ab.txt: 2.42 -> 2.41, 2.77 -> 2.745, 2.745 -> 2.69 ms).
	__device__ __forceinline__ void test_plane(candidate& best, vec3 o, vec3 d, float4 pl, uint32_t index)
	{
		const vec3 n = { pl.x, pl.y, pl.z };
		const float den = dot(n, d);
		const bool crosses = !(__builtin_fabsf(den) <= approx_zero_epsilon);
		const float num = dot(n, o) + pl.w;
		const float t = (-num) * rcp_rn_not_tiny_where(den, crosses);
		const bool accept = crosses && !(t < min_hit_dist) && !(best.have && best.t <= t);
		best.t = accept ? t : best.t;
		best.index = accept ? index : best.index;
		best.have = best.have || accept;
	}

	// The ONE plane of a scene that has one (rt's scenes with their ground plane), for the scalar-register kernels.
	// test_planes' candidate is accepted <=> crosses && !(t < min_hit_dist), and select() (:96-102) then asks `t >= 0` of it, which
	// a NaN fails and every other accepted distance passes (min_hit_dist > 0): both are the one ORDERED comparison t >= min_hit_dist.
	// And because nothing but that comparison reads t in a lane that is not selected, the reciprocal needs no guard here: the
	// caller's scene has tame normals (device_scene::planes_tame) and a ray's direction is normalised, so den = n . d is below 2^60,
	// infinite or NaN; a crossing lane has |den| > approx_zero_epsilon: inside the band rcp_in_band() is the rounded quotient, and
	// for an infinite or NaN den both it (NaN) and the quotient (0, NaN) make a t that fails the comparison
	// (tests/test_oracle_kat.py::test_one_plane_is_accepted_and_selected_by_one_ordered_comparison).
	__device__ __forceinline__ bool test_one_plane(vec3 o, vec3 d, float4 pl, float& t)
	{
		const vec3 n = { pl.x, pl.y, pl.z };
		const float den = dot(n, d);
		const bool crosses = !(__builtin_fabsf(den) <= approx_zero_epsilon);
		const float num = dot(n, o) + pl.w;
		t = (-num) * rcp_in_band(den);
		return crosses && t >= min_hit_dist;
	}

	// scan `count` primitives held in LDS (wave-uniform addresses: broadcast reads).  Spheres go four at a time:
	// four independent discriminants (instruction-level parallelism, LDS reads issued together), ONE ballot for
	// the group; the square-root halves run, in index order, only if some lane may hit one of the four.
	template <bool SPHERES>
	__device__ __forceinline__ void scan_lds(candidate& best, vec3 o, vec3 d, const float4* lds, uint32_t count, uint32_t first_index)
	{
		uint32_t i = 0;
		if (SPHERES)
		{
			for (; i + 4 <= count; i += 4)
			{
				const float4 s0 = lds[i], s1 = lds[i + 1], s2 = lds[i + 2], s3 = lds[i + 3];
				const sphere_probe p0 = probe_sphere(o, d, s0);
				const sphere_probe p1 = probe_sphere(o, d, s1);
				const sphere_probe p2 = probe_sphere(o, d, s2);
				const sphere_probe p3 = probe_sphere(o, d, s3);
				// the four comparison masks are combined on the scalar unit
				const unsigned long long m0 = __builtin_amdgcn_ballot_w64(p0.pos), m1 = __builtin_amdgcn_ballot_w64(p1.pos);
				const unsigned long long m2 = __builtin_amdgcn_ballot_w64(p2.pos), m3 = __builtin_amdgcn_ballot_w64(p3.pos);
				if ((m0 | m1 | m2 | m3) != 0)
				{
					finish_sphere(best, p0, s0.w, first_index + i, m0);
					finish_sphere(best, p1, s1.w, first_index + i + 1, m1);
					finish_sphere(best, p2, s2.w, first_index + i + 2, m2);
					finish_sphere(best, p3, s3.w, first_index + i + 3, m3);
				}
			}
		}
		for (; i < count; i++)
		{
			if (SPHERES)
				test_sphere(best, o, d, lds[i], first_index + i);
			else
				test_plane(best, o, d, lds[i], first_index + i);
		}
	}

	// cooperative copy of `count` primitives starting at `first` from the SoA columns into float4 LDS slots;
	// lane i of the workgroup reads element first+i of each column (coalesced), radius is squared on the way in
	__device__ __forceinline__ void stage_spheres(float4* lds, const device_scene& s, uint32_t first, uint32_t count)
	{
		for (uint32_t i = threadIdx.x; i < count; i += block_threads)
		{
			const float r = s.sphere_r[first + i];
			lds[i] = make_float4(s.sphere_cx[first + i], s.sphere_cy[first + i], s.sphere_cz[first + i], r * r);
		}
	}

	__device__ __forceinline__ void stage_planes(float4* lds, const device_scene& s, uint32_t first, uint32_t count)
	{
		for (uint32_t i = threadIdx.x; i < count; i += block_threads)
			lds[i] = make_float4(s.plane_nx[first + i], s.plane_ny[first + i], s.plane_nz[first + i], s.plane_d[first + i]);
	}

	// select(test_spheres, test_planes), then select(test_boxes, ...) which never changes anything —
	// mg_ray_tracer.cpp:96-102,160-162.  `operator bool` of hit_result is `distance >= 0` (:29-32), which also
	// rejects a NaN distance.  Returns 0 = miss, 1 = sphere, 2 = plane.
	__device__ __forceinline__ uint32_t select_hit(const candidate& spheres, const candidate& planes, float& distance, uint32_t& index)
	{
		const float sphere_distance = spheres.have ? spheres.t : -1.0f;
		const float plane_distance = planes.have ? planes.t : -1.0f;
		const bool a = sphere_distance >= 0.0f;
		const bool b = plane_distance >= 0.0f;
		const bool use_sphere = a && (!b || sphere_distance <= plane_distance);
		const bool use_plane = !use_sphere && b;
		distance = use_sphere ? sphere_distance : (use_plane ? plane_distance : -1.0f);
		index = use_sphere ? spheres.index : planes.index;
		return use_sphere ? 1u : (use_plane ? 2u : 0u);
	}

	// lookups of the winning primitive from the global per-primitive tables (one indexed read each)
	template <bool SM>
	__device__ __forceinline__ void
	fetch_hit(const device_scene& s, vec3 o, vec3 d, uint32_t kind, float distance, uint32_t index, vec3& normal, float4& shading, uint32_t& scatter)
	{
		scatter = scatter_lambert;
		if (kind) // normal and shading are meaningful only for a hit
		{
			const uint32_t primitive = kind == 1u ? index : s.n_spheres + index;
			const float4 g = s.primitive_geometry[primitive];
			shading = SM ? s.primitive_shading_sm[primitive] : s.primitive_shading[primitive];
			scatter = SM ? s.primitive_scatter_sm[primitive] : s.primitive_scatter[primitive];
			if (kind == 1u)
				normal = normalize(ray_at(o, d, distance) - vec3{ g.x, g.y, g.z }); // (:85)
			else
				normal = { g.x, g.y, g.z }; // the plane's normal, not flipped toward the ray (:58)
		}
	}
}
