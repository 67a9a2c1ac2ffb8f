// oracle/cpu_ref.cpp — CPU restatement of marzer/rt's mg_ray_tracer hot path (the parity oracle).
//
// TEST INFRASTRUCTURE, NOT PRODUCT (see cpu_ref.h).  PARITY UNPINNED (see cpu_ref.h).
//
// Follows, function by function (paths relative to the reference root):
//   render / worker ............ src/renderers/mg_ray_tracer.cpp:178-205
//   trace ....................... src/renderers/mg_ray_tracer.cpp:155-174
//   test_planes / test_spheres .. src/renderers/mg_ray_tracer.cpp:36-87
//   test_boxes / select ......... src/renderers/mg_ray_tracer.cpp:89-102
//   lambert / metal scatter ..... src/renderers/mg_ray_tracer.cpp:110-140, src/common.hpp:100-103
//   screen_to_world ............. src/camera.hpp:42-48
//   random<T>, unit vector ...... src/random.hpp:37-66   (engine replaced: see "random streams")
//   colour pack ................. src/colour.hpp:63-65,101-106
//   pixel addressing ............ src/image.hpp:143-159
//
// The vector/ray arithmetic itself lives in marzer/muu (absent offline).  The formulas chosen for it are
// SURVEY.md §8c's; the OPERATION ORDER below ("arithmetic contract v4": v1's floating-point rules + per-pixel keyed
// random streams (v2) + normalize()'s reciprocal square root in one step (v3) + one generator step per random<vecN>()
// and primary rays from a per-pixel base (v4)) is this project's own and is what the GPU kernels reproduce bit for
// bit.  The reference is built with -ffast-math -ffp-contract=fast (meson.build:153-160), so it defines no operation
// order of its own, and its generator (std::mt19937 seeded from std::random_device, src/random.cpp:9-26) pins nothing
// about the random numbers but that they are uniform and independent.
//
// Arithmetic contract — every value is IEEE-754 binary32, round-to-nearest-even, subnormals kept;
// no contraction or reassociation except the fmaf() written out here; sqrtf and '/' are correctly rounded:
//   dot(a,b)          = fmaf(a.z,b.z, fmaf(a.y,b.y, a.x*b.x))
//   normalize(v)      = v * inv_sqrt(dot(v,v))                  (inv_sqrt below; three products)
//   direction(a,b)    = normalize(b - a)
//   at(ray,t)         = fmaf(d, t, o) per component
//   lerp(a,b,t)       = fmaf(b - a, t, a) per component
//   transform_position(M,v): row_r = fmaf(M[r][0],v.x, fmaf(M[r][1],v.y, fmaf(M[r][2],v.z, M[r][3])));
//                            xyz = row_0..2 * (1.0f / row_3)       (the preview, and primary rays of a matrix with varying w)
//   primary ray (v4)       : PINHOLE camera (rt's: w constant over the frame and all near-to-far lines through one
//                            eye point; constants in binary64, make_frame): the near-to-far vector is affine in the
//                            pixel position — evaluated once per PIXEL at its corner, base_c = fmaf(d1_c, x,
//                            fmaf(d2_c, y, d0_c)), and moved per SAMPLE by the jitter's numerators (ka, kb; u = k * 2^-24):
//                            toward_c = fmaf(j1_c, ka, fmaf(j2_c, kb, base_c)), j = d * 2^-24; dir = normalize(toward);
//                            origin_c = eye_c + toward_c — the vector carried is kappa * (far - near) = near - eye, the
//                            near point lying on the line from the eye at the fixed fraction kappa = near / (far - near).
//                            A PERSPECTIVE matrix that is no pinhole's in binary32 (a tilted camera: rounding noise in its
//                            w row): all near-to-far lines pass through the eye E = Z.xyz / Z.w (Z = the depth column);
//                            N' = N.xyz - E N.w, the homogeneous near point relative to it, is affine in the pixel
//                            position like the pinhole's vector (per-pixel base + jitter share, binary64 constants), and
//                            so is N.w: toward = s N' (negated if N.w F.w < 0), origin_c = fmaf(toward_c, 1.0f / (s N.w), E_c),
//                            s = sign(-Z.w).  One division per sample, no far point, no cancellation.
//                            ANY OTHER matrix: homogeneous near / far points N, F (four fmaf rows each, as
//                            transform_position); origin = N.xyz * (1.0f / N.w); toward_c = fmaf(F_c, N.w, -(N_c * F.w)),
//                            negated if N.w * F.w < 0 — far/F.w - near/N.w times the positive factor |N.w F.w|, which
//                            normalize() removes: one division per sample instead of two.
//   ray-plane distance      : t = -(n.o + d) * (1.0f / (n.dir)) — the reciprocal correctly rounded, then one product (the
//                            reference's build lets its compiler do the same: -ffast-math implies -freciprocal-math).
//   pixel sum               : samples are added in CHUNKS of 16 consecutive samples (each chunk summed in sample
//                            order, starting from 0), and the chunk sums are added in chunk order (left fold starting
//                            from the first chunk's sum).  For spp <= 16 this is the reference's plain sequential
//                            `colour += trace(...)` (mg_ray_tracer.cpp:187-194); for larger spp it is the same sum
//                            associated differently — the unit of work a GPU lane takes is one chunk.
//   everything else is written out where it is used.
//
// Build: -O2 -ffp-contract=off (no -ffast-math); -mfma only makes fmaf() a single instruction.

#include "cpu_ref.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace
{
	struct vec3
	{
		float x, y, z;
	};

	inline vec3 operator+(vec3 a, vec3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
	inline vec3 operator-(vec3 a, vec3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
	inline vec3 operator*(vec3 a, vec3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
	inline vec3 operator*(vec3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }

	inline float dot(vec3 a, vec3 b) { return std::fmaf(a.z, b.z, std::fmaf(a.y, b.y, a.x * b.x)); }

	// Contract v3: normalize()'s reciprocal square root is ONE Newton-Raphson step in binary32, residual computed exactly,
	// from 1/sqrt(x) truncated toward zero (taken through binary64: a deterministic IEEE expression, no hardware estimate).
	// It is the correctly rounded 1/sqrt(x) for every significand but x = 4^k * (1 - 2^-23), which gives 2^-k
	// (tests/test_oracle_kat.py checks both statements over every significand).  muu's normalize() is not readable here
	// (call sites: mg_ray_tracer.cpp:85,120,138; random.hpp:64) and the reference is built -ffast-math anyway.
	inline float inv_sqrt_step(float x, float y) // the Newton-Raphson step from the estimate y
	{
		const float t = x * y;
		const float dt = std::fmaf(x, y, -t); // t + dt == x * y exactly
		const float e = std::fmaf(-dt, y, std::fmaf(-t, y, 1.0f)); // the residual 1 - x*y*y
		return std::fmaf(0.5f * y, e, y);
	}

	inline float inv_sqrt(float x)
	{
		const double exact = 1.0 / std::sqrt(static_cast<double>(x));
		float y = static_cast<float>(exact);
		uint32_t bits;
		std::memcpy(&bits, &x, 4);
		if (!((bits - 0x00800000u) < (0x7F800000u - 0x00800000u)))
			return y; // zero, subnormal, negative, infinite, NaN: the plain quotient
		if (static_cast<double>(y) > exact) // toward zero
		{
			uint32_t yb;
			std::memcpy(&yb, &y, 4);
			yb -= 1u;
			std::memcpy(&y, &yb, 4);
		}
		return inv_sqrt_step(x, y);
	}

	inline vec3 normalize(vec3 v)
	{
		const float inv = inv_sqrt(dot(v, v));
		return v * inv;
	}

	inline vec3 direction(vec3 from, vec3 to) { return normalize(to - from); }

	struct ray
	{
		vec3 origin;
		vec3 dir;

		vec3 at(float t) const
		{
			return { std::fmaf(dir.x, t, origin.x), std::fmaf(dir.y, t, origin.y), std::fmaf(dir.z, t, origin.z) };
		}
	};

	// ---- random streams -------------------------------------------------------------------------------------
	// The reference draws from a thread_local std::mt19937 seeded by std::random_device (src/random.cpp:9-26):
	// not reproducible, and tied to which host thread ran the pixel.  Replaced by counter-based streams in which
	// draw k of a sample is a pure function of (seed, GLOBAL pixel index, sample index, k).  Contract v2: every pixel
	// of a frame draws through its OWN keyed hash function, walked along its OWN arithmetic progression of counters
	// (64 bits of key per pixel), as the reference's generators are independent of one another:
	//   hash32   = the "lowbias32" integer finaliser (xorshift-multiply, two rounds)
	//   (fa, fb) = low and high half of mix64(seed), mix64 = the splitmix64 finaliser (a bijection of 64-bit words)
	//   k        = hash32(pixel_index ^ fa)          function key — a bijection of the pixel index: never shared in a frame
	//   stride   = hash32(k ^ fb) | 1                counter stride (odd)
	//   counter  = stride * (sample_index * 4096)    before the first step of a sample (mod 2^32; 4096 steps reserved per sample)
	//   step     : counter += stride;  x = counter;  x ^= x >> 16;  x = x * 0x7feb352d + k;  x ^= x >> 15;
	//              a = x * 0x846ca68b;  b = x * MB;  c = x * MC     (mod 2^32)
	//              u_a = float(a >> 8) * 2^-24, u_b, u_c likewise   in [0, 1)
	//              (the top 24 bits of the products as they are; lowbias32's closing xorshift would only touch a
	//              draw's lowest 8 bits)
	// Contract v4: ONE step serves one call of random<T>() — random<float>() takes u_a, random<vec2>() (u_a, u_b),
	// random<vec3>() (u_a, u_b, u_c), in the reference's brace-init order (src/random.hpp:37-46).  Under v2/v3 every
	// component was a step of its own: three hashes per unit vector, two per jitter — a fifth of the small kernel's
	// vector instructions (profiles/r04/headline_basic_1080p_256spp/pmc_summary.csv).  The three words of a step are
	// the multiples x * M2 * (1, A, A^2) of the mixed word x, with M2 = lowbias32's second multiplier and
	// A = 0xadb4a92d (Steele & Vigna, "Computationally easy, spectrally good multipliers", the 32-bit LCG multiplier):
	// MB = M2 * A, MC = M2 * A^2 (mod 2^32).  Seen together they are a point of the rank-1 lattice generated by
	// (1, A, A^2) / 2^32 — what three consecutive outputs of that congruential generator are — picked by a hashed index;
	// its figures of merit are 0.976 in two and 0.936 in three dimensions (tools/rng_lattice.py), i.e. 2^32 possible
	// unit-cube points about 1/1700 apart in every direction.  u_a alone is exactly the v2/v3 draw.
	// Two pixels with different strides evaluate their (different) functions at a common counter only at isolated
	// steps, never along a run.  (Contract v1 drew every pixel from ONE shared hash32 sequence at a hashed offset; a
	// 1920x1080x256 frame draws 3.6e9 numbers, so most sample windows overlapped another pixel's and some were identical.)
	inline uint32_t hash32(uint32_t x)
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

	inline frame_keys make_frame_keys(uint64_t seed)
	{
		uint64_t z = seed + 0x9E3779B97F4A7C15ull;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		z ^= z >> 31;
		return { static_cast<uint32_t>(z), static_cast<uint32_t>(z >> 32) };
	}

	struct random_stream
	{
		uint32_t function_key;
		uint32_t stride;
		uint32_t counter;

		random_stream(frame_keys frame, uint32_t pixel_index, uint32_t sample_index)
			: function_key{ hash32(pixel_index ^ frame.a) },
			  stride{ hash32(function_key ^ frame.b) | 1u },
			  counter{ stride * (sample_index << 12) }
		{}

		// one generator step: the numerators k (0 <= k < 2^24) of the three draws u = k * 2^-24
		struct step
		{
			uint32_t a, b, c;
		};
		static constexpr uint32_t mul_a = 0x846ca68bu;					   // lowbias32's second multiplier
		static constexpr uint32_t lattice = 0xadb4a92du;				   // A
		static constexpr uint32_t mul_b = mul_a * lattice;				   // (mod 2^32)
		static constexpr uint32_t mul_c = mul_a * lattice * lattice;	   // (mod 2^32)
		step next_step()
		{
			counter += stride;
			uint32_t x = counter;
			x ^= x >> 16;
			x = x * 0x7feb352du + function_key;
			x ^= x >> 15;
			return { (x * mul_a) >> 8, (x * mul_b) >> 8, (x * mul_c) >> 8 };
		}

		// random<float>(), src/random.hpp:12-17 / src/random.cpp:20-26: uniform in [0, 1)
		float next() { return static_cast<float>(next_step().a) * 0x1.0p-24f; }
	};

	// random_unit_vector(), src/random.hpp:57-66: components x, y, z of random<vec3>() (brace-init order, :43-46: one
	// step under contract v4), redrawn only if exactly zero, then normalised.  All components are >= 0: the direction
	// lies in the positive octant.
	inline vec3 random_unit_vector(random_stream& rng)
	{
		while (true)
		{
			const random_stream::step k = rng.next_step();
			const float x = static_cast<float>(k.a) * 0x1.0p-24f;
			const float y = static_cast<float>(k.b) * 0x1.0p-24f;
			const float z = static_cast<float>(k.c) * 0x1.0p-24f;
			if (x == 0.0f && y == 0.0f && z == 0.0f)
				continue;
			return normalize({ x, y, z });
		}
	}

	// ---- scene access -----------------------------------------------------------------------------------------
	struct material
	{
		uint32_t type;
		vec3 attenuation; // vec3{ albedo * reflectivity }, mg_ray_tracer.cpp:115,131 (colour * float, colour.hpp:144-149)
		float roughness;
		float reflectivity; // doubles as the index of refraction in sm_ray_tracer's dielectric_scatter
	};

	struct frame
	{
		const rt_hip_scene* scene;
		std::vector<material> materials;
		uint32_t width, height;
		float sx, sy; // 2 / W, 2 / H
		// contract v4: primary rays of a pinhole camera (rt's: camera.hpp:122-137) — the vector from the eye to the near
		// point, kappa * (far - near), as an affine function of the pixel position (d0 + d1 x + d2 y), the jitter's share of
		// it per numerator (j = d * 2^-24), and the eye; constants worked out in binary64
		bool pinhole_rays;
		float ray_d0[3], ray_d1[3], ray_d2[3], ray_j1[3], ray_j2[3], ray_eye[3];
		// a perspective matrix that is no pinhole's in binary32 (a tilted camera: its w row carries rounding noise): every
		// near-to-far line still passes through ONE point, the eye E = Z.xyz / Z.w (Z = the matrix's depth column), and the
		// homogeneous near point relative to it, N' = N.xyz - E N.w, is affine in the pixel position: the near-to-far
		// vector is N' (times a sign), the near point E + N' / N.w.  Constants in binary64, as above.
		bool eye_rays;
		float eye_q0[3], eye_q1[3], eye_q2[3], eye_jq1[3], eye_jq2[3]; // s N' = q0 + q1 x + q2 y; its share per jitter numerator
		float eye_w0, eye_w1, eye_w2, eye_jw1, eye_jw2;				  // s N.w likewise                       (s = sign(-Z.w))
		float eye_e[3], eye_zws;										  // E;  s Z.w
		// any other matrix: rows of the homogeneous near / far points
		float mx[4], my[4], k_near[4], k_far[4];
		frame_keys keys;
		int trace_order;
		bool sm_materials;
	};

	inline frame make_frame(const rt_hip_scene* s, uint32_t w, uint32_t h, uint64_t seed, int mode)
	{
		frame f{};
		f.scene = s;
		f.width = w;
		f.height = h;
		f.sx = 2.0f / static_cast<float>(w);
		f.sy = 2.0f / static_cast<float>(h);
		{
			// viewport::screen_to_world (camera.hpp:42-48) un-projects ndc = (2x/W - 1, -2y/H + 1, depth) through the inverse
			// view-projection M and divides by w.  For rt's camera the last row of M has no x and no y term, so w depends on
			// the depth alone: near(px, py) = (M0 X + M1 Y + k_near) / w_near is affine in the pixel position, and so is
			// far - near.  Its frustum is a pinhole's: every near-to-far line passes through the eye, so that
			// near = eye + kappa * (far - near) with ONE kappa = n / (f - n) for the whole frame.  The constants are worked out
			// here in binary64, in this order of operations, and rounded to binary32 once (rt_amd/csrc/render.hip has the same
			// lines).  Whether a matrix IS a pinhole's is decided from the same numbers: w constant over the frame, and the
			// near point's motion per pixel within 1e-5 (relative) of kappa times the near-to-far vector's — a slack of
			// 1e-10 of a pixel step, far below what binary32 resolves; an orthographic or sheared frustum fails it and takes the
			// general form, like a matrix whose w varies.
			const float* M = s->inverse_view_projection;
			for (int r = 0; r < 4; r++)
			{
				f.mx[r] = M[r * 4 + 0];
				f.my[r] = M[r * 4 + 1];
				f.k_near[r] = std::fmaf(M[r * 4 + 2], 0.0f, M[r * 4 + 3]);
				f.k_far[r] = std::fmaf(M[r * 4 + 2], 1.0f, M[r * 4 + 3]);
			}
			const float* k_near = f.k_near;
			const float* k_far = f.k_far;
			f.eye_rays = false;
			{
				// N_r(px, py) = mx_r X + my_r Y + k_near_r with X = (2/W) px - 1, Y = -(2/H) py + 1  =  n0_r + n1_r px + n2_r py
				const double sx = 2.0 / static_cast<double>(w), sy = -(2.0 / static_cast<double>(h));
				const double zw = M[3 * 4 + 2];
				const double e[3] = { M[0 * 4 + 2] / zw, M[1 * 4 + 2] / zw, M[2 * 4 + 2] / zw };
				const double sign = zw < 0.0 ? 1.0 : -1.0; // sign(-Z.w): s N' points from near to far where N.w F.w > 0
				const double n1w = static_cast<double>(M[12]) * sx, n2w = static_cast<double>(M[13]) * sy;
				const double n0w = static_cast<double>(k_near[3]) - static_cast<double>(M[12]) + static_cast<double>(M[13]);
				bool finite = zw != 0.0 && std::isfinite(e[0]) && std::isfinite(e[1]) && std::isfinite(e[2]) && std::isfinite(n0w) && std::isfinite(n1w) && std::isfinite(n2w);
				for (int c = 0; c < 3 && finite; c++)
				{
					const double mx = M[c * 4 + 0], my = M[c * 4 + 1];
					const double n1 = mx * sx, n2 = my * sy, n0 = static_cast<double>(k_near[c]) - mx + my;
					f.eye_q0[c] = static_cast<float>(sign * (n0 - e[c] * n0w)), f.eye_q1[c] = static_cast<float>(sign * (n1 - e[c] * n1w)), f.eye_q2[c] = static_cast<float>(sign * (n2 - e[c] * n2w));
					f.eye_jq1[c] = f.eye_q1[c] * 0x1.0p-24f, f.eye_jq2[c] = f.eye_q2[c] * 0x1.0p-24f;
					f.eye_e[c] = static_cast<float>(e[c]);
					finite = std::isfinite(f.eye_q0[c]) && std::isfinite(f.eye_q1[c]) && std::isfinite(f.eye_q2[c]);
				}
				if (finite)
				{
					f.eye_w0 = static_cast<float>(sign * n0w), f.eye_w1 = static_cast<float>(sign * n1w), f.eye_w2 = static_cast<float>(sign * n2w);
					f.eye_jw1 = f.eye_w1 * 0x1.0p-24f, f.eye_jw2 = f.eye_w2 * 0x1.0p-24f;
					f.eye_zws = static_cast<float>(sign * zw);
					f.eye_rays = true;
				}
			}
			f.pinhole_rays = false;
			if (M[12] == 0.0f && M[13] == 0.0f && k_near[3] != 0.0f && k_far[3] != 0.0f && std::isfinite(k_near[3]) && std::isfinite(k_far[3]))
			{
				const double sx = 2.0 / static_cast<double>(w), sy = -(2.0 / static_cast<double>(h));
				const double iwn = 1.0 / static_cast<double>(k_near[3]), iwf = 1.0 / static_cast<double>(k_far[3]);
				double o0[3], o1[3], o2[3], d0[3], d1[3], d2[3];
				for (int c = 0; c < 3; c++)
				{
					const double mx = M[c * 4 + 0], my = M[c * 4 + 1], kn = k_near[c], kf = k_far[c];
					o1[c] = mx * sx * iwn, o2[c] = my * sy * iwn, o0[c] = (kn - mx + my) * iwn;
					const double e1 = mx * sx * iwf, e2 = my * sy * iwf, e0 = (kf - mx + my) * iwf;
					d0[c] = e0 - o0[c], d1[c] = e1 - o1[c], d2[c] = e2 - o2[c];
				}
				const double dd = d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2] + d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2];
				const double od = o1[0] * d1[0] + o1[1] * d1[1] + o1[2] * d1[2] + o2[0] * d2[0] + o2[1] * d2[1] + o2[2] * d2[2];
				const double kappa = od / dd;
				double worst = 0.0, scale = 0.0;
				for (int c = 0; c < 3; c++)
				{
					worst = std::fmax(worst, std::fmax(std::fabs(o1[c] - kappa * d1[c]), std::fabs(o2[c] - kappa * d2[c])));
					scale = std::fmax(scale, std::fmax(std::fabs(o1[c]), std::fabs(o2[c])));
				}
				if (dd > 0.0 && kappa >= 0x1.0p-20 && kappa <= 0x1.0p20 && worst <= 1.0e-5 * scale) // (a NaN anywhere fails a comparison; kappa = near / (far - near) scales the vector the kernels normalise, so its sign and size matter)
				{
					f.pinhole_rays = true;
					for (int c = 0; c < 3; c++)
					{
						// (the vector the kernels carry is kappa * (far - near) = near - eye: the near point is then eye + it, one addition)
					f.ray_d0[c] = static_cast<float>(kappa * d0[c]), f.ray_d1[c] = static_cast<float>(kappa * d1[c]), f.ray_d2[c] = static_cast<float>(kappa * d2[c]);
						f.ray_j1[c] = f.ray_d1[c] * 0x1.0p-24f, f.ray_j2[c] = f.ray_d2[c] * 0x1.0p-24f;
						f.ray_eye[c] = static_cast<float>(o0[c] - kappa * d0[c]);
					}
				}
			}
		}
		f.keys = make_frame_keys(seed);
		f.trace_order = mode & ORACLE_TRACE_RECURSIVE;
		f.sm_materials = (mode & ORACLE_MATERIALS_SM) != 0;
		f.materials.resize(s->n_materials);
		for (uint32_t m = 0; m < s->n_materials; m++)
		{
			const float refl = s->material_reflectivity[m];
			f.materials[m].type = s->material_type[m];
			f.materials[m].attenuation = { s->material_albedo[m * 4 + 0] * refl,
										   s->material_albedo[m * 4 + 1] * refl,
										   s->material_albedo[m * 4 + 2] * refl };
			f.materials[m].roughness = s->material_roughness[m];
			f.materials[m].reflectivity = refl;
		}
		return f;
	}

	// ---- camera: viewport::screen_to_world, src/camera.hpp:42-48 -------------------------------------------------
	inline vec3 transform_position(const float* M, vec3 v)
	{
		float row[4];
		for (int r = 0; r < 4; r++)
			row[r] = std::fmaf(M[r * 4 + 0], v.x, std::fmaf(M[r * 4 + 1], v.y, std::fmaf(M[r * 4 + 2], v.z, M[r * 4 + 3])));
		const float inv_w = 1.0f / row[3];
		return { row[0] * inv_w, row[1] * inv_w, row[2] * inv_w };
	}

	inline vec3 screen_to_world(const frame& f, float px, float py, float depth)
	{
		// { 2*(x/W) - 1, -2*(y/H) + 1, depth }, with 2/W and 2/H hoisted per frame
		const vec3 ndc = { std::fmaf(px, f.sx, -1.0f), std::fmaf(py, -f.sy, 1.0f), depth };
		return transform_position(f.scene->inverse_view_projection, ndc);
	}

	// The primary ray of pixel (x, y) whose jitter is (ka, kb) * 2^-24 (numerators of one generator step; 2^23 each for the
	// centre): mg_ray_tracer.cpp:189-193.
	inline ray primary_ray(const frame& f, uint32_t x, uint32_t y, float ka, float kb)
	{
		const float fx = static_cast<float>(x), fy = static_cast<float>(y);
		if (f.pinhole_rays) // near-to-far vector from the pixel's base and the jitter, near point from it (see make_frame)
		{
			vec3 toward, origin;
			float* const t = &toward.x;
			float* const o = &origin.x;
			for (int c = 0; c < 3; c++)
			{
				const float base = std::fmaf(f.ray_d1[c], fx, std::fmaf(f.ray_d2[c], fy, f.ray_d0[c])); // once per pixel
				t[c] = std::fmaf(f.ray_j1[c], ka, std::fmaf(f.ray_j2[c], kb, base));
				o[c] = f.ray_eye[c] + t[c];
			}
			return { origin, normalize(toward) }; // :190-193
		}
		if (f.eye_rays) // a perspective matrix: the near point relative to the eye, and the eye plus that over N.w
		{
			float t[3];
			for (int c = 0; c < 3; c++)
			{
				const float base = std::fmaf(f.eye_q1[c], fx, std::fmaf(f.eye_q2[c], fy, f.eye_q0[c])); // once per pixel
				t[c] = std::fmaf(f.eye_jq1[c], ka, std::fmaf(f.eye_jq2[c], kb, base));
			}
			const float w_base = std::fmaf(f.eye_w1, fx, std::fmaf(f.eye_w2, fy, f.eye_w0)); // once per pixel
			const float ws = std::fmaf(f.eye_jw1, ka, std::fmaf(f.eye_jw2, kb, w_base));		 // s N.w
			const float inv = 1.0f / ws;
			const vec3 near_pos = { std::fmaf(t[0], inv, f.eye_e[0]), std::fmaf(t[1], inv, f.eye_e[1]), std::fmaf(t[2], inv, f.eye_e[2]) }; // :190
			// far - near (:191,193) = N' (-Z.w) / (N.w F.w) = (s N') |Z.w| / (N.w F.w): s N', with the sign of N.w F.w = (s N.w)(s N.w + s Z.w)
			vec3 toward = { t[0], t[1], t[2] };
			if (ws * (ws + f.eye_zws) < 0.0f)
				toward = { -toward.x, -toward.y, -toward.z };
			return { near_pos, normalize(toward) };
		}
		// any other matrix (no finite eye: an orthographic frustum): screen_to_world (camera.hpp:42-48) for depth 0 and 1 in
		// homogeneous form; ONE division
		const float px = std::fmaf(ka, 0x1.0p-24f, fx), py = std::fmaf(kb, 0x1.0p-24f, fy); // == fx + ka * 2^-24: the product is exact
		const float ndc_x = std::fmaf(px, f.sx, -1.0f), ndc_y = std::fmaf(py, -f.sy, 1.0f);
		float N[4], F[4];
		for (int r = 0; r < 4; r++)
		{
			N[r] = std::fmaf(f.mx[r], ndc_x, std::fmaf(f.my[r], ndc_y, f.k_near[r]));
			F[r] = std::fmaf(f.mx[r], ndc_x, std::fmaf(f.my[r], ndc_y, f.k_far[r]));
		}
		const float inv_wn = 1.0f / N[3];
		const vec3 near_pos = { N[0] * inv_wn, N[1] * inv_wn, N[2] * inv_wn }; // :190
		// far / F.w - near / N.w (:191,193) = (F N.w - N F.w) / (N.w F.w): the numerator, with the denominator's sign
		vec3 toward = { std::fmaf(F[0], N[3], -(N[0] * F[3])), std::fmaf(F[1], N[3], -(N[1] * F[3])), std::fmaf(F[2], N[3], -(N[2] * F[3])) };
		if (N[3] * F[3] < 0.0f)
			toward = { -toward.x, -toward.y, -toward.z };
		return { near_pos, normalize(toward) };
	}

	// ---- intersection (muu::ray::hits; formulas per SURVEY.md §8c) ------------------------------------------------
	constexpr float min_hit_dist = 0.001f; // mg_ray_tracer.cpp:20
	constexpr uint32_t sample_chunk = 16;  // samples per chunk of the pixel sum (arithmetic contract)
	constexpr float approx_zero_epsilon = 1.0e-6f;

	// returns false = no hit (std::nullopt in muu)
	inline bool hits_sphere(const ray& r, vec3 center, float radius, float& t)
	{
		const vec3 e = center - r.origin;
		const float a = dot(e, r.dir);
		const float e2 = dot(e, e);
		const float r2 = radius * radius;
		const float disc = r2 - std::fmaf(-a, a, e2); // r^2 - (e^2 - a^2)
		if (disc < 0.0f)
			return false;
		const float f = std::sqrt(disc);
		t = (e2 < r2) ? a + f : a - f; // origin inside: far root; outside: near root
		if (t < 0.0f)
			return false;
		return true;
	}

	// plane: n·p + d = 0
	inline bool hits_plane(const ray& r, vec3 n, float d, float& t)
	{
		const float den = dot(n, r.dir);
		if (std::fabs(den) <= approx_zero_epsilon)
			return false;
		const float num = dot(n, r.origin) + d;
		t = (-num) * (1.0f / den);
		if (t < 0.0f)
			return false;
		return true;
	}

	// Axis-aligned box, slab method (muu::ray::hits(bounding_box): restated; corners = center -/+ extents).  Only the
	// preview draws boxes.  Contract: reciprocal direction by IEEE division, compare-and-select minima / maxima (a NaN
	// from 0 * inf that survives them makes `tmax >= tmin` false = miss), origin inside -> the exit distance.
	inline float select_min(float a, float b) { return a < b ? a : b; }
	inline float select_max(float a, float b) { return a > b ? a : b; }
	inline bool hits_box(const ray& r, vec3 lo, vec3 hi, float& t)
	{
		const vec3 inv = { 1.0f / r.dir.x, 1.0f / r.dir.y, 1.0f / r.dir.z };
		const vec3 t1 = { (lo.x - r.origin.x) * inv.x, (lo.y - r.origin.y) * inv.y, (lo.z - r.origin.z) * inv.z };
		const vec3 t2 = { (hi.x - r.origin.x) * inv.x, (hi.y - r.origin.y) * inv.y, (hi.z - r.origin.z) * inv.z };
		const float tmin = select_max(select_max(select_min(t1.x, t2.x), select_min(t1.y, t2.y)), select_min(t1.z, t2.z));
		const float tmax = select_min(select_min(select_max(t1.x, t2.x), select_max(t1.y, t2.y)), select_max(t1.z, t2.z));
		if (!(tmax >= tmin) || tmax < 0.0f)
			return false;
		t = tmin >= 0.0f ? tmin : tmax;
		return true;
	}

	struct hit_result // mg_ray_tracer.cpp:22-33
	{
		float distance;
		vec3 normal;
		uint32_t material;
		uint32_t kind; // 0 none, 1 sphere, 2 plane (diagnostic only)
		uint32_t index;

		explicit operator bool() const { return distance >= 0.0f; }
	};

	constexpr hit_result no_hit = { -1.0f, { 0, 0, 0 }, 0, 0, 0 };

	// mg_ray_tracer.cpp:36-60
	inline hit_result test_planes(const rt_hip_scene& s, const ray& r)
	{
		bool have = false;
		uint32_t hit_index = 0;
		float hit_dist = 0.0f;
		for (uint32_t i = 0; i < s.n_planes; i++)
		{
			float t;
			const bool hit = hits_plane(r, { s.plane_normal_x[i], s.plane_normal_y[i], s.plane_normal_z[i] }, s.plane_d[i], t);
			if (!hit || t < min_hit_dist || (have && hit_dist <= t))
				continue;
			have = true;
			hit_index = i;
			hit_dist = t;
		}
		if (!have)
			return no_hit;
		return { hit_dist,
				 { s.plane_normal_x[hit_index], s.plane_normal_y[hit_index], s.plane_normal_z[hit_index] },
				 s.plane_material[hit_index],
				 2u,
				 hit_index };
	}

	// mg_ray_tracer.cpp:63-87
	inline hit_result test_spheres(const rt_hip_scene& s, const ray& r)
	{
		bool have = false;
		uint32_t hit_index = 0;
		float hit_dist = 0.0f;
		for (uint32_t i = 0; i < s.n_spheres; i++)
		{
			float t;
			const bool hit =
				hits_sphere(r, { s.sphere_center_x[i], s.sphere_center_y[i], s.sphere_center_z[i] }, s.sphere_radius[i], t);
			if (!hit || t < min_hit_dist || (have && hit_dist <= t))
				continue;
			have = true;
			hit_index = i;
			hit_dist = t;
		}
		if (!have)
			return no_hit;
		const vec3 center = { s.sphere_center_x[hit_index], s.sphere_center_y[hit_index], s.sphere_center_z[hit_index] };
		return { hit_dist, direction(center, r.at(hit_dist)), s.sphere_material[hit_index], 1u, hit_index };
	}

	// mg_ray_tracer.cpp:96-102
	inline hit_result select(const hit_result& a, const hit_result& b)
	{
		if (!a)
			return b;
		return (!b || a.distance <= b.distance) ? a : b;
	}

	inline hit_result closest_hit(const rt_hip_scene& s, const ray& r)
	{
		hit_result hit = test_planes(s, r);	  // :160
		hit = select(test_spheres(s, r), hit); // :161  (sphere wins a tie)
		hit = select(no_hit, hit);			  // :162  test_boxes always misses (:89-93)
		return hit;
	}

	// ---- shading ---------------------------------------------------------------------------------------------
	// mg_ray_tracer.cpp:164 — lerp(white, (0.5, 0.7, 1.0), 0.5 * (dir.y + 1))
	inline vec3 sky(float dir_y)
	{
		const float t = 0.5f * (dir_y + 1.0f);
		const vec3 a = { 1.0f, 1.0f, 1.0f };
		const vec3 b = { 0.5f, 0.7f, 1.0f };
		return { std::fmaf(b.x - a.x, t, a.x), std::fmaf(b.y - a.y, t, a.y), std::fmaf(b.z - a.z, t, a.z) };
	}

	// mg_ray_tracer.cpp:110-123
	inline bool lambert_scatter(const ray& r, const hit_result& hit, random_stream& rng, ray& out)
	{
		vec3 scatter = hit.normal + random_unit_vector(rng);
		if (std::fabs(scatter.x) <= approx_zero_epsilon && std::fabs(scatter.y) <= approx_zero_epsilon
			&& std::fabs(scatter.z) <= approx_zero_epsilon)
			scatter = hit.normal;
		out = { r.at(hit.distance), normalize(scatter) };
		return true;
	}

	// mg_ray_tracer.cpp:126-140; reflect = v - 2*dot(v,n)*n, src/common.hpp:100-103
	inline bool metal_scatter(const ray& r, const hit_result& hit, float roughness, random_stream& rng, ray& out)
	{
		const vec3 v = normalize(r.dir);
		const float k = 2.0f * dot(v, hit.normal);
		const vec3 reflected = { std::fmaf(-k, hit.normal.x, v.x), std::fmaf(-k, hit.normal.y, v.y), std::fmaf(-k, hit.normal.z, v.z) };
		const vec3 u = random_unit_vector(rng);
		const vec3 scatter = { std::fmaf(roughness, u.x, reflected.x),
							   std::fmaf(roughness, u.y, reflected.y),
							   std::fmaf(roughness, u.z, reflected.z) };
		if (dot(scatter, hit.normal) <= 0.0f)
			return false;
		out = { r.at(hit.distance), normalize(scatter) };
		return true;
	}

	// sm_ray_tracer.cpp:181-219 (with refract :161-172 and schlick :174-179), opt-in.
	// `refl` = the material's reflectivity column, used by the reference as the index of refraction.
	// Contract details: sin2_t = (eta*eta) * fma(-cos_i, cos_i, 1); cos_t = sqrtf(1 - sin2_t);
	// refracted = fma(k, n, eta*v) with k = fma(eta, cos_i, -cos_t); schlick's fifth power is taken in double by
	// three multiplications ((x*x)*(x*x))*x of x = (double)(1.0f - cosine) — the reference calls pow(double, int) —
	// and r0 + (1 - r0) * x^5 is evaluated in double with (1 - r0) in float, as the reference's promotions do.
	// The new direction is NOT normalised (the reference returns `reflected` / `refracted` as they are).
	inline vec3 dielectric_direction(vec3 d, vec3 n, float refl, float u, float* out_reflect_prob = nullptr);

	inline bool dielectric_scatter(const ray& r, const hit_result& hit, float refl, random_stream& rng, ray& out)
	{
		// argument evaluation order does not matter: exactly one draw
		out = { r.at(hit.distance), dielectric_direction(r.dir, hit.normal, refl, rng.next()) };
		return true;
	}

	inline vec3 dielectric_direction(vec3 d, vec3 n, float refl, float u, float* out_reflect_prob)
	{
		const float dn = dot(d, n);
		const float k = 2.0f * dn;
		const vec3 reflected = { std::fmaf(-k, n.x, d.x), std::fmaf(-k, n.y, d.y), std::fmaf(-k, n.z, d.z) };
		const float len = std::sqrt(dot(d, d));
		vec3 outward;
		float eta, cosine;
		if (dn > 0.0f)
		{
			outward = { -n.x, -n.y, -n.z };
			eta = refl;
			cosine = (refl * dn) / len;
		}
		else
		{
			outward = n;
			eta = 1.0f / refl;
			cosine = (-dn) / len;
		}
		float reflect_prob = 1.0f;
		vec3 refracted = { 0, 0, 0 };
		const float cos_i = -dot(d, outward);
		const float sin2_t = (eta * eta) * std::fmaf(-cos_i, cos_i, 1.0f);
		if (!(sin2_t > 1.0f)) // refract() returned true
		{
			const float cos_t = std::sqrt(1.0f - sin2_t);
			const float kk = std::fmaf(eta, cos_i, -cos_t);
			refracted = { std::fmaf(kk, outward.x, eta * d.x), std::fmaf(kk, outward.y, eta * d.y), std::fmaf(kk, outward.z, eta * d.z) };
			float r0 = (1.0f - refl) / (1.0f + refl);
			r0 = r0 * r0;
			const double x = static_cast<double>(1.0f - cosine);
			const double x2 = x * x;
			const double x5 = (x2 * x2) * x;
			reflect_prob = static_cast<float>(static_cast<double>(r0) + static_cast<double>(1.0f - r0) * x5);
		}
		if (out_reflect_prob)
			*out_reflect_prob = reflect_prob;
		return (u < reflect_prob) ? reflected : refracted;
	}

	inline bool refracts_in_sm(uint32_t type) // sm_ray_tracer.cpp:229-233
	{
		return type == RT_HIP_MATERIAL_DIELECTRIC || type == RT_HIP_MATERIAL_AIR || type == RT_HIP_MATERIAL_VACUUM || type == RT_HIP_MATERIAL_WATER
			|| type == RT_HIP_MATERIAL_ICE;
	}

	// scatter_funcs table, mg_ray_tracer.cpp:142-152: metal -> metal_scatter, everything else -> lambert_scatter
	// (sm_ray_tracer.cpp:221-236 additionally sends dielectric/air/vacuum/water/ice to dielectric_scatter)
	inline bool scatter(const frame& f, const ray& r, const hit_result& hit, random_stream& rng, ray& out, vec3& attenuation)
	{
		const material& m = f.materials[hit.material];
		attenuation = m.attenuation;
		if (m.type == RT_HIP_MATERIAL_METAL)
			return metal_scatter(r, hit, m.roughness, rng, out);
		if (f.sm_materials && refracts_in_sm(m.type))
			return dielectric_scatter(r, hit, m.reflectivity, rng, out);
		return lambert_scatter(r, hit, rng, out);
	}

	struct counters
	{
		uint64_t segments = 0;
	};

	// mg_ray_tracer.cpp:155-174, literal recursion
	vec3 trace_recursive(const frame& f, const ray& r, uint32_t max_bounces, random_stream& rng, counters& c)
	{
		if (!(max_bounces--))
			return { 0, 0, 0 };
		c.segments++;
		const hit_result hit = closest_hit(*f.scene, r);
		if (!hit)
			return sky(r.dir.y);
		vec3 attenuation;
		ray scattered;
		if (scatter(f, r, hit, rng, scattered, attenuation))
			return attenuation * trace_recursive(f, scattered, max_bounces, rng, c);
		return { 0, 0, 0 };
	}

	// The same function with the recursion unrolled front to back: the product of attenuations is carried
	// forward ("throughput") and multiplied into the sky colour at the end.  Mathematically identical to the
	// recursion; the products associate left-to-right instead of right-to-left.  THIS is the order of the
	// arithmetic contract (a GPU lane cannot recurse cheaply; the reference's own -ffast-math build does not
	// pin an association either).
	vec3 trace_iterative(const frame& f, ray r, uint32_t max_bounces, random_stream& rng, counters& c)
	{
		vec3 throughput = { 1.0f, 1.0f, 1.0f };
		while (true)
		{
			if (!(max_bounces--))
				return { 0, 0, 0 };
			c.segments++;
			const hit_result hit = closest_hit(*f.scene, r);
			if (!hit)
				return throughput * sky(r.dir.y);
			vec3 attenuation;
			ray scattered;
			if (!scatter(f, r, hit, rng, scattered, attenuation))
				return { 0, 0, 0 };
			throughput = throughput * attenuation;
			r = scattered;
		}
	}

	inline float clamp01(float x) { return x > 1.0f ? 1.0f : (x >= 0.0f ? x : 0.0f); } // NaN -> 0

	// rt::colour{vec3} (a = 1) -> uint32, src/colour.hpp:63-65,101-106
	inline uint32_t pack(vec3 c)
	{
		const uint32_t r = static_cast<uint32_t>(clamp01(c.x) * 255.99999f);
		const uint32_t g = static_cast<uint32_t>(clamp01(c.y) * 255.99999f);
		const uint32_t b = static_cast<uint32_t>(clamp01(c.z) * 255.99999f);
		const uint32_t a = static_cast<uint32_t>(clamp01(1.0f) * 255.99999f);
		return (r << 24u) | (g << 16u) | (b << 8u) | a;
	}

	// worker lambda, mg_ray_tracer.cpp:182-201, for the pixel at (x, y)
	inline void render_pixel(const frame& f, uint32_t x, uint32_t y, uint32_t& out_rgba, float* out_rgb, counters& c)
	{
		const rt_hip_scene& s = *f.scene;
		const uint32_t pixel_index = y * f.width + x; // image_view::position_of inverse, src/image.hpp:155-159
		vec3 colour = { 0, 0, 0 };
		vec3 chunk = { 0, 0, 0 };
		for (uint32_t i = 0, e = s.samples_per_pixel; i < e; i++)
		{
			random_stream rng{ f.keys, pixel_index, i };
			float ka = 0x1.0p23f, kb = 0x1.0p23f; // sample 0 goes through the pixel centre and draws nothing (:189)
			if (i)
			{
				const random_stream::step k = rng.next_step(); // random<vec2>(): one step
				ka = static_cast<float>(k.a);
				kb = static_cast<float>(k.b);
			}
			const ray r = primary_ray(f, x, y, ka, kb);
			const vec3 sample = f.trace_order == ORACLE_TRACE_RECURSIVE ? trace_recursive(f, r, s.max_bounces, rng, c)
																		 : trace_iterative(f, r, s.max_bounces, rng, c);
			chunk = chunk + sample;
			if ((i + 1) % sample_chunk == 0 || i + 1 == e) // end of a chunk of 16 samples
			{
				colour = (i < sample_chunk) ? chunk : colour + chunk;
				chunk = { 0, 0, 0 };
			}
		}
		const float n = static_cast<float>(s.samples_per_pixel);
		colour = { colour.x / n, colour.y / n, colour.z / n }; // :195
		if (out_rgb)
		{
			out_rgb[0] = colour.x;
			out_rgb[1] = colour.y;
			out_rgb[2] = colour.z;
		}
		colour = { std::sqrt(colour.x), std::sqrt(colour.y), std::sqrt(colour.z) }; // :196-198
		out_rgba = pack(colour);													 // :200
	}

	// ---- the preview: worker lambda of src/renderers/rasterizer.cpp:28-82 for the pixel at (x, y) ------------------
	// One ray through the pixel centre; a candidate replaces the current hit only if strictly nearer (:48), in the
	// order planes, boxes, spheres (:62-64); no minimum distance.  A box hit leaves the normal as it was (:56-59).
	inline void preview_pixel(const frame& f, uint32_t x, uint32_t y, uint32_t& out_rgba, float* out_rgb)
	{
		const rt_hip_scene& s = *f.scene;
		const float px = static_cast<float>(x) + 0.5f, py = static_cast<float>(y) + 0.5f;
		const vec3 near_pos = screen_to_world(f, px, py, 0.0f); // :30
		const vec3 far_pos = screen_to_world(f, px, py, 1.0f);	// :31
		const vec3 delta = far_pos - near_pos;
		float dist = std::sqrt(dot(delta, delta)) + 1.0f; // max_dist + 1 (:33,35)
		const ray r = { near_pos, normalize(delta) };	  // :39
		bool hit = false;
		uint32_t material = 0;
		vec3 hit_pos = { 0, 0, 0 };
		vec3 normal = { 0, 1, 0 }; // vec3::constants::up (:38)
		for (uint32_t i = 0; i < s.n_planes; i++)
		{
			const vec3 n = { s.plane_normal_x[i], s.plane_normal_y[i], s.plane_normal_z[i] };
			float t;
			if (hits_plane(r, n, s.plane_d[i], t) && t < dist)
				dist = t, hit = true, material = s.plane_material[i], hit_pos = r.at(t), normal = n;
		}
		for (uint32_t i = 0; i < s.n_boxes; i++)
		{
			const vec3 c = { s.box_center_x[i], s.box_center_y[i], s.box_center_z[i] };
			const vec3 e = { s.box_extents_x[i], s.box_extents_y[i], s.box_extents_z[i] };
			float t;
			if (hits_box(r, c - e, c + e, t) && t < dist)
				dist = t, hit = true, material = s.box_material[i], hit_pos = r.at(t);
		}
		for (uint32_t i = 0; i < s.n_spheres; i++)
		{
			const vec3 c = { s.sphere_center_x[i], s.sphere_center_y[i], s.sphere_center_z[i] };
			float t;
			if (hits_sphere(r, c, s.sphere_radius[i], t) && t < dist)
			{
				dist = t, hit = true, material = s.sphere_material[i], hit_pos = r.at(t);
				normal = direction(c, hit_pos); // :54
			}
		}
		vec3 colour;
		if (hit)
		{
			// min(0.25 + lambert(N, direction(hit, near), albedo) * 0.75, 1) (:66-73); lambert = (L . N) * albedo * 1 (:13-20)
			const float* albedo = s.material_albedo + material * 4;
			const float k = dot(direction(hit_pos, near_pos), normal);
			const vec3 v = { (k * albedo[0]) * 0.75f + 0.25f, (k * albedo[1]) * 0.75f + 0.25f, (k * albedo[2]) * 0.75f + 0.25f };
			colour = { select_min(v.x, 1.0f), select_min(v.y, 1.0f), select_min(v.z, 1.0f) };
		}
		else
		{
			// lerp(sky_start, sky_end, y / (H - 1)) (:76-80) with sky_start = colour{208, 228, 255} and sky_end =
			// colour{238, 245, 255} (:66-67): the integer constructor clamps each channel to [0, 1] (colour.hpp:72-91), so
			// both are white.  A one-row frame divides 0 by 0: NaN, which packs to black.
			const float t = static_cast<float>(y) / static_cast<float>(f.height - 1u);
			const float c = std::fmaf(1.0f - 1.0f, t, 1.0f);
			colour = { c, c, c };
		}
		if (out_rgb)
			out_rgb[0] = colour.x, out_rgb[1] = colour.y, out_rgb[2] = colour.z;
		out_rgba = pack(colour);
	}
}

extern "C" int oracle_render(const rt_hip_scene* scene,
							 uint32_t width,
							 uint32_t height,
							 uint64_t seed,
							 int mode,
							 const rt_hip_partition* part,
							 uint32_t* rgba8,
							 float* rgb_f32,
							 int n_threads,
							 oracle_stats* stats)
{
	if (!scene || !rgba8 || !width || !height || !scene->samples_per_pixel || !scene->max_bounces)
		return 1;
	const rt_hip_partition whole = { 0, 1, RT_HIP_DEFAULT_STRIPE_ROWS };
	const rt_hip_partition p = part ? *part : whole;
	if (!p.world || p.rank >= p.world || !p.stripe_rows)
		return 1;

	const frame f = make_frame(scene, width, height, seed, mode);

	// rows owned by this rank, in local order
	std::vector<uint32_t> rows;
	for (uint32_t y = 0; y < height; y++)
		if ((y / p.stripe_rows) % p.world == p.rank)
			rows.push_back(y);

	unsigned threads = n_threads > 0 ? static_cast<unsigned>(n_threads) : std::thread::hardware_concurrency();
	if (!threads)
		threads = 1;
	if (threads > rows.size())
		threads = rows.empty() ? 1u : static_cast<unsigned>(rows.size());

	std::atomic<size_t> next_row{ 0 };
	std::atomic<uint64_t> segments{ 0 };
	const auto t0 = std::chrono::steady_clock::now();
	const auto body = [&]()
	{
		counters c;
		for (size_t local_y = next_row++; local_y < rows.size(); local_y = next_row++)
		{
			const uint32_t y = rows[local_y];
			for (uint32_t x = 0; x < width; x++)
			{
				const size_t o = local_y * width + x;
				if (mode & ORACLE_PREVIEW)
				{
					preview_pixel(f, x, y, rgba8[o], rgb_f32 ? rgb_f32 + o * 3 : nullptr);
					c.segments++;
				}
				else
					render_pixel(f, x, y, rgba8[o], rgb_f32 ? rgb_f32 + o * 3 : nullptr, c);
			}
		}
		segments += c.segments;
	};
	std::vector<std::thread> pool;
	for (unsigned t = 1; t < threads; t++)
		pool.emplace_back(body);
	body();
	for (auto& t : pool)
		t.join();
	const auto t1 = std::chrono::steady_clock::now();

	if (stats)
	{
		stats->primary_samples = static_cast<uint64_t>(rows.size()) * width * ((mode & ORACLE_PREVIEW) ? 1u : scene->samples_per_pixel);
		stats->segments = segments.load();
		stats->sphere_tests = stats->segments * scene->n_spheres;
		stats->plane_tests = stats->segments * scene->n_planes;
		stats->seconds = std::chrono::duration<double>(t1 - t0).count();
	}
	return 0;
}

extern "C" void oracle_random(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, float* out)
{
	// the stream of (pixel, sample) flattened: u_a, u_b, u_c of its first step, then of its second, ...
	random_stream rng{ make_frame_keys(seed), pixel, sample };
	for (uint32_t i = 0; i < n; i += 3)
	{
		const random_stream::step k = rng.next_step();
		const uint32_t word[3] = { k.a, k.b, k.c };
		for (uint32_t j = 0; j < 3 && i + j < n; j++)
			out[i + j] = static_cast<float>(word[j]) * 0x1.0p-24f;
	}
}

extern "C" void oracle_stream_keys(uint64_t seed, uint32_t n, const uint32_t* pixels, const uint32_t* samples, uint32_t* out_function_key, uint32_t* out_stride, uint32_t* out_counter)
{
	const frame_keys keys = make_frame_keys(seed);
	for (uint32_t i = 0; i < n; i++)
	{
		const random_stream rng{ keys, pixels[i], samples[i] };
		out_function_key[i] = rng.function_key;
		out_stride[i] = rng.stride;
		out_counter[i] = rng.counter;
	}
}

extern "C" void oracle_closest_hit(const rt_hip_scene* scene,
								   uint32_t n,
								   const float* origins,
								   const float* directions,
								   float* out_distance,
								   uint32_t* out_kind,
								   uint32_t* out_index,
								   float* out_normal)
{
	for (uint32_t i = 0; i < n; i++)
	{
		const ray r = { { origins[i * 3], origins[i * 3 + 1], origins[i * 3 + 2] },
						{ directions[i * 3], directions[i * 3 + 1], directions[i * 3 + 2] } };
		const hit_result h = closest_hit(*scene, r);
		out_distance[i] = h.distance;
		out_kind[i] = h.kind;
		out_index[i] = h.index;
		out_normal[i * 3 + 0] = h.normal.x;
		out_normal[i * 3 + 1] = h.normal.y;
		out_normal[i * 3 + 2] = h.normal.z;
	}
}

extern "C" void oracle_sqrt_div(uint32_t n, const float* a, const float* b, float* out_sqrt, float* out_div)
{
	for (uint32_t i = 0; i < n; i++)
	{
		out_sqrt[i] = std::sqrt(a[i]);
		out_div[i] = a[i] / b[i];
	}
}

extern "C" void oracle_inv_sqrt(uint32_t n, const float* x, float* out)
{
	for (uint32_t i = 0; i < n; i++)
		out[i] = inv_sqrt(x[i]);
}

extern "C" void oracle_inv_sqrt_step(uint32_t n, const float* x, const float* estimate, float* out)
{
	for (uint32_t i = 0; i < n; i++)
		out[i] = inv_sqrt_step(x[i], estimate[i]);
}

extern "C" uint32_t oracle_pack(float r, float g, float b)
{
	return pack({ r, g, b });
}

extern "C" void oracle_sky(float dir_y, float* out_rgb)
{
	const vec3 c = sky(dir_y);
	out_rgb[0] = c.x;
	out_rgb[1] = c.y;
	out_rgb[2] = c.z;
}

extern "C" int oracle_primary_ray(const rt_hip_scene* scene, uint32_t width, uint32_t height, uint32_t x, uint32_t y, float ka, float kb, float* out_origin, float* out_dir)
{
	const frame f = make_frame(scene, width, height, 0, ORACLE_TRACE_ITERATIVE);
	const ray r = primary_ray(f, x, y, ka, kb);
	out_origin[0] = r.origin.x;
	out_origin[1] = r.origin.y;
	out_origin[2] = r.origin.z;
	out_dir[0] = r.dir.x;
	out_dir[1] = r.dir.y;
	out_dir[2] = r.dir.z;
	return f.pinhole_rays ? 1 : (f.eye_rays ? 2 : 0);
}

extern "C" void oracle_dielectric_direction(const float* dir, const float* normal, float reflectivity, float u, float* out_dir, float* out_reflect_prob)
{
	const vec3 r = dielectric_direction({ dir[0], dir[1], dir[2] }, { normal[0], normal[1], normal[2] }, reflectivity, u, out_reflect_prob);
	out_dir[0] = r.x;
	out_dir[1] = r.y;
	out_dir[2] = r.z;
}

extern "C" int oracle_hits_box(const float* origin, const float* dir, const float* center, const float* extents, float* out_t)
{
	const ray r = { { origin[0], origin[1], origin[2] }, { dir[0], dir[1], dir[2] } };
	const vec3 c = { center[0], center[1], center[2] }, e = { extents[0], extents[1], extents[2] };
	float t = 0.0f;
	const bool hit = hits_box(r, c - e, c + e, t);
	*out_t = hit ? t : -1.0f;
	return hit ? 1 : 0;
}
