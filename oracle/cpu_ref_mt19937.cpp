// oracle/cpu_ref_mt19937.cpp — reference-faithful COST model of mg_ray_tracer for the CPU baseline timing.
//
// TEST INFRASTRUCTURE, NOT PRODUCT (see cpu_ref.h).  PARITY UNPINNED (see cpu_ref.h).
//
// Same algorithm as cpu_ref.cpp, but shaped like the reference's CPU code so that its run time is a fair
// stand-in for the (unbuildable) reference binary:
//   - array-of-structs spheres/planes scanned linearly (scene.spheres.value()[i], mg_ray_tracer.cpp:45,72)
//   - recursive trace with `attenuation * trace(...)` (mg_ray_tracer.cpp:155-174)
//   - scatter functions returning std::optional<ray>, dispatched through a function-pointer table (:104-152)
//   - thread_local std::mt19937 + std::uniform_real_distribution<float>(0,1) (src/random.cpp:9-26)
//   - built with the reference's release flags: -O3 -mavx2 -mfma -ffast-math -ffp-contract=fast (meson.build:147-160)
// Work distribution: rows are handed to the worker threads dynamically (the best case for the CPU; the
// reference hands a contiguous index range to muu::thread_pool::for_range, mg_ray_tracer.cpp:203).
// Output is NOT bit-comparable with anything (fast-math, random_device seed); tests use it statistically.

#include "cpu_ref.h"

#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <optional>
#include <random>
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
	inline vec3 operator*(float s, vec3 a) { return a * s; }
	inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
	inline vec3 normalize(vec3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }

	struct ray
	{
		vec3 origin, dir;
		vec3 at(float t) const { return origin + dir * t; }
	};

	struct sphere_aos
	{
		vec3 center;
		float radius;
	};
	struct plane_aos
	{
		vec3 normal;
		float d;
	};

	struct scene_aos
	{
		std::vector<sphere_aos> spheres;
		std::vector<uint32_t> sphere_material;
		std::vector<plane_aos> planes;
		std::vector<uint32_t> plane_material;
		std::vector<uint32_t> type;
		std::vector<std::array<float, 4>> albedo;
		std::vector<float> roughness, reflectivity;
		uint32_t spp, max_bounces;
		float M[16];
		uint32_t width, height;
	};

	uint32_t g_fixed_seed = 0;

	std::mt19937& engine()
	{
		thread_local std::random_device rdev;
		thread_local std::mt19937 e{ g_fixed_seed ? g_fixed_seed + static_cast<uint32_t>(std::hash<std::thread::id>{}(std::this_thread::get_id()))
												   : rdev() };
		return e;
	}

	float random_float()
	{
		thread_local std::uniform_real_distribution<float> dist(0.0f, 1.0f);
		return dist(engine());
	}

	vec3 random_unit_vector()
	{
		while (true)
		{
			const vec3 p = { random_float(), random_float(), random_float() };
			if (p.x == 0.0f && p.y == 0.0f && p.z == 0.0f)
				continue;
			return normalize(p);
		}
	}

	std::optional<float> hits(const ray& r, const sphere_aos& s)
	{
		const vec3 e = s.center - r.origin;
		const float a = dot(e, r.dir);
		const float e2 = dot(e, e);
		const float r2 = s.radius * s.radius;
		const float disc = r2 - (e2 - a * a);
		if (disc < 0.0f)
			return {};
		const float f = std::sqrt(disc);
		const float t = e2 < r2 ? a + f : a - f;
		if (t < 0.0f)
			return {};
		return t;
	}

	std::optional<float> hits(const ray& r, const plane_aos& p)
	{
		const float den = dot(p.normal, r.dir);
		if (std::fabs(den) <= 1.0e-6f)
			return {};
		const float t = -(dot(p.normal, r.origin) + p.d) / den;
		if (t < 0.0f)
			return {};
		return t;
	}

	constexpr float min_hit_dist = 0.001f;

	struct hit_result
	{
		float distance;
		vec3 normal;
		unsigned material;
		explicit operator bool() const { return distance >= 0.0f; }
	};

	hit_result test_planes(const scene_aos& s, const ray r)
	{
		std::optional<size_t> hit_index;
		float hit_dist{};
		for (size_t i = 0; i < s.planes.size(); i++)
		{
			const auto obj = s.planes[i];
			const auto hit = hits(r, obj);
			if (!hit || *hit < min_hit_dist || (hit_index && hit_dist <= *hit))
				continue;
			hit_index = i;
			hit_dist = *hit;
		}
		if (!hit_index)
			return { -1, {}, 0 };
		return { hit_dist, s.planes[*hit_index].normal, s.plane_material[*hit_index] };
	}

	hit_result test_spheres(const scene_aos& s, const ray r)
	{
		std::optional<size_t> hit_index;
		float hit_dist{};
		for (size_t i = 0; i < s.spheres.size(); i++)
		{
			const auto obj = s.spheres[i];
			const auto hit = hits(r, obj);
			if (!hit || *hit < min_hit_dist || (hit_index && hit_dist <= *hit))
				continue;
			hit_index = i;
			hit_dist = *hit;
		}
		if (!hit_index)
			return { -1, {}, 0 };
		return { hit_dist, normalize(r.at(hit_dist) - s.spheres[*hit_index].center), s.sphere_material[*hit_index] };
	}

	hit_result select(const hit_result& a, const hit_result& b)
	{
		if (!a)
			return b;
		return !b || a.distance <= b.distance ? a : b;
	}

	using scatter_func = std::optional<ray>(const scene_aos&, const ray&, const hit_result&, vec3&);

	std::optional<ray> lambert_scatter(const scene_aos& s, const ray& r, const hit_result& hit, vec3& attenuation)
	{
		const auto& al = s.albedo[hit.material];
		const float refl = s.reflectivity[hit.material];
		attenuation = { al[0] * refl, al[1] * refl, al[2] * refl };
		vec3 scatter = hit.normal + random_unit_vector();
		if (std::fabs(scatter.x) <= 1.0e-6f && std::fabs(scatter.y) <= 1.0e-6f && std::fabs(scatter.z) <= 1.0e-6f)
			scatter = hit.normal;
		return ray{ r.at(hit.distance), normalize(scatter) };
	}

	std::optional<ray> metal_scatter(const scene_aos& s, const ray& r, const hit_result& hit, vec3& attenuation)
	{
		const auto& al = s.albedo[hit.material];
		const float refl = s.reflectivity[hit.material];
		attenuation = { al[0] * refl, al[1] * refl, al[2] * refl };
		const vec3 v = normalize(r.dir);
		vec3 scatter = (v - 2 * dot(v, hit.normal) * hit.normal) + s.roughness[hit.material] * random_unit_vector();
		if (dot(scatter, hit.normal) <= 0.0f)
			return {};
		return ray{ r.at(hit.distance), normalize(scatter) };
	}

	const std::array<scatter_func*, RT_HIP_MATERIAL_COUNT> scatter_funcs = []()
	{
		std::array<scatter_func*, RT_HIP_MATERIAL_COUNT> funcs{};
		for (auto& f : funcs)
			f = lambert_scatter;
		funcs[RT_HIP_MATERIAL_METAL] = metal_scatter;
		return funcs;
	}();

	vec3 trace(const scene_aos& s, const ray r, unsigned max_bounces, uint64_t& segments)
	{
		if (!(max_bounces--))
			return {};
		segments++;
		auto hit = test_planes(s, r);
		hit = select(test_spheres(s, r), hit);
		if (!hit)
		{
			const float t = 0.5f * (r.dir.y + 1.0f);
			return vec3{ 1, 1, 1 } + (vec3{ 0.5f, 0.7f, 1.0f } - vec3{ 1, 1, 1 }) * t;
		}
		vec3 attenuation;
		if (const auto scatter = scatter_funcs[s.type[hit.material]](s, r, hit, attenuation))
			return attenuation * trace(s, *scatter, max_bounces, segments);
		return {};
	}

	vec3 screen_to_world(const scene_aos& s, float px, float py, float depth)
	{
		const vec3 v = { 2.0f * (px / static_cast<float>(s.width)) - 1.0f, -2.0f * (py / static_cast<float>(s.height)) + 1.0f, depth };
		float row[4];
		for (int r = 0; r < 4; r++)
			row[r] = s.M[r * 4] * v.x + s.M[r * 4 + 1] * v.y + s.M[r * 4 + 2] * v.z + s.M[r * 4 + 3];
		return { row[0] / row[3], row[1] / row[3], row[2] / row[3] };
	}

	inline float clamp01(float x) { return x > 1.0f ? 1.0f : (x >= 0.0f ? x : 0.0f); }
}

extern "C" int oracle_render_mt19937(const rt_hip_scene* scene,
									 uint32_t width,
									 uint32_t height,
									 uint32_t fixed_seed,
									 uint32_t* rgba8,
									 float* rgb_f32,
									 int n_threads,
									 oracle_stats* stats)
{
	if (!scene || !rgba8 || !width || !height || !scene->samples_per_pixel || !scene->max_bounces)
		return 1;
	g_fixed_seed = fixed_seed;

	scene_aos s;
	for (uint32_t i = 0; i < scene->n_spheres; i++)
	{
		s.spheres.push_back({ { scene->sphere_center_x[i], scene->sphere_center_y[i], scene->sphere_center_z[i] }, scene->sphere_radius[i] });
		s.sphere_material.push_back(scene->sphere_material[i]);
	}
	for (uint32_t i = 0; i < scene->n_planes; i++)
	{
		s.planes.push_back({ { scene->plane_normal_x[i], scene->plane_normal_y[i], scene->plane_normal_z[i] }, scene->plane_d[i] });
		s.plane_material.push_back(scene->plane_material[i]);
	}
	for (uint32_t i = 0; i < scene->n_materials; i++)
	{
		s.type.push_back(scene->material_type[i] < RT_HIP_MATERIAL_COUNT ? scene->material_type[i] : 0u);
		s.albedo.push_back({ scene->material_albedo[i * 4], scene->material_albedo[i * 4 + 1], scene->material_albedo[i * 4 + 2], scene->material_albedo[i * 4 + 3] });
		s.roughness.push_back(scene->material_roughness[i]);
		s.reflectivity.push_back(scene->material_reflectivity[i]);
	}
	s.spp = scene->samples_per_pixel;
	s.max_bounces = scene->max_bounces;
	for (int i = 0; i < 16; i++)
		s.M[i] = scene->inverse_view_projection[i];
	s.width = width;
	s.height = height;

	unsigned threads = n_threads > 0 ? static_cast<unsigned>(n_threads) : std::thread::hardware_concurrency();
	if (!threads)
		threads = 1;

	std::atomic<uint32_t> next_row{ 0 };
	std::atomic<uint64_t> total_segments{ 0 };
	const auto t0 = std::chrono::steady_clock::now();
	const auto body = [&]()
	{
		uint64_t segments = 0;
		for (uint32_t y = next_row++; y < height; y = next_row++)
		{
			for (uint32_t x = 0; x < width; x++)
			{
				vec3 colour{};
				for (unsigned i = 0, e = s.spp; i < e; i++)
				{
					const float jx = i ? random_float() : 0.5f;
					const float jy = i ? random_float() : 0.5f;
					const float px = static_cast<float>(x) + jx, py = static_cast<float>(y) + jy;
					const vec3 near_pos = screen_to_world(s, px, py, 0.0f);
					const vec3 far_pos = screen_to_world(s, px, py, 1.0f);
					colour = colour + trace(s, ray{ near_pos, normalize(far_pos - near_pos) }, s.max_bounces, segments);
				}
				colour = colour * (1.0f / static_cast<float>(s.spp));
				const size_t o = static_cast<size_t>(y) * width + x;
				if (rgb_f32)
				{
					rgb_f32[o * 3] = colour.x;
					rgb_f32[o * 3 + 1] = colour.y;
					rgb_f32[o * 3 + 2] = colour.z;
				}
				const uint32_t r = static_cast<uint32_t>(clamp01(std::sqrt(colour.x)) * 255.99999f);
				const uint32_t g = static_cast<uint32_t>(clamp01(std::sqrt(colour.y)) * 255.99999f);
				const uint32_t b = static_cast<uint32_t>(clamp01(std::sqrt(colour.z)) * 255.99999f);
				rgba8[o] = (r << 24u) | (g << 16u) | (b << 8u) | 255u;
			}
		}
		total_segments += segments;
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
		stats->primary_samples = static_cast<uint64_t>(width) * height * s.spp;
		stats->segments = total_segments.load();
		stats->sphere_tests = stats->segments * scene->n_spheres;
		stats->plane_tests = stats->segments * scene->n_planes;
		stats->seconds = std::chrono::duration<double>(t1 - t0).count();
	}
	return 0;
}
