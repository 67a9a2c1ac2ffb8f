// rt_amd/csrc/scene.hip — the caller's scene columns: pointer and index checks, the fingerprint that decides whether the
// columns resident in HBM are still the caller's (rt has no scene version counter, src/main.cpp:233-311), the image of the
// single HBM block with its derived per-primitive tables, and residency per context.
//
// The columns are exactly the soagen columns the reference fills at load (src/scene.cpp:583,595; src/soa.toml:6-45) and
// whose split float columns none of its renderers reads.
#include "internal.hpp"

#include <algorithm>
#include <cmath>
#include <exception>

using namespace rt_hip;

namespace
{
	// the part of the scene check that is cheap enough for every call: counts against NULL pointers
	rt_hip_status check_scene_pointers(const rt_hip_scene& s)
	{
		if (!s.samples_per_pixel || !s.max_bounces)
			return fail(RT_HIP_INVALID_ARGUMENT, "scene: samples_per_pixel and max_bounces must be >= 1");
		if (s.n_spheres && (!s.sphere_center_x || !s.sphere_center_y || !s.sphere_center_z || !s.sphere_radius || !s.sphere_material))
			return fail(RT_HIP_INVALID_ARGUMENT, "scene: %u spheres but a sphere column is NULL", s.n_spheres);
		if (s.n_planes && (!s.plane_normal_x || !s.plane_normal_y || !s.plane_normal_z || !s.plane_d || !s.plane_material))
			return fail(RT_HIP_INVALID_ARGUMENT, "scene: %u planes but a plane column is NULL", s.n_planes);
		if (!s.n_materials && (s.n_spheres || s.n_planes || s.n_boxes))
			return fail(RT_HIP_INVALID_ARGUMENT, "scene: primitives present but no materials");
		if (s.n_materials && (!s.material_type || !s.material_albedo || !s.material_roughness || !s.material_reflectivity))
			return fail(RT_HIP_INVALID_ARGUMENT, "scene: %u materials but a material column is NULL", s.n_materials);
		if (s.n_boxes && (!s.box_center_x || !s.box_center_y || !s.box_center_z || !s.box_extents_x || !s.box_extents_y || !s.box_extents_z || !s.box_material))
			return fail(RT_HIP_INVALID_ARGUMENT, "scene: %u boxes but a box column is NULL", s.n_boxes);
		return ok();
	}

	// the part that walks the columns: needed only when their content is new (an unchanged fingerprint means these very
	// bytes passed before).  The reference's loader rejects out-of-range material indices (src/scene.cpp:568-574); an index
	// that got past it would read out of bounds on the device, so it is refused here too.
	rt_hip_status check_scene_indices(const rt_hip_scene& s)
	{
		for (uint32_t i = 0; i < s.n_spheres; i++)
			if (s.sphere_material[i] >= s.n_materials)
				return fail(RT_HIP_INVALID_ARGUMENT, "scene: sphere %u has material index %u out-of-range", i, s.sphere_material[i]);
		for (uint32_t i = 0; i < s.n_planes; i++)
			if (s.plane_material[i] >= s.n_materials)
				return fail(RT_HIP_INVALID_ARGUMENT, "scene: plane %u has material index %u out-of-range", i, s.plane_material[i]);
		for (uint32_t i = 0; i < s.n_boxes; i++)
			if (s.box_material[i] >= s.n_materials)
				return fail(RT_HIP_INVALID_ARGUMENT, "scene: box %u has material index %u out-of-range", i, s.box_material[i]);
		return ok();
	}

	constexpr size_t align_up(size_t v, size_t a)
	{
		return (v + a - 1) / a * a;
	}

	// A fingerprint of the caller's scene columns — FNV-1a style over 8-byte words, not a cryptographic hash.  rt has no
	// scene version counter (src/main.cpp:233-311), so every render() has to decide whether the columns in HBM are
	// still the caller's.  The columns are hashed WHERE THEY LIE, before anything is staged or copied: an unchanged scene
	// costs one pass over its bytes and nothing else.  Four interleaved lanes: one multiply per word is a dependency chain
	// of 3-4 cycles, and a single lane would spend 0.3 ms on the 2 MB of a 100 000-sphere scene.
	struct fingerprinter
	{
		static constexpr uint64_t prime = 0x100000001B3ull;
		uint64_t lane[4] = { 0xCBF29CE484222325ull, 0x9E3779B97F4A7C15ull, 0xBF58476D1CE4E5B9ull, 0x94D049BB133111EBull };

		void add(const void* data, size_t bytes)
		{
			const unsigned char* p = static_cast<const unsigned char*>(data);
			const size_t total = bytes;
			for (; bytes >= 32; bytes -= 32, p += 32)
			{
				uint64_t w[4];
				std::memcpy(w, p, 32);
				for (int k = 0; k < 4; k++)
					lane[k] = (lane[k] ^ w[k]) * prime;
			}
			for (; bytes >= 8; bytes -= 8, p += 8)
			{
				uint64_t w;
				std::memcpy(&w, p, 8);
				lane[0] = (lane[0] ^ w) * prime;
			}
			for (; bytes; bytes--, p++)
				lane[1] = (lane[1] ^ *p) * prime;
			lane[2] = (lane[2] ^ total) * prime; // the column's length: moving a row from one column to the next changes the print
		}
		void add_count(uint32_t n) { lane[3] = (lane[3] ^ n) * prime; }
		uint64_t value() const
		{
			uint64_t h = lane[0];
			for (int k = 1; k < 4; k++)
				h = (h ^ (lane[k] + (h << 6) + (h >> 2))) * prime;
			return h;
		}
	};

	uint64_t fingerprint_of(const rt_hip_scene& s)
	{
		fingerprinter f;
		f.add_count(s.n_spheres), f.add_count(s.n_planes), f.add_count(s.n_materials), f.add_count(s.n_boxes);
		const size_t sphere_bytes = static_cast<size_t>(s.n_spheres) * 4, plane_bytes = static_cast<size_t>(s.n_planes) * 4;
		const size_t material_bytes = static_cast<size_t>(s.n_materials) * 4, box_bytes = static_cast<size_t>(s.n_boxes) * 4;
		for (const void* column : { static_cast<const void*>(s.sphere_center_x), static_cast<const void*>(s.sphere_center_y), static_cast<const void*>(s.sphere_center_z),
									static_cast<const void*>(s.sphere_radius), static_cast<const void*>(s.sphere_material) })
			f.add(column, sphere_bytes);
		for (const void* column : { static_cast<const void*>(s.plane_normal_x), static_cast<const void*>(s.plane_normal_y), static_cast<const void*>(s.plane_normal_z),
									static_cast<const void*>(s.plane_d), static_cast<const void*>(s.plane_material) })
			f.add(column, plane_bytes);
		f.add(s.material_type, material_bytes);
		f.add(s.material_albedo, material_bytes * 4);
		f.add(s.material_roughness, material_bytes);
		f.add(s.material_reflectivity, material_bytes);
		for (const void* column : { static_cast<const void*>(s.box_center_x), static_cast<const void*>(s.box_center_y), static_cast<const void*>(s.box_center_z),
									static_cast<const void*>(s.box_extents_x), static_cast<const void*>(s.box_extents_y), static_cast<const void*>(s.box_extents_z),
									static_cast<const void*>(s.box_material) })
			f.add(column, box_bytes);
		return f.value();
	}

	scene_layout layout_of(const rt_hip_scene& s)
	{
		constexpr size_t column_alignment = 256;
		size_t offset = 0;
		const auto place = [&](size_t bytes)
		{
			const size_t at = offset;
			offset = align_up(offset + bytes, column_alignment);
			return at;
		};
		const size_t sphere_bytes = static_cast<size_t>(s.n_spheres) * 4;
		const size_t plane_bytes = static_cast<size_t>(s.n_planes) * 4;
		const size_t n_primitives = static_cast<size_t>(s.n_spheres) + s.n_planes;
		scene_layout L{};
		L.scx = place(sphere_bytes), L.scy = place(sphere_bytes), L.scz = place(sphere_bytes), L.sr = place(sphere_bytes), L.sm = place(sphere_bytes);
		L.pnx = place(plane_bytes), L.pny = place(plane_bytes), L.pnz = place(plane_bytes), L.pd = place(plane_bytes), L.pm = place(plane_bytes);
		L.shading = place(static_cast<size_t>(s.n_materials) * sizeof(float4));
		L.type = place(static_cast<size_t>(s.n_materials) * 4);
		L.geometry = place(n_primitives * sizeof(float4));
		L.prim_shading = place(n_primitives * sizeof(float4));
		L.prim_metal = place(n_primitives * 4);
		L.prim_shading_sm = place(n_primitives * sizeof(float4));
		L.prim_scatter_sm = place(n_primitives * 4);
		L.box_bounds = place(static_cast<size_t>(s.n_boxes) * 2 * sizeof(float4));
		L.albedo = place(static_cast<size_t>(s.n_materials) * sizeof(float4));
		L.total = offset ? offset : column_alignment;
		return L;
	}

}

namespace rt_hip
{
	rt_hip_status open_request(scene_request& r, const rt_hip_scene* scene)
	{
		if (const rt_hip_status st = check_scene_pointers(*scene))
			return st;
		r.scene = scene;
		r.layout = layout_of(*scene);
		r.print = fingerprint_of(*scene);
		return ok();
	}

	// The image is built in `owner`'s page-locked staging buffer and copied to HBM from there — by every member of a
	// multi-GPU context that needs it (the buffer is portable).  Not in a std::vector: the HIP runtime page-locks pageable
	// sources of large copies on the fly and keeps such locks cached by address after the copy (frame.hip, "Why the module
	// stages"); this module hands it no pageable memory at all, its own included.
	static rt_hip_status build_image(rt_hip_ctx* owner, scene_request& r)
	{
		const rt_hip_scene& s = *r.scene;
		const scene_layout& L = r.layout;
		RT_HIP_TRY(hipSetDevice(owner->device));
		RT_HIP_TRY(owner->scene_staging.reserve(L.total));
		unsigned char* const host = owner->scene_staging.as<unsigned char>();
		std::memset(host, 0, L.total);
		const auto put = [&](size_t at, const void* src, size_t bytes)
		{
			if (bytes)
				std::memcpy(host + at, src, bytes);
		};
		const size_t sphere_bytes = static_cast<size_t>(s.n_spheres) * 4;
		const size_t plane_bytes = static_cast<size_t>(s.n_planes) * 4;
		const size_t n_primitives = static_cast<size_t>(s.n_spheres) + s.n_planes;
		put(L.scx, s.sphere_center_x, sphere_bytes);
		put(L.scy, s.sphere_center_y, sphere_bytes);
		put(L.scz, s.sphere_center_z, sphere_bytes);
		put(L.sr, s.sphere_radius, sphere_bytes);
		put(L.sm, s.sphere_material, sphere_bytes);
		put(L.pnx, s.plane_normal_x, plane_bytes);
		put(L.pny, s.plane_normal_y, plane_bytes);
		put(L.pnz, s.plane_normal_z, plane_bytes);
		put(L.pd, s.plane_d, plane_bytes);
		put(L.pm, s.plane_material, plane_bytes);
		for (uint32_t m = 0; m < s.n_materials; m++)
		{
			// attenuation = vec3{ albedo * reflectivity } (mg_ray_tracer.cpp:115,131; colour * float, colour.hpp:144-149)
			const float refl = s.material_reflectivity[m];
			const float shading[4] = { s.material_albedo[m * 4 + 0] * refl,
									   s.material_albedo[m * 4 + 1] * refl,
									   s.material_albedo[m * 4 + 2] * refl,
									   s.material_roughness[m] };
			put(L.shading + m * sizeof(float4), shading, sizeof(shading));
		}
		put(L.type, s.material_type, static_cast<size_t>(s.n_materials) * 4);
		put(L.albedo, s.material_albedo, static_cast<size_t>(s.n_materials) * sizeof(float4));
		for (uint32_t i = 0; i < s.n_boxes; i++)
		{
			// corners = center -/+ extents (muu::bounding_box), material index riding in the spare lane
			float bounds[8] = { s.box_center_x[i] - s.box_extents_x[i], s.box_center_y[i] - s.box_extents_y[i], s.box_center_z[i] - s.box_extents_z[i], 0.0f,
								s.box_center_x[i] + s.box_extents_x[i], s.box_center_y[i] + s.box_extents_y[i], s.box_center_z[i] + s.box_extents_z[i], 0.0f };
			std::memcpy(&bounds[3], &s.box_material[i], 4);
			put(L.box_bounds + i * 2 * sizeof(float4), bounds, sizeof(bounds));
		}
		// derived per-primitive tables (spheres, then planes)
		r.small = small_scene{};
		r.small_sm = small_scene{};
		for (size_t i = 0; i < n_primitives; i++)
		{
			const bool is_sphere = i < s.n_spheres;
			const size_t k = is_sphere ? i : i - s.n_spheres;
			float geometry[4];
			uint32_t material;
			if (is_sphere)
			{
				const float radius = s.sphere_radius[k];
				geometry[0] = s.sphere_center_x[k], geometry[1] = s.sphere_center_y[k], geometry[2] = s.sphere_center_z[k];
				geometry[3] = radius * radius; // radius^2, as hits_sphere squares it
				material = s.sphere_material[k];
			}
			else
			{
				geometry[0] = s.plane_normal_x[k], geometry[1] = s.plane_normal_y[k], geometry[2] = s.plane_normal_z[k];
				geometry[3] = s.plane_d[k];
				material = s.plane_material[k];
			}
			const uint32_t type = s.material_type[material];
			const uint32_t metal = type == RT_HIP_MATERIAL_METAL ? scatter_metal : scatter_lambert; // mg_ray_tracer.cpp:142-152
			// sm_ray_tracer.cpp:221-236
			const bool refracts = type == RT_HIP_MATERIAL_DIELECTRIC || type == RT_HIP_MATERIAL_AIR || type == RT_HIP_MATERIAL_VACUUM
							   || type == RT_HIP_MATERIAL_WATER || type == RT_HIP_MATERIAL_ICE;
			const uint32_t scatter_sm = refracts ? scatter_dielectric : metal;
			float shading_mg[4], shading_sm[4];
			std::memcpy(shading_mg, host + L.shading + material * sizeof(float4), sizeof(float4));
			std::memcpy(shading_sm, shading_mg, sizeof(float4));
			if (refracts)
				shading_sm[3] = s.material_reflectivity[material]; // index of refraction instead of the (unused) roughness
			put(L.geometry + i * sizeof(float4), geometry, sizeof(geometry));
			put(L.prim_shading + i * sizeof(float4), shading_mg, sizeof(float4));
			put(L.prim_metal + i * 4, &metal, 4);
			put(L.prim_shading_sm + i * sizeof(float4), shading_sm, sizeof(float4));
			put(L.prim_scatter_sm + i * 4, &scatter_sm, 4);
			if (i < scalar_max_spheres) // (what the scalar-register kernel is given: meaningful for scenes of <= 8 primitives)
			{
				std::memcpy(&r.small.geometry[i], geometry, sizeof(geometry));
				std::memcpy(&r.small.shading[i], shading_mg, sizeof(float4));
				r.small.scatter[i] = metal;
				std::memcpy(&r.small_sm.geometry[i], geometry, sizeof(geometry));
				std::memcpy(&r.small_sm.shading[i], shading_sm, sizeof(float4));
				r.small_sm.scatter[i] = scatter_sm;
			}
		}
		r.image = host;
		return ok();
	}

	// Make `ctx` hold the request's scene: nothing but the frame's scalars if the columns' fingerprint is the resident
	// one (the reference re-renders an unchanged scene every dirty frame, src/main.cpp:315-321), the full path otherwise.
	// Leaves ctx->device current.
	rt_hip_status make_resident(rt_hip_ctx* ctx, scene_request& r)
	{
		const auto t0 = std::chrono::steady_clock::now();
		const rt_hip_scene& s = *r.scene;
		const scene_layout& L = r.layout;
		RT_HIP_TRY(hipSetDevice(ctx->device));
		const bool resident = ctx->have_scene && ctx->scene_fingerprint == r.print && ctx->scene_bytes == L.total && ctx->scene_columns.bytes >= L.total;
		if (!resident)
		{
			if (!r.indices_checked)
			{
				if (const rt_hip_status st = check_scene_indices(s))
					return st;
				r.indices_checked = true;
			}
			if (!r.image)
				if (const rt_hip_status st = build_image(ctx, r))
					return st;
			RT_HIP_TRY(hipDeviceSynchronize()); // a previous frame may still be reading the old scene
			ctx->have_scene = false;
			RT_HIP_TRY(ctx->scene_columns.reserve(L.total));
			RT_HIP_TRY(hipMemcpy(ctx->scene_columns.ptr, r.image, L.total, hipMemcpyHostToDevice));
			ctx->scene_fingerprint = r.print;
			ctx->scene_bytes = L.total;
			ctx->small = r.small;
			ctx->small_sm = r.small_sm;

			unsigned char* base = ctx->scene_columns.as<unsigned char>();
			device_scene& d = ctx->scene;
			d.n_spheres = s.n_spheres;
			d.n_planes = s.n_planes;
			d.n_materials = s.n_materials;
			d.sphere_cx = reinterpret_cast<const float*>(base + L.scx);
			d.sphere_cy = reinterpret_cast<const float*>(base + L.scy);
			d.sphere_cz = reinterpret_cast<const float*>(base + L.scz);
			d.sphere_r = reinterpret_cast<const float*>(base + L.sr);
			d.sphere_material = reinterpret_cast<const uint32_t*>(base + L.sm);
			d.plane_nx = reinterpret_cast<const float*>(base + L.pnx);
			d.plane_ny = reinterpret_cast<const float*>(base + L.pny);
			d.plane_nz = reinterpret_cast<const float*>(base + L.pnz);
			d.plane_d = reinterpret_cast<const float*>(base + L.pd);
			d.plane_material = reinterpret_cast<const uint32_t*>(base + L.pm);
			d.material_shading = reinterpret_cast<const float4*>(base + L.shading);
			d.material_type = reinterpret_cast<const uint32_t*>(base + L.type);
			d.primitive_geometry = reinterpret_cast<const float4*>(base + L.geometry);
			d.primitive_shading = reinterpret_cast<const float4*>(base + L.prim_shading);
			d.primitive_scatter = reinterpret_cast<const uint32_t*>(base + L.prim_metal);
			d.primitive_shading_sm = reinterpret_cast<const float4*>(base + L.prim_shading_sm);
			d.primitive_scatter_sm = reinterpret_cast<const uint32_t*>(base + L.prim_scatter_sm);
			d.n_boxes = s.n_boxes;
			d.box_bounds = reinterpret_cast<const float4*>(base + L.box_bounds);
			d.material_albedo = reinterpret_cast<const float4*>(base + L.albedo);
			d.planes_tame = 1u;
			for (uint32_t k = 0; k < s.n_planes; k++)
				for (const float n : { s.plane_normal_x[k], s.plane_normal_y[k], s.plane_normal_z[k] })
					if (!(std::fabs(n) <= 0x1p40f)) // (a NaN fails too)
						d.planes_tame = 0u;
		}
		ctx->samples_per_pixel = s.samples_per_pixel;
		ctx->max_bounces = s.max_bounces;
		std::memcpy(ctx->inverse_view_projection, s.inverse_view_projection, sizeof(ctx->inverse_view_projection));
		ctx->have_scene = true;
		ctx->phases.scene_resident = resident ? 1u : 0u;
		ctx->stats.upload_ms = static_cast<float>(seconds_since(t0) * 1e3);
		return ok();
	}
}

extern "C" rt_hip_status rt_hip_scene_check(const rt_hip_scene* scene, uint64_t* out_fingerprint)
{
	if (!scene)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_scene_check: NULL argument");
	if (const rt_hip_status st = check_scene_pointers(*scene))
		return st;
	if (const rt_hip_status st = check_scene_indices(*scene))
		return st;
	if (out_fingerprint)
		*out_fingerprint = fingerprint_of(*scene);
	return ok();
}

extern "C" rt_hip_status rt_hip_scene_upload(rt_hip_ctx* ctx, const rt_hip_scene* scene)
{
	if (!ctx || !scene)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_scene_upload: NULL argument");
	try // nothing may propagate through the C boundary (the staging image allocates)
	{
		const auto t0 = std::chrono::steady_clock::now();
		scene_request request;
		if (const rt_hip_status st = open_request(request, scene))
			return st;
		if (const rt_hip_status st = make_resident(ctx, request))
			return st;
		ctx->stats.upload_ms = static_cast<float>(seconds_since(t0) * 1e3); // including the fingerprint pass
		return ok();
	}
	catch (const std::exception& e)
	{
		return fail(RT_HIP_RUNTIME_ERROR, "rt_hip_scene_upload: %s", e.what());
	}
	catch (...)
	{
		return fail(RT_HIP_RUNTIME_ERROR, "rt_hip_scene_upload: unknown exception");
	}
}
