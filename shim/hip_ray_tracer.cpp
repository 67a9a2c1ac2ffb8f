// shim/hip_ray_tracer.cpp — the reference-side binding of the MI355X renderer module.
//
// Drop this file into marzer/rt's src/renderers/, add it to src/renderers/meson.build next to mg_ray_tracer.cpp
// (reference src/renderers/meson.build:7-12), put include/rt_hip.h on the include path and link librt_hip.so
// (see INTEGRATION.md).  It is the only code that sees rt / muu types; everything below it is plain C.
//
// It cannot be compiled in this repository: rt's headers pull in marzer/muu, which is a network-fetched meson
// wrap (reference subprojects/muu.wrap).  rt_amd/host/hip_ray_tracer.cpp is the same logic compiled and tested
// against a mirror of the rt interfaces (tests/test_headless.py).
//
// Replaces: mg_ray_tracer::render (reference src/renderers/mg_ray_tracer.cpp:178-205) behind
// renderer_interface::render (reference src/renderer.hpp:9-14).
#include "../scene.hpp"
#include "../image.hpp"
#include "../renderer.hpp"
MUU_DISABLE_WARNINGS;
#include <muu/thread_pool.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <rt_hip.h>
MUU_ENABLE_WARNINGS;

using namespace rt;

namespace
{
	// RT_HIP_DEVICES picks the GPUs: unset = GPU 0 (or RT_HIP_DEVICE=<n>); "all" = every visible GPU; "0,1,2,3" = those, the
	// first being the root that assembles the frame.  More than one GPU = one rt_hip_create_multi context: the frame is
	// split into row stripes, gathered over RCCL and delivered by the same single blocking call.
	//
	// RT_HIP_GROUP="<shm name>:<rank>:<world>" makes this process ONE RANK of a renderer whose ranks are processes, one per
	// GPU (RT_HIP_DEVICE picks this process's): rt_hip_create + rt_hip_join_frame_group.  The image the application hands to
	// render() must then be every rank's mapping of one shared buffer; render() returns on every rank
	// when the whole frame is in it.
	rt_hip_status create_context(rt_hip_ctx** ctx)
	{
		const char* list = std::getenv("RT_HIP_DEVICES");
		if (const char* group = std::getenv("RT_HIP_GROUP"); group && *group)
		{
			char name[208] = {};
			int rank = -1, world = 0;
			const char* const first_colon = std::strchr(group, ':');
			if (!first_colon || static_cast<size_t>(first_colon - group) >= sizeof(name) || std::sscanf(first_colon, ":%d:%d", &rank, &world) != 2)
			{
				std::cerr << "error: hip_ray_tracer: RT_HIP_GROUP must look like /name:rank:world\n";
				return RT_HIP_INVALID_ARGUMENT;
			}
			std::memcpy(name, group, static_cast<size_t>(first_colon - group));
			const char* device = std::getenv("RT_HIP_DEVICE");
			if (const rt_hip_status st = rt_hip_create(ctx, device ? std::atoi(device) : 0))
				return st;
			if (const rt_hip_status st = rt_hip_join_frame_group(*ctx, rank, world, name, 0))
			{
				rt_hip_destroy(*ctx); // (keeps the message)
				*ctx = nullptr;
				return st;
			}
			return RT_HIP_OK;
		}
		if (!list || !*list)
		{
			const char* device = std::getenv("RT_HIP_DEVICE");
			return rt_hip_create(ctx, device ? std::atoi(device) : 0);
		}
		int devices[64];
		int n = 0;
		if (std::strcmp(list, "all") == 0)
		{
			if (rt_hip_device_count(&n) != RT_HIP_OK)
				return RT_HIP_NO_DEVICE;
			n = n > 64 ? 64 : n;
			for (int i = 0; i < n; i++)
				devices[i] = i;
		}
		else
		{
			for (const char* p = list; *p && n < 64;)
			{
				char* end = nullptr;
				const long v = std::strtol(p, &end, 10);
				if (end == p)
					break;
				devices[n++] = static_cast<int>(v);
				p = (*end == ',') ? end + 1 : end;
			}
		}
		// RT_HIP_FRAME=direct (or locked+direct): no gather — every GPU stores its pixels straight into the frame (the module's
		// own page-locked one by default; rt's own back buffer with "locked")
		const char* frame = std::getenv("RT_HIP_FRAME");
		const uint32_t flags = (frame && std::strstr(frame, "direct")) ? static_cast<uint32_t>(RT_HIP_MULTI_DIRECT_FRAME) : static_cast<uint32_t>(RT_HIP_MULTI_NONE);
		return rt_hip_create_multi(ctx, devices, n, flags); // n == 0 fails there with a message
	}

	// The default passes NO frame flag: the module renders into a page-locked frame of its own and its host threads carry the
	// pixels into rt's image while the frame is still being traced.  rt's image stays plain memory to the module — rt
	// re-creates its images on resize, also while another renderer is the active one (src/window.cpp:198-203,213), and a
	// renderer's render() is noexcept / void with nowhere to report a fault (src/renderer.hpp:11).
	// RT_HIP_FRAME=locked opts in to RT_HIP_FLAG_PERSISTENT_FRAME (rt's image itself is page-locked and mapped; the kernels
	// store straight into it, a few tens of microseconds less per frame): only for a build of rt that calls
	// rt_hip_forget_frame wherever it re-creates its images (INTEGRATION.md §3 "The back buffer").
	uint32_t frame_flags()
	{
		static const uint32_t flags = []
		{
			const char* frame = std::getenv("RT_HIP_FRAME");
			return (frame && std::strstr(frame, "locked")) ? static_cast<uint32_t>(RT_HIP_FLAG_PERSISTENT_FRAME) : static_cast<uint32_t>(RT_HIP_FLAG_NONE);
		}();
		return flags;
	}

	// ModeFlags: 0 = mg_ray_tracer's scatter table; RT_HIP_FLAG_SM_MATERIALS = sm_ray_tracer's (dielectrics refract);
	// RT_HIP_FLAG_PREVIEW = the one-ray-per-pixel preview of src/renderers/rasterizer.cpp
	template <uint32_t ModeFlags>
	struct hip_renderer : renderer_interface
	{
		rt_hip_ctx* ctx = nullptr;
		bool failed_to_create = false;
		uint64_t frame_number = 0;

		~hip_renderer() noexcept override
		{
			rt_hip_destroy(ctx);
		}

		void render(const rt::scene& scene, image_view& pixels, muu::thread_pool& /*threads*/) noexcept override
		{
			if (!pixels || failed_to_create)
				return;

			if (!ctx && create_context(&ctx) != RT_HIP_OK)
			{
				if (rt_hip_last_error()[0])
					std::cerr << "error: hip_ray_tracer: " << rt_hip_last_error() << "\n";
				failed_to_create = true;
				return;
			}

			static_assert(sizeof(material_type) == sizeof(uint32_t));
			static_assert(sizeof(rt::colour) == 4 * sizeof(float));

			rt_hip_scene s{};
			s.n_spheres				= static_cast<uint32_t>(scene.spheres.size());
			s.sphere_center_x		= scene.spheres.center_x(); // soagen column accessors, src/soa.hpp
			s.sphere_center_y		= scene.spheres.center_y();
			s.sphere_center_z		= scene.spheres.center_z();
			s.sphere_radius			= scene.spheres.radius();
			s.sphere_material		= scene.spheres.material();
			s.n_planes				= static_cast<uint32_t>(scene.planes.size());
			s.plane_normal_x		= scene.planes.normal_x();
			s.plane_normal_y		= scene.planes.normal_y();
			s.plane_normal_z		= scene.planes.normal_z();
			s.plane_d				= scene.planes.d();
			s.plane_material		= scene.planes.material();
			s.n_materials			= static_cast<uint32_t>(scene.materials.size());
			s.material_type			= reinterpret_cast<const uint32_t*>(scene.materials.type());
			s.material_albedo		= reinterpret_cast<const float*>(scene.materials.albedo());
			s.material_roughness	= scene.materials.roughness();
			s.material_reflectivity = scene.materials.reflectivity();
			s.n_boxes				= static_cast<uint32_t>(scene.boxes.size()); // drawn by the preview only
			s.box_center_x			= scene.boxes.center_x();
			s.box_center_y			= scene.boxes.center_y();
			s.box_center_z			= scene.boxes.center_z();
			s.box_extents_x			= scene.boxes.extents_x();
			s.box_extents_y			= scene.boxes.extents_y();
			s.box_extents_z			= scene.boxes.extents_z();
			s.box_material			= scene.boxes.material();
			s.samples_per_pixel		= scene.samples_per_pixel;
			s.max_bounces			= scene.max_bounces;

			// element-wise through m(r, c): independent of muu's storage order (accessor form: src/scene.cpp:179)
			const auto view = scene.camera.viewport(pixels.size());
			for (size_t r = 0; r < 4; r++)
				for (size_t c = 0; c < 4; c++)
					s.inverse_view_projection[r * 4 + c] = view.inverse_view_projection(r, c);

			// a fresh seed per frame, like the reference's random_device-seeded engines (src/random.cpp:12-13)
			const char* fixed	= std::getenv("RT_HIP_SEED");
			const uint64_t seed = fixed ? std::strtoull(fixed, nullptr, 0) : ++frame_number;

			if (rt_hip_render(ctx, &s, pixels.data(), pixels.size().x, pixels.size().y, seed, frame_flags() | ModeFlags, nullptr, nullptr)
				!= RT_HIP_OK)
				std::cerr << "error: hip_ray_tracer: " << rt_hip_last_error() << "\n";
		}
	};

	struct hip_ray_tracer final : hip_renderer<RT_HIP_FLAG_NONE>
	{};
	struct hip_sm_ray_tracer final : hip_renderer<RT_HIP_FLAG_SM_MATERIALS>
	{};
	struct hip_rasterizer final : hip_renderer<RT_HIP_FLAG_PREVIEW> // could stand in for "rasterizer" at src/main.cpp:106
	{};

	REGISTER_RENDERER(hip_ray_tracer);
	REGISTER_RENDERER(hip_sm_ray_tracer);
	REGISTER_RENDERER(hip_rasterizer);
}
