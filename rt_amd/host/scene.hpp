// rt_amd/host/scene.hpp — rt::scene as the renderer receives it, mirroring reference src/scene.hpp:8-25.
#pragma once

#include "camera.hpp"
#include "soa.hpp"

#include <string>
#include <string_view>

namespace rt
{
	struct scene
	{
		unsigned samples_per_pixel = 30; // reference src/scene.hpp:10
		unsigned max_bounces = 10;		 // reference src/scene.hpp:11

		std::string path;
		rt::camera camera;
		rt::materials materials;
		rt::planes planes;
		rt::spheres spheres;
		size_t box_count = 0; // boxes are parsed and validated but mg_ray_tracer never hits them (mg_ray_tracer.cpp:89-93)

		// reference src/scene.cpp:483-618.  Throws std::runtime_error with the reference's messages.
		static scene load(std::string_view file);
		// the same, from TOML text already in memory ("-" / stdin branch of the reference, src/scene.cpp:490-493)
		static scene parse(std::string_view toml_text, std::string_view source_name = "<string>");
		// reference src/scene.cpp:620-643
		static scene load_first_available();

		// SURVEY.md §8d "synthetic-100k": ground sphere + (count - 1) small spheres from splitmix64(seed 20250310)
		static scene synthetic(unsigned sphere_count = 100000);
	};

	// named colour lookup with the reference's saturation quirk (src/colour.hpp:72-98): false if unknown
	bool named_colour(std::string_view name, colour& out) noexcept;
}
