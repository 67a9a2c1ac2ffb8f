// rt_amd/host/scene.hpp — what a renderer is handed: the scene description of reference src/scene.hpp:8-25.
#pragma once

#include "camera.hpp"
#include "soa.hpp"

#include <string>
#include <string_view>

namespace rt
{
	struct scene
	{
		// ---- how to sample (reference src/scene.hpp:10-11; the loader clamps both to [1, 1000]) ----
		unsigned samples_per_pixel{ 30u };
		unsigned max_bounces{ 10u };

		// ---- what is in it ----
		rt::camera camera{};
		rt::spheres spheres{};
		rt::planes planes{};
		rt::materials materials{};
		rt::boxes boxes{}; // loaded like the rest; mg_ray_tracer never hits them (mg_ray_tracer.cpp:89-93), the preview draws them

		// ---- where it came from ("" for parse() / synthetic()) ----
		std::string path{};

		// Loaders; all throw std::runtime_error carrying the reference's messages.
		//   load: a scene file, relative names searched like the reference does (src/scene.cpp:479-529); "-" = stdin
		//   parse: TOML text already in memory
		//   load_first_available: the first *.toml of the search directories (src/scene.cpp:620-643)
		//   synthetic: SURVEY.md 8d "synthetic-100k" — a ground sphere plus (count - 1) small ones from splitmix64(20250310)
		static scene load(std::string_view file);
		static scene parse(std::string_view toml_text, std::string_view source_name = "<string>");
		static scene load_first_available();
		static scene synthetic(unsigned sphere_count = 100000);
	};

	// named colour lookup with the reference's saturation quirk (src/colour.hpp:72-98); false if the name is unknown
	bool named_colour(std::string_view name, colour& out) noexcept;
}
