// rt_amd/host/main.cpp — rt_headless: a windowless driver for rt renderers.
//
// The reference's only driver is an SDL window (src/main.cpp, src/window.cpp): no headless mode, and no flags for
// frame size, samples, seed or an output file.  This driver keeps the reference's CLI where it has
// one — `--list`, `--scene <path>` (default: first *.toml found), `--renderer <name>` with prefix matching, the
// same log lines (src/main.cpp:30-47,68-81,97-99,335-351) — and adds what a benchmark needs:
//   --size WxH   frame size (the reference uses the window size, 800x600 initially: src/main.cpp:153)
//   --spp N / --bounces N   override the scene file's values
//   --seed N     pin the random streams (exported to the renderer as RT_HIP_SEED)
//   --frames N   render N frames, report the last
//   --out file.ppm   write the frame (binary PPM, RGB)
// It renders through renderer_interface::render exactly as window::loop does (src/window.cpp:213-217 via
// src/main.cpp:315-321): clear to opaque black, then render(scene, pixels, threads).
#include "renderer.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>

using namespace rt;
using namespace std::string_view_literals;

namespace
{
	// the do-nothing renderer of the reference (src/renderers/null_renderer.cpp): the minimal shape of a plug-in, and a
	// second registry entry for --list / --renderer
	struct null_renderer final : renderer_interface
	{
		void render(const rt::scene&, image_view&, muu::thread_pool&) noexcept override {}
	};
	REGISTER_RENDERER(null_renderer);

	template <typename... Args>
	void log(Args&&... args)
	{
		(std::cout << ... << args) << "\n";
	}

	template <typename... Args>
	void error(Args&&... args)
	{
		std::cerr << "error: ";
		(std::cerr << ... << args) << "\n";
	}

	// reference src/main.cpp:68-81
	const renderers::description* find_renderer_by_name_fuzzy(std::string_view name) noexcept
	{
		if (name.empty())
			return {};
		if (auto desc = renderers::find_by_name(name))
			return desc;
		for (auto& r : renderers::all())
			if (r.name.starts_with(name))
				return &r;
		return {};
	}

	bool write_ppm(const std::string& path, const image& img)
	{
		std::ofstream out{ path, std::ios::binary };
		if (!out)
			return false;
		out << "P6\n" << img.size().x << " " << img.size().y << "\n255\n";
		const size_t n = static_cast<size_t>(img.size().x) * img.size().y;
		std::string rgb(n * 3, '\0');
		for (size_t i = 0; i < n; i++)
		{
			const uint32_t p = img.data()[i]; // RGBA8888, reference src/colour.hpp:105
			rgb[i * 3 + 0] = static_cast<char>(p >> 24);
			rgb[i * 3 + 1] = static_cast<char>(p >> 16);
			rgb[i * 3 + 2] = static_cast<char>(p >> 8);
		}
		out.write(rgb.data(), static_cast<std::streamsize>(rgb.size()));
		return static_cast<bool>(out);
	}
}

int main(int argc, char** argv)
{
	std::string scene_path, renderer_name, out_path;
	unsigned width = 800, height = 600, spp = 0, bounces = 0, frames = 1;
	bool list = false;
	// default renderer: the first whose name starts with "hip", else the first registered (reference: first "mg", :350)
	for (auto& r : renderers::all())
		if (renderer_name.empty() && r.name.starts_with("hip"))
			renderer_name = r.name;
	if (renderer_name.empty() && !renderers::all().empty())
		renderer_name = renderers::all().front().name;

	for (int i = 1; i < argc; i++)
	{
		const std::string_view arg = argv[i];
		const auto value = [&]() -> const char*
		{
			if (i + 1 >= argc)
			{
				error("missing value for ", arg);
				std::exit(2);
			}
			return argv[++i];
		};
		if (arg == "--list"sv)
			list = true;
		else if (arg == "--scene"sv)
			scene_path = value();
		else if (arg == "--renderer"sv)
			renderer_name = value();
		else if (arg == "--size"sv)
		{
			if (std::sscanf(value(), "%ux%u", &width, &height) != 2 || !width || !height)
			{
				error("--size expects WxH");
				return 2;
			}
		}
		else if (arg == "--spp"sv)
			spp = static_cast<unsigned>(std::strtoul(value(), nullptr, 10));
		else if (arg == "--bounces"sv)
			bounces = static_cast<unsigned>(std::strtoul(value(), nullptr, 10));
		else if (arg == "--frames"sv)
			frames = static_cast<unsigned>(std::strtoul(value(), nullptr, 10));
		else if (arg == "--seed"sv)
			::setenv("RT_HIP_SEED", value(), 1);
		else if (arg == "--out"sv)
			out_path = value();
		else if (arg == "--help"sv || arg == "-h"sv)
		{
			log("usage: rt_headless [--list] [--scene file.toml] [--renderer name] [--size WxH] [--spp N] [--bounces N] [--seed N] [--frames N] [--out file.ppm]");
			return 0;
		}
		else
		{
			error("unknown argument '", arg, "'");
			return 2;
		}
	}

	log("working directory: ", std::filesystem::current_path().string());
	log("renderers:");
	for (auto& r : renderers::all())
		log("    ", r.name);
	if (list)
		return 0;

	const auto desc = find_renderer_by_name_fuzzy(renderer_name);
	if (!desc)
	{
		error("no known renderer with name '", renderer_name, "'");
		return 1;
	}
	std::unique_ptr<renderer_interface> renderer{ desc->create() };
	log("created renderer: ", desc->name);

	rt::scene scene;
	try
	{
		scene = scene_path.empty() ? rt::scene::load_first_available() : rt::scene::load(scene_path);
	}
	catch (const std::exception& e)
	{
		error(e.what());
		return 1;
	}
	log("scene '", scene.path, "' loaded");
	if (spp)
		scene.samples_per_pixel = spp;
	if (bounces)
		scene.max_bounces = bounces;

	image frame{ vec2u{ width, height } };
	image_view pixels{ frame };
	muu::thread_pool threads;
	double seconds = 0.0;
	for (unsigned f = 0; f < (frames ? frames : 1u); f++)
	{
		pixels.clear(0x000000FFu); // reference src/main.cpp:318
		const auto t0 = std::chrono::steady_clock::now();
		renderer->render(scene, pixels, threads);
		seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	}
	const double rays = static_cast<double>(width) * height * scene.samples_per_pixel;
	std::printf("%ux%u, %u spp, max_bounces %u: %.3f ms per frame (render() wall clock incl. upload and read-back), %.1f Mrays/s\n",
				width, height, scene.samples_per_pixel, scene.max_bounces, seconds * 1e3, rays / seconds / 1e6);

	if (!out_path.empty())
	{
		if (!write_ppm(out_path, frame))
		{
			error("could not write '", out_path, "'");
			return 1;
		}
		log("wrote ", out_path);
	}
	return 0;
}
