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
//   --shared-frame NAME --rank R --world N   this process is rank R of N rt_headless processes that render ONE frame
//                together (one per GPU: RT_HIP_DEVICE picks this one's): the back buffer is every process's mapping of the
//                POSIX shared-memory object /NAME_frame (rank 0 creates it) and the plug-in joins the frame group
//                /NAME_group (RT_HIP_GROUP, rt_hip_join_frame_group).  Rank 0 clears the buffer and writes --out.
// It renders through renderer_interface::render exactly as window::loop does (src/window.cpp:213-217 via
// src/main.cpp:315-321): clear to opaque black, then render(scene, pixels, threads).
#include "renderer.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

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

	// every rank's mapping of the one shared back buffer: rank 0 creates and sizes the object, the others wait for it
	uint32_t* map_shared_frame(const std::string& name, unsigned rank, size_t bytes)
	{
		int fd = -1;
		for (int attempt = 0; attempt < 600 && fd < 0; attempt++) // (up to a minute for rank 0 to get there)
		{
			fd = rank == 0 ? shm_open(name.c_str(), O_CREAT | O_RDWR, 0600) : shm_open(name.c_str(), O_RDWR, 0600);
			struct stat st;
			if (fd >= 0 && rank != 0 && (fstat(fd, &st) != 0 || static_cast<size_t>(st.st_size) < bytes))
			{
				close(fd);
				fd = -1;
			}
			if (fd < 0)
			{
				if (rank == 0)
					return nullptr;
				usleep(100000);
			}
		}
		if (fd < 0 || (rank == 0 && ftruncate(fd, static_cast<off_t>(bytes)) != 0))
			return nullptr;
		void* const mapping = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		close(fd);
		return mapping == MAP_FAILED ? nullptr : static_cast<uint32_t*>(mapping);
	}

	bool write_ppm(const std::string& path, const image_view& img)
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
	std::string scene_path, renderer_name, out_path, shared_name;
	unsigned width = 800, height = 600, spp = 0, bounces = 0, frames = 1, rank = 0, world = 1;
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
		else if (arg == "--shared-frame"sv)
			shared_name = value();
		else if (arg == "--rank"sv)
			rank = static_cast<unsigned>(std::strtoul(value(), nullptr, 10));
		else if (arg == "--world"sv)
			world = static_cast<unsigned>(std::strtoul(value(), nullptr, 10));
		else if (arg == "--help"sv || arg == "-h"sv)
		{
			log("usage: rt_headless [--list] [--scene file.toml] [--renderer name] [--size WxH] [--spp N] [--bounces N] [--seed N] [--frames N] [--out file.ppm] [--shared-frame NAME --rank R --world N]");
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

	image frame;
	image_view pixels;
	if (shared_name.empty())
	{
		frame = image{ vec2u{ width, height } };
		pixels = image_view{ frame };
	}
	else
	{
		if (!world || rank >= world || shared_name.find('/') != std::string::npos)
		{
			error("--shared-frame NAME needs --rank R --world N with R < N, and a NAME without '/'");
			return 2;
		}
		const size_t bytes = static_cast<size_t>(width) * height * sizeof(uint32_t);
		uint32_t* const shared = map_shared_frame("/" + shared_name + "_frame", rank, bytes);
		if (!shared)
		{
			error("could not map the shared frame '/", shared_name, "_frame'");
			return 1;
		}
		pixels = image_view{ shared, vec2u{ width, height } };
		::setenv("RT_HIP_GROUP", ("/" + shared_name + "_group:" + std::to_string(rank) + ":" + std::to_string(world)).c_str(), 1);
		log("rank ", rank, " of ", world, ": back buffer is the shared mapping /", shared_name, "_frame");
	}
	muu::thread_pool threads;
	double seconds = 0.0;
	for (unsigned f = 0; f < (frames ? frames : 1u); f++)
	{
		if (rank == 0)
			pixels.clear(0x000000FFu); // reference src/main.cpp:318 (the other ranks of a shared frame leave it alone)
		const auto t0 = std::chrono::steady_clock::now();
		renderer->render(scene, pixels, threads);
		seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	}
	const double rays = static_cast<double>(width) * height * scene.samples_per_pixel;
	std::printf("%ux%u, %u spp, max_bounces %u: %.3f ms per frame (render() wall clock incl. upload and read-back), %.1f Mrays/s\n",
				width, height, scene.samples_per_pixel, scene.max_bounces, seconds * 1e3, rays / seconds / 1e6);

	if (!shared_name.empty() && rank == 0)
		shm_unlink(("/" + shared_name + "_frame").c_str()); // (every rank has had it mapped since before its first frame)
	if (!out_path.empty() && rank == 0)
	{
		if (!write_ppm(out_path, pixels))
		{
			error("could not write '", out_path, "'");
			return 1;
		}
		log("wrote ", out_path);
	}
	return 0;
}
