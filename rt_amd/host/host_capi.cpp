// rt_amd/host/host_capi.cpp — see host_capi.h.
#include "host_capi.h"
#include "scene.hpp"
#include "toml_subset.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>

#include <exception>
#include <string>

struct rt_host_scene
{
	rt::scene scene;
};

namespace
{
	thread_local std::string g_error;

	template <typename F>
	rt_host_scene* guarded(F&& make) noexcept
	{
		try
		{
			g_error.clear();
			return new rt_host_scene{ make() };
		}
		catch (const std::exception& e)
		{
			g_error = e.what();
		}
		catch (...)
		{
			g_error = "unknown error";
		}
		return nullptr;
	}
}

extern "C" const char* rt_host_last_error(void)
{
	return g_error.c_str();
}

extern "C" rt_host_scene* rt_host_scene_parse(const char* toml_text)
{
	return guarded([&] { return rt::scene::parse(toml_text ? toml_text : ""); });
}

extern "C" rt_host_scene* rt_host_scene_load(const char* path)
{
	return guarded([&] { return rt::scene::load(path ? path : ""); });
}

extern "C" rt_host_scene* rt_host_scene_synthetic(unsigned sphere_count)
{
	return guarded([&] { return rt::scene::synthetic(sphere_count); });
}

extern "C" void rt_host_scene_free(rt_host_scene* scene)
{
	delete scene;
}

extern "C" void rt_host_scene_set_sampling(rt_host_scene* scene, unsigned samples_per_pixel, unsigned max_bounces)
{
	if (!scene)
		return;
	if (samples_per_pixel)
		scene->scene.samples_per_pixel = samples_per_pixel;
	if (max_bounces)
		scene->scene.max_bounces = max_bounces;
}

extern "C" void rt_host_scene_set_camera(rt_host_scene* scene, const float position[3], const float direction[3])
{
	if (!scene || !position || !direction)
		return;
	scene->scene.camera.pose(rt::vec3{ position[0], position[1], position[2] }, rt::vec3{ direction[0], direction[1], direction[2] });
}

extern "C" int rt_host_scene_describe(const rt_host_scene* scene, unsigned width, unsigned height, rt_hip_scene* out)
{
	if (!scene || !out || !width || !height)
	{
		g_error = "rt_host_scene_describe: invalid argument";
		return 1;
	}
	const rt::scene& s = scene->scene;
	static_assert(sizeof(rt::material_type) == sizeof(uint32_t));
	static_assert(sizeof(rt::colour) == 4 * sizeof(float));
	static_assert(sizeof(unsigned) == sizeof(uint32_t));

	*out = {};
	out->n_spheres = static_cast<uint32_t>(s.spheres.size());
	out->sphere_center_x = s.spheres.center_x();
	out->sphere_center_y = s.spheres.center_y();
	out->sphere_center_z = s.spheres.center_z();
	out->sphere_radius = s.spheres.radius();
	out->sphere_material = s.spheres.material();
	out->n_planes = static_cast<uint32_t>(s.planes.size());
	out->plane_normal_x = s.planes.normal_x();
	out->plane_normal_y = s.planes.normal_y();
	out->plane_normal_z = s.planes.normal_z();
	out->plane_d = s.planes.d();
	out->plane_material = s.planes.material();
	out->n_materials = static_cast<uint32_t>(s.materials.size());
	out->material_type = reinterpret_cast<const uint32_t*>(s.materials.type());
	out->material_albedo = reinterpret_cast<const float*>(s.materials.albedo());
	out->material_roughness = s.materials.roughness();
	out->material_reflectivity = s.materials.reflectivity();
	out->n_boxes = static_cast<uint32_t>(s.boxes.size());
	out->box_center_x = s.boxes.center_x();
	out->box_center_y = s.boxes.center_y();
	out->box_center_z = s.boxes.center_z();
	out->box_extents_x = s.boxes.extents_x();
	out->box_extents_y = s.boxes.extents_y();
	out->box_extents_z = s.boxes.extents_z();
	out->box_material = s.boxes.material();
	out->samples_per_pixel = s.samples_per_pixel;
	out->max_bounces = s.max_bounces;
	const rt::viewport view = s.camera.viewport({ width, height });
	for (size_t r = 0; r < 4; r++)
		for (size_t c = 0; c < 4; c++)
			out->inverse_view_projection[r * 4 + c] = view.inverse_view_projection(r, c);
	return 0;
}

extern "C" void rt_host_screen_to_world(const rt_host_scene* scene, unsigned width, unsigned height, float x, float y, float depth, float out[3])
{
	const rt::viewport view = scene->scene.camera.viewport({ width, height });
	const rt::vec3 p = view.screen_to_world(x, y, depth);
	out[0] = p.x, out[1] = p.y, out[2] = p.z;
}

extern "C" int rt_host_named_colour(const char* name, float out_rgba[4])
{
	rt::colour c;
	if (!name || !rt::named_colour(name, c))
		return 0;
	out_rgba[0] = c.r, out_rgba[1] = c.g, out_rgba[2] = c.b, out_rgba[3] = c.a;
	return 1;
}

namespace
{
	void json_string(std::string& out, const std::string& s)
	{
		out += '"';
		for (const unsigned char c : s)
		{
			switch (c)
			{
				case '"': out += "\\\""; break;
				case '\\': out += "\\\\"; break;
				case '\n': out += "\\n"; break;
				case '\r': out += "\\r"; break;
				case '\t': out += "\\t"; break;
				default:
					if (c < 0x20)
					{
						char buf[8];
						std::snprintf(buf, sizeof(buf), "\\u%04x", c);
						out += buf;
					}
					else
						out += static_cast<char>(c);
			}
		}
		out += '"';
	}

	void json_node(std::string& out, const rt::toml::node& n)
	{
		using rt::toml::node_type;
		switch (n.type)
		{
			case node_type::table:
			{
				out += '{';
				bool first = true;
				for (const auto& m : n.members)
				{
					if (!first)
						out += ',';
					first = false;
					json_string(out, m.first);
					out += ':';
					json_node(out, m.second);
				}
				out += '}';
				break;
			}
			case node_type::array:
			{
				out += '[';
				for (size_t i = 0; i < n.elements.size(); i++)
				{
					if (i)
						out += ',';
					json_node(out, n.elements[i]);
				}
				out += ']';
				break;
			}
			case node_type::string: json_string(out, n.string_value); break;
			case node_type::integer: out += std::to_string(n.integer_value); break;
			case node_type::floating_point:
			{
				if (std::isnan(n.float_value))
					out += "\"nan\"";
				else if (std::isinf(n.float_value))
					out += n.float_value < 0 ? "\"-inf\"" : "\"inf\"";
				else
				{
					char buf[40];
					std::snprintf(buf, sizeof(buf), "%.17g", n.float_value);
					out += buf;
					if (!std::strpbrk(buf, ".eE"))
						out += ".0"; // keep it a float for the reader on the other side
				}
				break;
			}
			case node_type::boolean: out += n.boolean_value ? "true" : "false"; break;
			default: out += "null";
		}
	}
}

extern "C" long rt_host_toml_to_json(const char* toml_text, char* out, unsigned long capacity)
{
	try
	{
		g_error.clear();
		const rt::toml::node root = rt::toml::parse(toml_text ? toml_text : "");
		std::string json;
		json_node(json, root);
		if (out && capacity)
		{
			const size_t n = json.size() < capacity - 1 ? json.size() : capacity - 1;
			std::memcpy(out, json.data(), n);
			out[n] = '\0';
		}
		return static_cast<long>(json.size());
	}
	catch (const std::exception& e)
	{
		g_error = e.what();
		return -1;
	}
}
