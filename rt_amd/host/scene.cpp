// rt_amd/host/scene.cpp — scene files -> rt::scene, following reference src/scene.cpp:483-643.
//
// Same keys, defaults, clamps, aliases and failure conditions as the reference loader:
//   samples_per_pixel (30) / max_bounces (10), both clamped to [1, 1000]        src/scene.cpp:531-532
//   camera = { position (0,1,0), direction 'forward' }                          src/scene.cpp:534-538
//   materials[] = { type, name, albedo, roughness, reflectivity } with per-type default reflectivity
//                                                                               src/scene.cpp:540-566
//   planes[] = { position (0,0,0), normal (0,1,0) normalised, material }        src/scene.cpp:576-585
//   spheres[] = { position (0,1,-3), radius 0.5, material }                     src/scene.cpp:587-597
//   boxes[] = { position, extents, material } (not hit by mg; drawn by the preview)   src/scene.cpp:599-615
//   vectors: alias string | one number (broadcast) | array of <= N numbers      src/scene.cpp:118-167
//   colours: named alias (saturating, src/colour.hpp:72-98) | array of <= 4     src/scene.cpp:187-357
//   enums: integer value or enumerator name                                     src/scene.cpp:383-405
//   relative paths searched under scenes/, ../scenes/, ../../scenes/, ., ../, ../../   src/scene.cpp:479-480
#include "scene.hpp"
#include "toml_subset.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <filesystem>
#include <iostream>
#include <iterator>
#include <sstream>
#include <stdexcept>

namespace fs = std::filesystem;

namespace rt
{
	namespace
	{
		struct named_colour_row
		{
			std::string_view name;
			uint32_t rgb;
		};
		constexpr named_colour_row named_colour_table[] = {
#include "named_colours.inc"
		};

		template <typename... T>
		[[noreturn]] void error(const toml::node& node, const T&... args)
		{
			std::ostringstream msg;
			((msg << args), ...);
			msg << "\n\n(line " << node.line << ", column " << node.column << ")";
			throw std::runtime_error{ msg.str() };
		}

		[[noreturn]] void mismatch_error(const toml::node& node, std::string_view target)
		{
			std::ostringstream msg;
			msg << "No mapping from TOML ";
			if (node.is_array())
				msg << "array[" << node.elements.size() << "]";
			else
				msg << toml::to_string(node.type);
			msg << " to " << target;
			msg << "\n\n(line " << node.line << ", column " << node.column << ")";
			throw std::runtime_error{ msg.str() };
		}

		void deserialize(const toml::node& node, float& val)
		{
			double v;
			if (node.is_float())
				v = node.float_value;
			else if (node.is_integer())
				v = static_cast<double>(node.integer_value);
			else
				mismatch_error(node, "float");
			if (std::isinf(v) || std::isnan(v))
				error(node, "Infinities and NaNs are not allowed.");
			val = static_cast<float>(v);
		}

		void deserialize(const toml::node& node, unsigned& val)
		{
			int64_t v;
			if (node.is_integer())
				v = node.integer_value;
			else if (node.is_float() && std::floor(node.float_value) == node.float_value && std::fabs(node.float_value) < 9.0e15)
				v = static_cast<int64_t>(node.float_value);
			else
				mismatch_error(node, "unsigned int");
			if (v < 0 || v > 0xFFFFFFFFll)
				mismatch_error(node, "unsigned int");
			val = static_cast<unsigned>(v);
		}

		void deserialize(const toml::node& node, std::string& val)
		{
			if (!node.is_string())
				mismatch_error(node, "std::string");
			val = node.string_value;
		}

		// src/scene.cpp:118-167
		void deserialize(const toml::node& node, vec3& val)
		{
			if (node.is_string())
			{
				const std::string& s = node.string_value;
				if (s == "origin" || s == "zero")
					val = { 0, 0, 0 };
				else if (s == "one")
					val = { 1, 1, 1 };
				else if (s == "forward")
					val = forward_axis;
				else if (s == "back" || s == "backward")
					val = -forward_axis;
				else if (s == "up")
					val = up_axis;
				else if (s == "down")
					val = -up_axis;
				else if (s == "left")
					val = -right_axis;
				else if (s == "right")
					val = right_axis;
				else if (s == "x" || s == "x_axis")
					val = { 1, 0, 0 };
				else if (s == "y" || s == "y_axis")
					val = { 0, 1, 0 };
				else if (s == "z" || s == "z_axis")
					val = { 0, 0, 1 };
				else
					error(node, "unknown vector alias '", s, "'");
				return;
			}
			if (node.is_number())
			{
				float f;
				deserialize(node, f);
				val = { f, f, f };
				return;
			}
			if (!node.is_array() || node.elements.size() > 3)
				mismatch_error(node, "vector<float, 3>");
			float* comps[3] = { &val.x, &val.y, &val.z };
			for (size_t i = 0; i < node.elements.size(); i++)
				deserialize(node.elements[i], *comps[i]);
		}

		// src/scene.cpp:187-357
		void deserialize(const toml::node& node, colour& val)
		{
			if (node.is_string())
			{
				if (!named_colour(node.string_value, val))
					error(node, "unknown colour alias '", node.string_value, "'");
				return;
			}
			if (!node.is_array() || node.elements.size() > 4)
				mismatch_error(node, "rt::colour");
			val = { 0, 0, 0, 0 };
			float* comps[4] = { &val.r, &val.g, &val.b, &val.a };
			for (size_t i = 0; i < node.elements.size(); i++)
				deserialize(node.elements[i], *comps[i]);
			if (node.elements.size() < 4)
				val.a = 1.0f;
		}

		constexpr std::array<std::string_view, 8> material_type_names = { "lambert", "metal", "dielectric", "air",
																		  "vacuum",	 "water", "ice",		"diamond" };

		// src/scene.cpp:383-405
		void deserialize(const toml::node& node, material_type& val)
		{
			if (node.is_integer())
			{
				if (node.integer_value < 0 || node.integer_value >= static_cast<int64_t>(material_type_names.size()))
					error(node, "integer value ", node.integer_value, " was not a member of enum rt::material_type");
				val = static_cast<material_type>(node.integer_value);
				return;
			}
			if (node.is_string())
			{
				const auto it = std::find(material_type_names.begin(), material_type_names.end(), node.string_value);
				if (it == material_type_names.end())
					error(node, "string value '", node.string_value, "' was not a member of enum rt::material_type");
				val = static_cast<material_type>(std::distance(material_type_names.begin(), it));
				return;
			}
			mismatch_error(node, "rt::material_type");
		}

		template <typename T>
		T deserialize(const toml::node& parent, std::string_view key, T val)
		{
			if (const toml::node* node = parent.get(key))
				deserialize(*node, val);
			return val;
		}

		const toml::node* get_table(const toml::node& parent, std::string_view key)
		{
			const toml::node* node = parent.get(key);
			if (!node)
				return nullptr;
			if (!node->is_table())
				error(parent, "expected table at key '", key, "', got ", toml::to_string(node->type));
			return node;
		}

		const toml::node* get_array(const toml::node& parent, std::string_view key)
		{
			const toml::node* node = parent.get(key);
			if (!node)
				return nullptr;
			if (!node->is_array())
				error(parent, "expected array at key '", key, "', got ", toml::to_string(node->type));
			return node;
		}

		constexpr std::array<std::string_view, 6> path_search_prefixes = { "scenes/", "../scenes/", "../../scenes/", "", "../", "../../" };

		scene from_config(const toml::node& config)
		{
			scene s;
			s.samples_per_pixel = std::clamp(deserialize(config, "samples_per_pixel", 30u), 1u, 1000u);
			s.max_bounces = std::clamp(deserialize(config, "max_bounces", 10u), 1u, 1000u);

			if (const auto camera = get_table(config, "camera"))
				s.camera.pose(deserialize(*camera, "position", vec3{ 0, 1, 0 }), deserialize(*camera, "direction", forward_axis));

			const colour fuchsia = [] { colour c; named_colour("fuchsia", c); return c; }();

			if (const auto materials = get_array(config, "materials"))
			{
				for (const auto& tbl : materials->elements)
				{
					const auto type = deserialize(tbl, "type", material_type::lambert);
					float reflectiveness;
					switch (type)
					{
						case material_type::metal: reflectiveness = 0.8f; break;
						case material_type::dielectric: reflectiveness = 1.52f; break;
						case material_type::air: reflectiveness = 1.000293f; break;
						case material_type::vacuum: reflectiveness = 1.0f; break;
						case material_type::ice: reflectiveness = 1.31f; break;
						case material_type::water: reflectiveness = 1.333f; break;
						default: reflectiveness = 0.5f;
					}
					s.materials.push_back(deserialize(tbl, "name", std::string{}),
										  type,
										  deserialize(tbl, "albedo", fuchsia),
										  deserialize(tbl, "roughness", type == material_type::dielectric ? 0.0f : 0.5f),
										  deserialize(tbl, "reflectivity", reflectiveness));
				}
			}
			if (s.materials.empty())
				s.materials.push_back(std::string{}, material_type::lambert, fuchsia, 0.05f, 0.5f);

			const auto get_material = [&](const toml::node& parent) -> unsigned
			{
				const auto material = deserialize(parent, "material", unsigned{});
				if (material >= s.materials.size())
					error(parent, "material index ", material, " out-of-range");
				return material;
			};

			if (const auto planes = get_array(config, "planes"))
			{
				for (const auto& tbl : planes->elements)
				{
					const auto value = rt::plane{ deserialize(tbl, "position", vec3{ 0, 0, 0 }),
												  vec3::normalize(deserialize(tbl, "normal", vec3{ 0, 1, 0 })) };
					s.planes.push_back(value, get_material(tbl), value.normal.x, value.normal.y, value.normal.z, value.d);
				}
			}

			if (const auto spheres = get_array(config, "spheres"))
			{
				for (const auto& tbl : spheres->elements)
				{
					const auto value = rt::sphere{ deserialize(tbl, "position", vec3{ 0, 1, -3 }), deserialize(tbl, "radius", 0.5f) };
					s.spheres.push_back(value, get_material(tbl), value.center.x, value.center.y, value.center.z, value.radius);
				}
			}

			if (const auto boxes = get_array(config, "boxes"))
			{
				for (const auto& tbl : boxes->elements)
				{
					const auto value = rt::box{ deserialize(tbl, "position", vec3{ 0, 1, -3 }), deserialize(tbl, "extents", vec3{ 0.5f, 0.5f, 0.5f }) };
					s.boxes.push_back(value, get_material(tbl), value.center.x, value.center.y, value.center.z, value.extents.x, value.extents.y, value.extents.z);
				}
			}
			return s;
		}

		toml::node parse_or_rethrow(std::string_view text, std::string_view source_name)
		{
			try
			{
				return toml::parse(text, source_name);
			}
			catch (const toml::parse_error& e)
			{
				throw std::runtime_error{ e.what() };
			}
		}
	}

	bool named_colour(std::string_view name, colour& out) noexcept
	{
		const auto end = std::end(named_colour_table);
		const auto it = std::lower_bound(std::begin(named_colour_table), end, name, [](const named_colour_row& row, std::string_view n) { return row.name < n; });
		if (it == end || it->name != name)
			return false;
		// reference src/colour.hpp:72-98: the channel BYTE is cast to float and clamped to [0, 1] — not divided by 255 —
		// so every non-zero byte becomes 1.0
		const auto channel = [](uint32_t byte) { return std::clamp(static_cast<float>(byte), 0.0f, 1.0f); };
		out = { channel((it->rgb >> 16) & 0xFF), channel((it->rgb >> 8) & 0xFF), channel(it->rgb & 0xFF), channel(0xFF) };
		return true;
	}

	scene scene::parse(std::string_view toml_text, std::string_view source_name)
	{
		return from_config(parse_or_rethrow(toml_text, source_name));
	}

	scene scene::load(std::string_view path_sv)
	{
		if (path_sv.empty())
			throw std::runtime_error{ "no scene file path provided" };

		if (path_sv == "-")
		{
			std::ostringstream text;
			text << std::cin.rdbuf();
			return parse(text.str(), "stdin");
		}

		fs::path path{ path_sv };
		path.make_preferred();
		bool ok = false;
		if (path.is_relative())
		{
			for (const auto& root : path_search_prefixes)
			{
				auto p = path;
				if (!root.empty())
				{
					p = fs::path{ root } / p;
					p.make_preferred();
				}
				if (fs::is_regular_file(p))
				{
					path = std::move(p);
					ok = true;
					break;
				}
			}
		}
		else
			ok = fs::is_regular_file(path);

		if (!ok)
			throw std::runtime_error{ "scene path '" + path.string() + "' did not exist or was not a file" };

		toml::node config;
		try
		{
			config = toml::parse_file(path.string());
		}
		catch (const toml::parse_error& e)
		{
			throw std::runtime_error{ e.what() };
		}
		scene s = from_config(config);
		s.path = path.string();
		return s;
	}

	scene scene::load_first_available()
	{
		for (const auto& dir_sv : path_search_prefixes)
		{
			fs::path dir{ dir_sv.empty() ? std::string_view{ "." } : dir_sv };
			std::error_code ec;
			if (fs::status(dir, ec).type() != fs::file_type::directory)
				continue;
			std::vector<fs::path> candidates;
			for (const auto& file : fs::directory_iterator{ dir, ec })
			{
				if (!file.path().has_stem() || !file.path().has_extension() || file.path().extension() != ".toml")
					continue;
				if (fs::status(file, ec).type() != fs::file_type::regular)
					continue;
				candidates.push_back(file.path());
			}
			if (candidates.empty())
				continue;
			std::sort(candidates.begin(), candidates.end()); // directory order is unspecified; make it repeatable
			return load(candidates.front().string());
		}
		throw std::runtime_error{ "no scene files found" };
	}

	scene scene::synthetic(unsigned sphere_count)
	{
		// splitmix64, seed 20250310, u = (next() >> 40) * 2^-24
		uint64_t state = 20250310ull;
		const auto next = [&]() -> float
		{
			uint64_t z = (state += 0x9E3779B97F4A7C15ull);
			z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
			z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
			z ^= z >> 31;
			return static_cast<float>(z >> 40) * 0x1.0p-24f;
		};

		scene s;
		s.samples_per_pixel = 64;
		s.max_bounces = 10;
		s.camera.pose(vec3{ 0, 6, 3 }, vec3{ 0, -0.35f, -1 });
		s.materials.push_back("ground", material_type::lambert, colour{ 1, 1, 1, 1 }, 0.5f, 0.5f);
		for (unsigned k = 1; k <= 7; k++)
		{
			const bool lambert = (k & 1u) != 0;
			const colour albedo = { 0.25f + 0.75f * static_cast<float>(k & 1u),
									0.25f + 0.75f * static_cast<float>((k >> 1) & 1u),
									0.25f + 0.75f * static_cast<float>((k >> 2) & 1u),
									1.0f };
			s.materials.push_back("m" + std::to_string(k),
								  lambert ? material_type::lambert : material_type::metal,
								  albedo,
								  0.05f * static_cast<float>(k),
								  lambert ? 0.5f : 0.8f);
		}
		if (sphere_count)
			s.spheres.push_back(sphere{ { 0, -1000, 0 }, 1000 }, 0, 0, -1000, 0, 1000);
		for (unsigned i = 1; i < sphere_count; i++)
		{
			const float u1 = next(), u2 = next(), u3 = next();
			const float cx = -40.0f + 80.0f * u1;
			const float cz = -2.0f - 78.0f * u2;
			const float r = 0.05f + 0.20f * u3;
			const float cy = r;
			s.spheres.push_back(sphere{ { cx, cy, cz }, r }, 1u + (i % 7u), cx, cy, cz, r);
		}
		return s;
	}
}
