// rt_amd/host/renderer.hpp — the renderer plug-in interface and registry, mirroring reference src/renderer.hpp.
//
// renderer_interface::render(const scene&, image_view&, thread_pool&) noexcept is THE drop-in boundary of the path
// (reference src/renderer.hpp:9-14); REGISTER_RENDERER(T) installs {key, name, create} from a static initialiser
// (:34-41) into a function-local registry with de-duplication by key (reference src/renderer.cpp:11-69; header-only here).
#pragma once

#include "image.hpp"
#include "scene.hpp"

#include <algorithm>
#include <span>
#include <string_view>
#include <vector>

namespace muu
{
	// stand-in for muu::thread_pool (reference src/main.cpp:165): the GPU renderer receives it and ignores it
	// (SURVEY.md §8b "Threading"); CPU renderers built against this mirror may use for_range.
	class thread_pool
	{
	  public:
		template <typename T, typename F>
		void for_range(T begin, T end, F&& worker)
		{
			for (T i = begin; i < end; i++)
				worker(i);
		}
		void wait() noexcept {}
	};
}

namespace rt
{
	struct renderer_interface
	{
		virtual void render(const scene&, image_view&, muu::thread_pool&) noexcept = 0;
		virtual ~renderer_interface() noexcept = default;
	};

	namespace renderers
	{
		struct description
		{
			using create_func = renderer_interface*();

			std::string_view key;
			std::string_view name;
			create_func* create;
		};

		namespace detail
		{
			// one registry per program; function-local so that static initialisers of other translation units can use it
			inline std::vector<description>& registry() noexcept
			{
				static std::vector<description> entries;
				return entries;
			}

			template <typename Match>
			inline const description* first_match(Match&& match) noexcept
			{
				const auto& entries = registry();
				const auto it = std::find_if(entries.begin(), entries.end(), match);
				return it == entries.end() ? nullptr : &*it;
			}
		}

		// a second registration under the same key replaces the first (reference src/renderer.cpp:21-37)
		inline void install(const description& desc)
		{
			auto& entries = detail::registry();
			const auto same_key = std::find_if(entries.begin(), entries.end(), [&](const description& d) { return d.key == desc.key; });
			if (same_key != entries.end())
				*same_key = desc;
			else
				entries.push_back(desc);
		}

		inline std::span<const description> all() noexcept { return detail::registry(); }

		inline const description* find_by_key(std::string_view key) noexcept
		{
			return key.empty() ? nullptr : detail::first_match([&](const description& d) { return d.key == key; });
		}

		inline const description* find_by_name(std::string_view name) noexcept
		{
			return name.empty() ? nullptr : detail::first_match([&](const description& d) { return d.name == name; });
		}
	}
}

#define RT_HOST_STRINGIFY_2(x) #x
#define RT_HOST_STRINGIFY(x)   RT_HOST_STRINGIFY_2(x)
#define RT_HOST_CONCAT_2(a, b) a##b
#define RT_HOST_CONCAT(a, b)   RT_HOST_CONCAT_2(a, b)

#define REGISTER_RENDERER(T)                                                                                           \
	static const int RT_HOST_CONCAT(register_val_impl_, __LINE__) =                                                    \
		(::rt::renderers::install({ .key = __FILE__ ":" RT_HOST_STRINGIFY(__LINE__) ":" RT_HOST_STRINGIFY(T),            \
									.name = RT_HOST_STRINGIFY(T),                                                      \
									.create = []() -> ::rt::renderer_interface* { return new T; } }),                  \
		 0)
