// rt_amd/host/renderer.cpp — renderer registry (reference src/renderer.cpp:11-69).
#include "renderer.hpp"

#include <cassert>
#include <vector>

namespace rt::renderers
{
	namespace
	{
		std::vector<description>& registry() noexcept
		{
			static std::vector<description> r; // function-local: usable from other TUs' static initialisers
			return r;
		}
	}

	void install(const description& desc)
	{
		assert(!desc.key.empty() && !desc.name.empty() && desc.create);
		for (auto& r : registry())
			if (r.key == desc.key)
			{
				r = desc;
				return;
			}
		registry().push_back(desc);
	}

	std::span<const description> all() noexcept
	{
		return { registry().data(), registry().size() };
	}

	const description* find_by_key(std::string_view val) noexcept
	{
		if (val.empty())
			return {};
		for (const auto& r : registry())
			if (r.key == val)
				return &r;
		return {};
	}

	const description* find_by_name(std::string_view val) noexcept
	{
		if (val.empty())
			return {};
		for (const auto& r : registry())
			if (r.name == val)
				return &r;
		return {};
	}
}
