// rt_amd/host/null_renderer.cpp — the do-nothing renderer (reference src/renderers/null_renderer.cpp): the
// minimal shape of a plug-in, and a second registry entry for the driver's --list / --renderer handling.
#include "renderer.hpp"

using namespace rt;

namespace
{
	struct null_renderer final : renderer_interface
	{
		void render(const rt::scene&, image_view&, muu::thread_pool&) noexcept override {}
	};

	REGISTER_RENDERER(null_renderer);
}
