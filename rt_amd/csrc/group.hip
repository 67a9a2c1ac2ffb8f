// rt_amd/csrc/group.hip — rt_hip_render as ONE RANK of a frame group (rt_hip_join_frame_group): rank processes, one per GPU,
// that all map the caller's back buffer and store their stripes straight into it; no data-path collective (frame_group.hpp
// has the protocol).  This form page-locks the caller's (shared) buffer by design: the caller set that buffer up for it.
#include "internal.hpp"

using namespace rt_hip;

namespace rt_hip
{
	// rt_hip_render on a context that joined a frame group: this rank's stripes, straight into the back buffer all ranks map
	rt_hip_status render_group(rt_hip_ctx* ctx,
							   const rt_hip_scene* scene,
							   uint32_t* pixels_rgba8888,
							   uint32_t width,
							   uint32_t height,
							   uint64_t seed,
							   uint32_t flags,
							   float* rgb_f32,
							   rt_hip_stats* stats,
							   std::chrono::steady_clock::time_point entered)
	{
		frame_group& group = *ctx->group;
		const uint32_t rank = group.rank, world = group.world;
		const bool keep_stats = stats || (flags & RT_HIP_FLAG_STATS);
		// Whatever goes wrong on this rank alone breaks the group: the other ranks are (or will be) waiting for this one.
		const auto give_up = [&](rt_hip_status status) -> rt_hip_status
		{
			group.break_group("rank %u: %s", rank, last_error().c_str());
			return status;
		};
		const auto group_failed = [&](frame_group::outcome o) -> rt_hip_status
		{
			return fail(o == frame_group::outcome::timed_out ? RT_HIP_TIMEOUT : RT_HIP_RUNTIME_ERROR, "rt_hip_render: %s", group.error.c_str());
		};
		if (group.is_broken())
			return fail(RT_HIP_RUNTIME_ERROR, "rt_hip_render: the frame group is broken: %s", group.why_broken().c_str());
		if (!pixels_rgba8888)
			return give_up(fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_render: every rank of a frame group passes its mapping of the shared pixel buffer"));
		if (rgb_f32)
			return give_up(fail(RT_HIP_UNSUPPORTED, "rt_hip_render: the float mean does not travel through a shared frame (use rt_hip_join_ranks)"));

		// this rank's own preparations: the scene (normally one fingerprint pass and one comparison), the page-lock
		const auto scene_t0 = std::chrono::steady_clock::now();
		scene_request request;
		if (const rt_hip_status st = open_request(request, scene))
			return give_up(st);
		ctx->phases = rt_hip_phases{};
		ctx->phases.transport = RT_HIP_TRANSPORT_SHARED_FRAME;
		if (const rt_hip_status st = make_resident(ctx, request))
			return give_up(st);
		ctx->stats.upload_ms = static_cast<float>(seconds_since(scene_t0) * 1e3);
		const size_t frame_bytes = static_cast<size_t>(width) * height * sizeof(uint32_t);
		// a buffer this rank has not seen before: the old page-lock goes now, the new one comes when the group has looked at it
		const bool new_buffer = ctx->pinned_frame != pixels_rgba8888 || ctx->pinned_bytes != frame_bytes;
		if (new_buffer)
			track_frame_buffer(ctx, pixels_rgba8888, frame_bytes, false);
		frame_group_rank& mine = group.block->ranks[rank];
		mine.new_buffer = new_buffer ? 1u : 0u;

		// 1. everybody is in the call, with the same arguments (rank 0's are the reference)
		// (the columns' fingerprint does not cover what changes per frame: camera and bounce limit are folded in here)
		uint64_t print = request.print ^ (0x9E3779B97F4A7C15ull * (scene->max_bounces + 1ull));
		for (const float m : scene->inverse_view_projection)
		{
			uint32_t bits;
			std::memcpy(&bits, &m, sizeof(bits));
			print = (print ^ bits) * 0x100000001B3ull;
		}
		const frame_group_call call = { width, height, flags & render_flag_mask, scene->samples_per_pixel, seed, print };
		if (rank == 0)
			group.block->call = call;
		if (const frame_group::outcome o = group.enter_frame(); o != frame_group::outcome::ok)
			return group_failed(o);
		if (rank != 0 && !same_call(group.block->call, call))
		{
			const frame_group_call& theirs = group.block->call;
			return give_up(fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_render: rank %u was called with %ux%u, %u spp, seed %llu, flags 0x%x, scene %016llx; rank 0 with %ux%u, %u spp, seed %llu, flags 0x%x, scene %016llx",
								rank, width, height, call.samples_per_pixel, static_cast<unsigned long long>(seed), call.flags, static_cast<unsigned long long>(call.scene_fingerprint),
								theirs.width, theirs.height, theirs.samples_per_pixel, static_cast<unsigned long long>(theirs.seed), theirs.flags, static_cast<unsigned long long>(theirs.scene_fingerprint)));
		}
		// 2. a buffer somebody sees for the first time.  While nobody has page-locked it yet (and only rank 0's process has
		//    touched it: its caller clears the frame, src/main.cpp:318) rank 0 moves every stripe's pages to the host NUMA node
		//    of the GPU that will store into them; then the group checks that the ranks' mappings are one memory.
		if (group.any_new_buffer())
		{
			if (rank == 0 && new_buffer)
			{
				std::vector<int> nodes(world, -1);
				for (uint32_t r = 0; r < world; r++)
					nodes[r] = group.block->ranks[r].numa_node;
				place_stripes(pixels_rgba8888, frame_bytes, width, height, RT_HIP_DEFAULT_STRIPE_ROWS, nodes);
			}
			if (const frame_group::outcome o = group.check_buffer(pixels_rgba8888); o != frame_group::outcome::ok)
				return group_failed(o);
		}
		if (new_buffer)
			track_frame_buffer(ctx, pixels_rgba8888, frame_bytes, true, false); // (the pages are where they should be: lock them there)
		uint32_t* mapped_frame = nullptr;
		if (ctx->pinned_frame == pixels_rgba8888)
		{
			void* view = nullptr;
			if (hipHostGetDevicePointer(&view, pixels_rgba8888, 0) == hipSuccess && view)
				mapped_frame = static_cast<uint32_t*>(view);
			else
				(void)hipGetLastError();
		}
		if (!mapped_frame)
			return give_up(fail(RT_HIP_RUNTIME_ERROR, "rt_hip_render: rank %u could not page-lock and map the shared pixel buffer %p (%zu bytes)", rank, static_cast<void*>(pixels_rgba8888), frame_bytes));

		// 3. this rank's stripes, stored straight into their image rows (system-scope stores over this GPU's own PCIe link)
		const rt_hip_partition part = { rank, world, RT_HIP_DEFAULT_STRIPE_ROWS };
		const rt_hip_status launched = render_device(ctx, width, height, seed, flags & render_flag_mask, &part, mapped_frame, nullptr, ctx->stream, true, keep_stats, true);
		const auto issued = std::chrono::steady_clock::now();
		// (from here on the device may be storing into the shared buffer: no return before the stream has drained)
		const hipError_t drained = hipStreamSynchronize(ctx->stream);
		if (launched != RT_HIP_OK)
			return give_up(launched);
		if (drained != hipSuccess)
			return give_up(fail(RT_HIP_RUNTIME_ERROR, "hipStreamSynchronize failed: %s", hipGetErrorString(drained)));
		if (keep_stats)
		{
			if (const rt_hip_status st = fetch_member_stats(ctx))
				return give_up(st);
		}
		else
		{
			ctx->stats.segments = ctx->stats.sphere_tests = ctx->stats.plane_tests = 0;
			ctx->stats.render_ms = 0.0f;
		}
		mine.primary_samples = ctx->stats.primary_samples;
		mine.segments = ctx->stats.segments;
		mine.sphere_tests = ctx->stats.sphere_tests;
		mine.plane_tests = ctx->stats.plane_tests;
		mine.render_ms = ctx->stats.render_ms;
		mine.upload_ms = ctx->stats.upload_ms;
		mine.kernel_variant = ctx->stats.kernel_variant;

		// 4. the frame is complete when every rank's stripes are in place
		const auto own_done = std::chrono::steady_clock::now();
		if (const frame_group::outcome o = group.finish_frame(); o != frame_group::outcome::ok)
			return group_failed(o);
		ctx->stats.readback_ms = 0.0f;
		ctx->phases.render_ms = ctx->stats.render_ms;
		ctx->phases.gather_ms = static_cast<float>(seconds_since(own_done) * 1e3); // waiting for the slowest rank (host clock)
		ctx->phases.host_issue_ms = static_cast<float>(std::chrono::duration<double>(issued - entered).count() * 1e3);
		ctx->phases.host_wait_ms = static_cast<float>(seconds_since(issued) * 1e3);
		if (stats)
		{
			*stats = ctx->stats;
			sum_group_stats(ctx, stats);
		}
		return ok();
	}
}
