// rt_amd/csrc/multi.hip — rt_hip_render on several GPUs (rt_hip_create_multi: one process; rt_hip_join_ranks: one process per
// GPU): the scene replicated, row stripes of 8 dealt round-robin, every member's share on its own stream, ONE gather of the
// compact stripe buffers to rank 0 (RCCL over xGMI), assemble, and the frame's delivery to rank 0's caller.  Replaces the
// reference's only parallelism, threads.for_range + wait() (src/renderers/mg_ray_tracer.cpp:203-204).
#include "internal.hpp"

#include <algorithm>

using namespace rt_hip;

namespace rt_hip
{
	// rt_hip_render on a context made by rt_hip_create_multi / rt_hip_join_ranks
	rt_hip_status render_multi(rt_hip_ctx* root,
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
		const bool keep_stats = stats || (flags & RT_HIP_FLAG_STATS);
		const uint32_t render_flags = flags & render_flag_mask;
		const int n = 1 + static_cast<int>(root->peers.size()); // members in this process
		const uint32_t world = root->world;						 // ranks in all
		const bool have_root = root->first_rank == 0;			 // rank 0 assembles the frame and hands it to its caller
		if (have_root && !pixels_rgba8888)
			return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_render: rank 0 needs the frame buffer");
		const rt_hip_partition whole = { 0, world, RT_HIP_DEFAULT_STRIPE_ROWS };
		uint32_t padded_rows = 0;
		if (const rt_hip_status st = rt_hip_padded_local_rows(height, &whole, &padded_rows))
			return st;
		const size_t pixels = static_cast<size_t>(width) * height;
		const size_t stripe_pixels = static_cast<size_t>(padded_rows) * width; // what every member sends
		if (stripe_pixels * 3u > 0x7FFFFFFFull)
			return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_render: a %ux%u frame over %u ranks exceeds the gather's element count", width, height, world);

		// the caller's columns are looked at ONCE per frame, whatever the number of members: pointer check, fingerprint
		const auto scene_t0 = std::chrono::steady_clock::now();
		scene_request request;
		if (const rt_hip_status st = open_request(request, scene))
			return st;
		const float fingerprint_ms = static_cast<float>(seconds_since(scene_t0) * 1e3); // (added to every member's upload_ms below)
		root->phases = rt_hip_phases{};
		root->phases.transport = root->peer_copy ? RT_HIP_TRANSPORT_PEER_COPY : RT_HIP_TRANSPORT_RCCL_GATHER;
		const auto finish = [&](std::chrono::steady_clock::time_point issued) -> rt_hip_status
		{
			root->phases.host_issue_ms = static_cast<float>(std::chrono::duration<double>(issued - entered).count() * 1e3);
			root->phases.host_wait_ms = static_cast<float>(seconds_since(issued) * 1e3);
			if (keep_stats)
				root->phases.render_ms = elapsed_or_zero(root->render_begin, root->render_end);
			if (stats)
				return rt_hip_stats_fetch(root, stats);
			return ok();
		};

		// The frame as the root's GPU sees it (NULL: this process holds no rank 0, it has no frame): the caller's own buffer if
		// it is page-locked and mapped (RT_HIP_FLAG_PERSISTENT_FRAME), else the module's own page-locked frame, from which
		// the carrier's threads take the pixels on into the caller's buffer while the frame is still coming in (frame.hip).
		uint32_t* mapped_frame = nullptr;
		frame_delivery* const delivery = have_root ? delivery_of(root) : nullptr;
		struct abandon_unless_finished // any early exit: no thread keeps copying, the staging frame is wiped before its next use
		{
			frame_delivery* delivery;
			~abandon_unless_finished()
			{
				if (delivery)
					delivery->abandon();
			}
		} staged{ nullptr }; // (declared BEFORE `settle`: the members' streams are drained first, then the carrier stops)
		if (have_root)
		{
			RT_HIP_TRY(hipSetDevice(root->device));
			if (!delivery)
				return fail(RT_HIP_RUNTIME_ERROR, "rt_hip_render: out of host memory");
			if (root->pinned_frame == pixels_rgba8888)
			{
				void* view = nullptr;
				if (hipHostGetDevicePointer(&view, pixels_rgba8888, 0) == hipSuccess && view)
					mapped_frame = static_cast<uint32_t*>(view);
				else
					(void)hipGetLastError();
			}
			if (!mapped_frame)
			{
				// (gathered: the bulk of the frame arrives through the assemble kernel, top rows first; direct: bottom first)
				if (const rt_hip_status st = delivery->begin(pixels_rgba8888, pixels, &mapped_frame, root->direct_frame && n == static_cast<int>(world) && !rgb_f32))
					return st;
				staged.delivery = delivery;
			}
		}
		settle_members settle(root);
		// everything the devices stored into the frame is visible (all streams have drained): hand the frame over
		const auto deliver = [&]
		{
			if (staged.delivery)
			{
				staged.delivery->finish();
				staged.delivery = nullptr;
			}
		};

		// RT_HIP_MULTI_DIRECT_FRAME: no gather at all.  The frame — the caller's page-locked back buffer, or the module's own —
		// is mapped into every member's address space; each member's kernel stores its pixels straight into their image
		// rows (system-scope stores over that GPU's own PCIe link), and the call is over when the last member's launch is.
		// Needs all ranks in this process; the float mean still goes the gathered way.
		if (root->direct_frame && mapped_frame && n == static_cast<int>(world) && !rgb_f32)
		{
			bool mapped_everywhere = true;
			std::vector<uint32_t*> views(static_cast<size_t>(n), nullptr);
			views[0] = mapped_frame;
			for (int r = 1; r < n && mapped_everywhere; r++)
			{
				if (staged.delivery) // the module's own frame is portable: every device has a view of it
					views[static_cast<size_t>(r)] = staged.delivery->view_on(member_of(root, r)->device);
				else
				{
					RT_HIP_TRY(hipSetDevice(member_of(root, r)->device));
					void* view = nullptr;
					if (hipHostGetDevicePointer(&view, pixels_rgba8888, 0) == hipSuccess && view)
						views[static_cast<size_t>(r)] = static_cast<uint32_t*>(view);
					else
						(void)hipGetLastError();
				}
				mapped_everywhere = views[static_cast<size_t>(r)] != nullptr;
			}
			if (mapped_everywhere)
			{
				root->phases.transport = RT_HIP_TRANSPORT_DIRECT_FRAME;
				for (int r = 0; r < n; r++) // scenes first (normally: n fingerprint comparisons), then nothing but launches
				{
					if (const rt_hip_status st = make_resident(member_of(root, r), request))
						return st;
					member_of(root, r)->stats.upload_ms += fingerprint_ms;
				}
				settle.armed = true;
				for (int r = 0; r < n; r++)
				{
					rt_hip_ctx* member = member_of(root, r);
					const rt_hip_partition part = { static_cast<uint32_t>(r), world, RT_HIP_DEFAULT_STRIPE_ROWS };
					if (const rt_hip_status st = render_device(member, width, height, seed, render_flags, &part, views[static_cast<size_t>(r)], nullptr, member->stream, true, keep_stats, true))
						return st;
				}
				if (staged.delivery)
					staged.delivery->launched();
				const auto issued = std::chrono::steady_clock::now();
				RT_HIP_TRY(hipSetDevice(root->device));
				if (keep_stats)
					RT_HIP_TRY(hipEventSynchronize(root->render_end));
				const auto t0 = std::chrono::steady_clock::now();
				for (int r = n - 1; r >= 0; r--) // (the root last: its device is then the current one again)
				{
					rt_hip_ctx* member = member_of(root, r);
					RT_HIP_TRY(hipSetDevice(member->device));
					RT_HIP_TRY(hipStreamSynchronize(member->stream));
				}
				settle.armed = false;
				deliver();
				root->stats.readback_ms = keep_stats ? static_cast<float>(seconds_since(t0) * 1e3) : 0.0f;
				return finish(issued);
			}
		}

		// The root's OWN stripes need no exchange: with a page-locked back buffer its kernel stores them straight into
		// their image rows of the caller's frame, like a single GPU does with the whole frame.  (Not when the float mean
		// is wanted: both outputs of a launch share one layout.)
		const bool root_direct = mapped_frame && !rgb_f32;

		// 1. every member: scene resident (normally one fingerprint comparison each), stripe buffers in place
		for (int r = 0; r < n; r++)
		{
			rt_hip_ctx* member = member_of(root, r);
			if (const rt_hip_status st = make_resident(member, request))
				return st;
			member->stats.upload_ms += fingerprint_ms;
			if (!(root_direct && r == 0))
				RT_HIP_TRY(member->stripes_rgba.reserve(stripe_pixels * sizeof(uint32_t)));
			if (rgb_f32)
				RT_HIP_TRY(member->stripes_rgb.reserve(stripe_pixels * 3 * sizeof(float)));
		}
		RT_HIP_TRY(hipSetDevice(root->device));
		if (have_root)
		{
			RT_HIP_TRY(root->gathered_rgba.reserve(stripe_pixels * sizeof(uint32_t) * world));
			if (rgb_f32)
				RT_HIP_TRY(root->gathered_rgb.reserve(stripe_pixels * 3 * sizeof(float) * world));
			if (rgb_f32)
			{
				RT_HIP_TRY(root->frame_rgb.reserve(pixels * 3 * sizeof(float)));
				RT_HIP_TRY(root->staging_rgb.reserve(pixels * 3 * sizeof(float)));
			}
		}

		// 2. every member: its share of the frame launched on its own stream — nothing between two launches but the next
		//    launch, and nothing here waits for a GPU, so the members run concurrently
		settle.armed = true;
		for (int r = 0; r < n; r++)
		{
			rt_hip_ctx* member = member_of(root, r);
			const rt_hip_partition part = { root->first_rank + static_cast<uint32_t>(r), world, RT_HIP_DEFAULT_STRIPE_ROWS };
			const bool direct = root_direct && r == 0;
			uint32_t* const target = direct ? mapped_frame : member->stripes_rgba.as<uint32_t>();
			if (const rt_hip_status st = render_device(member, width, height, seed, render_flags, &part, target, rgb_f32 ? member->stripes_rgb.as<float>() : nullptr, member->stream, direct, keep_stats, direct))
				return st;
			if (root->peer_copy && r)
				RT_HIP_TRY(hipEventRecord(member->stripes_ready, member->stream));
		}
		if (staged.delivery)
			staged.delivery->launched();

		// 3. ONE gather of the compact stripe buffers to rank 0, rank order.  With root_direct the root contributes
		//    nothing: it "sends" its own slot of the receive buffer in place, which RCCL does not copy.
		RT_HIP_TRY(hipSetDevice(root->device));
		if (root->peer_copy)
		{
			for (int r = root_direct ? 1 : 0; r < n; r++)
			{
				rt_hip_ctx* member = member_of(root, r);
				if (r)
					RT_HIP_TRY(hipStreamWaitEvent(root->stream, member->stripes_ready, 0));
				RT_HIP_TRY(hipMemcpyPeerAsync(root->gathered_rgba.as<uint32_t>() + stripe_pixels * static_cast<size_t>(r), root->device, member->stripes_rgba.ptr, member->device, stripe_pixels * sizeof(uint32_t), root->stream));
				if (rgb_f32)
					RT_HIP_TRY(hipMemcpyPeerAsync(root->gathered_rgb.as<float>() + stripe_pixels * 3 * static_cast<size_t>(r), root->device, member->stripes_rgb.ptr, member->device, stripe_pixels * 3 * sizeof(float), root->stream));
			}
		}
		else
		{
			// rccl.h: ncclGather(sendbuff, recvbuff, sendcount, datatype, root, comm, stream); recvbuff is read on the
			// root only.  One communicator per member, so the calls of all members go into one group.
			RT_HIP_TRY_NCCL(ncclGroupStart());
			ncclResult_t res = ncclSuccess;
			for (int r = 0; r < n && res == ncclSuccess; r++)
			{
				rt_hip_ctx* member = member_of(root, r);
				const bool receives = have_root && r == 0;
				const void* const send = (receives && root_direct) ? root->gathered_rgba.ptr : member->stripes_rgba.ptr;
				res = ncclGather(send, receives ? root->gathered_rgba.ptr : nullptr, stripe_pixels, ncclUint32, 0, root->comms[static_cast<size_t>(r)], member->stream);
				if (res == ncclSuccess && rgb_f32)
					res = ncclGather(member->stripes_rgb.ptr, receives ? root->gathered_rgb.ptr : nullptr, stripe_pixels * 3, ncclFloat, 0, root->comms[static_cast<size_t>(r)], member->stream);
			}
			const ncclResult_t end = ncclGroupEnd();
			RT_HIP_TRY_NCCL(res);
			RT_HIP_TRY_NCCL(end);
		}

		if (!have_root)
		{
			// a rank of a renderer whose rank 0 lives in another process: done when its stripes have been sent
			const auto issued = std::chrono::steady_clock::now();
			RT_HIP_TRY(hipStreamSynchronize(root->stream));
			settle.armed = false;
			return finish(issued);
		}

		// 4. rank 0: de-interleave the other ranks' stripes into the frame — page-locked host memory either way (the caller's
		//    or the module's own): system-scope stores, the pixels cross PCIe while the kernel runs; no frame in HBM, no
		//    device-to-host copy.
		if (keep_stats)
			RT_HIP_TRY(hipEventRecord(root->gathered, root->stream));
		if (!(root_direct && world == 1u)) // (a world of one rendered everything in place)
			launch_assemble(width, height, world, RT_HIP_DEFAULT_STRIPE_ROWS, padded_rows, root->gathered_rgba.as<uint32_t>(), mapped_frame, root_direct ? 1u : 0u, true, root->stream);
		RT_HIP_TRY(hipGetLastError());
		if (rgb_f32)
		{
			launch_assemble(width * 3u, height, world, RT_HIP_DEFAULT_STRIPE_ROWS, padded_rows, root->gathered_rgb.as<uint32_t>(), root->frame_rgb.as<uint32_t>(), 0u, false, root->stream);
			RT_HIP_TRY(hipGetLastError());
		}
		if (keep_stats)
			RT_HIP_TRY(hipEventRecord(root->assembled, root->stream));
		if (rgb_f32) // the float mean lands in the module's own page-locked buffer and is copied on from there
			RT_HIP_TRY(hipMemcpyAsync(root->staging_rgb.ptr, root->frame_rgb.ptr, pixels * 3 * sizeof(float), hipMemcpyDeviceToHost, root->stream));
		if (keep_stats)
			RT_HIP_TRY(hipEventRecord(root->copied, root->stream));
		const auto issued = std::chrono::steady_clock::now();
		if (keep_stats)
			RT_HIP_TRY(hipEventSynchronize(root->render_end)); // (the root's own kernel: where the read-back clock starts)
		const auto t0 = std::chrono::steady_clock::now();
		RT_HIP_TRY(hipStreamSynchronize(root->stream));
		deliver(); // (the root's kernel and the assemble kernel are the only writers of the frame, both on this stream)
		if (rgb_f32)
			delivery->carrier.copy(rgb_f32, root->staging_rgb.ptr, pixels * 3 * sizeof(float));
		root->stats.readback_ms = keep_stats ? static_cast<float>(seconds_since(t0) * 1e3) : 0.0f;
		// the other members' streams end with their send, which the root's receive has already waited for; settle them
		// anyway, so that a caller who changes the scene next finds every device idle
		for (rt_hip_ctx* member : root->peers)
		{
			RT_HIP_TRY(hipSetDevice(member->device));
			RT_HIP_TRY(hipStreamSynchronize(member->stream));
		}
		RT_HIP_TRY(hipSetDevice(root->device));
		settle.armed = false;
		if (keep_stats)
		{
			root->phases.gather_ms = elapsed_or_zero(root->render_end, root->gathered);
			root->phases.assemble_ms = elapsed_or_zero(root->gathered, root->assembled);
			root->phases.copy_ms = elapsed_or_zero(root->assembled, root->copied);
		}
		return finish(issued);
	}
}
