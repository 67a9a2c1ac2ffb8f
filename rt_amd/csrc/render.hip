// rt_amd/csrc/render.hip — one launch of the render kernels on a context (rt_hip_render_device), the work counters, and the
// single-GPU drop-in for renderer_interface::render (rt_hip_render; reference src/renderer.hpp:11,
// src/renderers/mg_ray_tracer.cpp:178-205).
#include "internal.hpp"

#include <algorithm>
#include <cmath>
#include <exception>

using namespace rt_hip;

namespace
{
	// Is this "device" pointer page-locked host memory (a caller may hand rt_hip_render_device the device view of its own
	// registered buffer)?  Asked once per pointer: the answer of the last one is kept.
	bool is_host_memory(rt_hip_ctx* ctx, const void* pointer)
	{
		if (pointer != ctx->asked_pointer)
		{
			hipPointerAttribute_t attributes{};
			const hipError_t e = hipPointerGetAttributes(&attributes, pointer);
			if (e != hipSuccess)
				(void)hipGetLastError();
			ctx->asked_pointer = pointer;
			ctx->asked_pointer_is_host = e == hipSuccess && attributes.type == hipMemoryTypeHost;
		}
		return ctx->asked_pointer_is_host;
	}
}

extern "C" rt_hip_status rt_hip_render_device(rt_hip_ctx* ctx,
											  uint32_t width,
											  uint32_t height,
											  uint64_t seed,
											  uint32_t flags,
											  const rt_hip_partition* part,
											  uint32_t* d_rgba8,
											  float* d_rgb_f32,
											  void* stream)
{
	return render_device(ctx, width, height, seed, flags, part, d_rgba8, d_rgb_f32, stream, false, true, ctx && d_rgba8 && is_host_memory(ctx, d_rgba8));
}

namespace rt_hip
{
rt_hip_status render_device(rt_hip_ctx* ctx, uint32_t width, uint32_t height, uint64_t seed, uint32_t flags, const rt_hip_partition* part, uint32_t* d_rgba8, float* d_rgb_f32, void* stream, bool whole_frame_buffers, bool keep_stats, bool host_frame)
{
	if (!ctx || !d_rgba8)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_render_device: NULL argument");
	if (!width || !height)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_render_device: empty frame %ux%u", width, height);
	if (static_cast<uint64_t>(width) * height > 0xFFFFFFFFull)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_render_device: %ux%u exceeds the 32-bit pixel index of image_view", width, height);
	if (flags & ~static_cast<uint32_t>(RT_HIP_FLAG_FORCE_TILED | RT_HIP_FLAG_FORCE_RESIDENT | RT_HIP_FLAG_PERSISTENT_FRAME | RT_HIP_FLAG_SM_MATERIALS | RT_HIP_FLAG_PREVIEW | RT_HIP_FLAG_FORCE_STREAMED | RT_HIP_FLAG_FAST | RT_HIP_FLAG_STATS | RT_HIP_FLAG_FORCE_HALF_CHUNKS | RT_HIP_FLAG_FORCE_WHOLE_CHUNKS))
		return fail(RT_HIP_UNSUPPORTED, "rt_hip_render_device: unknown flag bits 0x%x", flags);
	if ((flags & RT_HIP_FLAG_FORCE_HALF_CHUNKS) && (flags & RT_HIP_FLAG_FORCE_WHOLE_CHUNKS))
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_render_device: RT_HIP_FLAG_FORCE_HALF_CHUNKS and RT_HIP_FLAG_FORCE_WHOLE_CHUNKS exclude each other");
	if ((flags & RT_HIP_FLAG_FAST) && (flags & (RT_HIP_FLAG_SM_MATERIALS | RT_HIP_FLAG_PREVIEW)))
		return fail(RT_HIP_UNSUPPORTED, "rt_hip_render_device: RT_HIP_FLAG_FAST applies to mg_ray_tracer's path only (not with RT_HIP_FLAG_SM_MATERIALS / RT_HIP_FLAG_PREVIEW)");
	if (!ctx->have_scene)
		return fail(RT_HIP_NO_SCENE, "rt_hip_render_device: no scene uploaded");
	if (height > 65535u * 2u) // the launch grid's y dimension counts pixel tiles at least two rows high
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_render_device: frame height %u exceeds the supported 131070 rows", height);
	const rt_hip_partition whole = { 0, 1, RT_HIP_DEFAULT_STRIPE_ROWS };
	const rt_hip_partition p = part ? *part : whole;
	if (!valid_partition(p))
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_render_device: invalid partition {rank %u, world %u, stripe_rows %u}", p.rank, p.world, p.stripe_rows);

	RT_HIP_TRY(hipSetDevice(ctx->device));
	const hipStream_t s = static_cast<hipStream_t>(stream);
	// A context serialises its launches: they share the work counters, the tile queue's head and the timing events.  Work
	// on one stream is ordered by the stream; a caller that moves to ANOTHER stream first waits for the old one to drain
	// (a rare event: rt_hip_render always uses the context's own stream).
	if (ctx->launched && ctx->last_stream != s)
		RT_HIP_TRY(hipStreamSynchronize(ctx->last_stream));

	frame_params f{};
	f.width = width;
	f.height = height;
	f.local_rows = local_rows_of(height, p.rank, p.world, p.stripe_rows);
	f.rank = p.rank;
	f.world = p.world;
	f.stripe_rows = p.stripe_rows;
	f.frame_rows = whole_frame_buffers ? 1u : 0u;
	f.stripe_shift = 0xFFFFFFFFu;
	if ((p.stripe_rows & (p.stripe_rows - 1u)) == 0u)
		for (f.stripe_shift = 0; (1u << f.stripe_shift) != p.stripe_rows; f.stripe_shift++)
		{}
	f.samples_per_pixel = ctx->samples_per_pixel;
	// (the kernels keep "tracing" and the bounces still allowed in one word, lane_trace + count: a count of 2^30 is as good as any
	// larger one — no path of a frame that ever finishes is that long)
	f.max_bounces = std::min<uint32_t>(ctx->max_bounces, 1u << 30);
	const frame_keys keys = make_frame_keys(seed);
	f.frame_key_a = keys.a;
	f.frame_key_b = keys.b;
	f.sx = 2.0f / static_cast<float>(width);
	f.neg_sy = -(2.0f / static_cast<float>(height));
	const float* M = ctx->inverse_view_projection;
	for (int r = 0; r < 4; r++)
	{
		f.mx[r] = M[r * 4 + 0];
		f.my[r] = M[r * 4 + 1];
		f.k_near[r] = std::fmaf(M[r * 4 + 2], 0.0f, M[r * 4 + 3]);
		f.k_far[r] = std::fmaf(M[r * 4 + 2], 1.0f, M[r * 4 + 3]);
	}
	// Contract v4, primary rays (oracle/cpu_ref.cpp make_frame has the same lines).  w = fma(mx[3], ndc.x, fma(my[3], ndc.y,
	// k[3])) is exactly k[3] for every finite ndc when mx[3] and my[3] are (+-)0 and k[3] is not; then
	//   near(px, py) = (mx X + my Y + k_near) / w_near,   X = (2/W) px - 1,   Y = -(2/H) py + 1,
	// is affine in the pixel position, and so is far - near.  rt's frustum is a pinhole's on top of that: every near-to-far
	// line passes through the eye, near = eye + kappa (far - near) with one kappa for the frame.  The constants are worked
	// out in binary64, in THIS order of operations, and rounded to binary32 once.  A matrix is taken as a pinhole's when the
	// near point's motion per pixel is kappa times the near-to-far vector's to within 1e-5 (relative; 1e-10 of a pixel
	// step: far below what binary32 resolves); anything else goes through the homogeneous form.
	f.eye_form = 0u;
	{
		// the eye form (oracle/cpu_ref.cpp make_frame has the same lines): N_r(px, py) = mx_r X + my_r Y + k_near_r with
		// X = (2/W) px - 1, Y = -(2/H) py + 1, i.e. n0_r + n1_r px + n2_r py; E = Z.xyz / Z.w; s = sign(-Z.w)
		const double sx = 2.0 / static_cast<double>(width), sy = -(2.0 / static_cast<double>(height));
		const double zw = M[3 * 4 + 2];
		const double e[3] = { M[0 * 4 + 2] / zw, M[1 * 4 + 2] / zw, M[2 * 4 + 2] / zw };
		const double sign = zw < 0.0 ? 1.0 : -1.0;
		const double n1w = static_cast<double>(M[12]) * sx, n2w = static_cast<double>(M[13]) * sy;
		const double n0w = static_cast<double>(f.k_near[3]) - static_cast<double>(M[12]) + static_cast<double>(M[13]);
		bool finite = zw != 0.0 && std::isfinite(e[0]) && std::isfinite(e[1]) && std::isfinite(e[2]) && std::isfinite(n0w) && std::isfinite(n1w) && std::isfinite(n2w);
		for (int c = 0; c < 3 && finite; c++)
		{
			const double mx = M[c * 4 + 0], my = M[c * 4 + 1];
			const double n1 = mx * sx, n2 = my * sy, n0 = static_cast<double>(f.k_near[c]) - mx + my;
			f.eye_q0[c] = static_cast<float>(sign * (n0 - e[c] * n0w)), f.eye_q1[c] = static_cast<float>(sign * (n1 - e[c] * n1w)), f.eye_q2[c] = static_cast<float>(sign * (n2 - e[c] * n2w));
			f.eye_jq1[c] = f.eye_q1[c] * 0x1.0p-24f, f.eye_jq2[c] = f.eye_q2[c] * 0x1.0p-24f;
			f.eye_e[c] = static_cast<float>(e[c]);
			finite = std::isfinite(f.eye_q0[c]) && std::isfinite(f.eye_q1[c]) && std::isfinite(f.eye_q2[c]);
		}
		if (finite)
		{
			f.eye_w0 = static_cast<float>(sign * n0w), f.eye_w1 = static_cast<float>(sign * n1w), f.eye_w2 = static_cast<float>(sign * n2w);
			f.eye_jw1 = f.eye_w1 * 0x1.0p-24f, f.eye_jw2 = f.eye_w2 * 0x1.0p-24f;
			f.eye_zws = static_cast<float>(sign * zw);
			f.eye_form = 1u;
			// eye_form 2: over the whole frame s N.w and s N.w + s Z.w (= s F.w) keep ONE sign and stay far inside the band of the
			// kernels' unguarded reciprocal (2^-60 .. 2^60) — both are affine in the pixel position, so their extremes sit at the
			// frame's corners.  Then no lane ever needs the reciprocal's guard or the "near and far straddle w = 0" flip, and the
			// kernels skip both (same bits: the guarded forms are identities there).  Every camera rt can make is such a one.
			bool plain = true;
			double first_w = 0.0;
			for (int corner = 0; corner < 4 && plain; corner++)
			{
				const double x = (corner & 1) ? static_cast<double>(width) : 0.0, y = (corner & 2) ? static_cast<double>(height) : 0.0;
				const double ws = sign * (n0w + n1w * x + n2w * y), fs = ws + sign * zw;
				if (corner == 0)
					first_w = ws;
				plain = std::fabs(ws) >= 0x1.0p-50 && std::fabs(ws) <= 0x1.0p50 && std::fabs(fs) >= 0x1.0p-50 && std::fabs(fs) <= 0x1.0p50 && (ws > 0.0) == (first_w > 0.0) && (ws > 0.0) == (fs > 0.0);
			}
			if (plain)
				f.eye_form = 2u;
		}
	}
	f.pinhole = 0u;
	if (f.mx[3] == 0.0f && f.my[3] == 0.0f && f.k_near[3] != 0.0f && f.k_far[3] != 0.0f && std::isfinite(f.k_near[3]) && std::isfinite(f.k_far[3]))
	{
		const double sx = 2.0 / static_cast<double>(width), sy = -(2.0 / static_cast<double>(height));
		const double iwn = 1.0 / static_cast<double>(f.k_near[3]), iwf = 1.0 / static_cast<double>(f.k_far[3]);
		double o0[3], o1[3], o2[3], d0[3], d1[3], d2[3];
		for (int c = 0; c < 3; c++)
		{
			const double mx = f.mx[c], my = f.my[c], kn = f.k_near[c], kf = f.k_far[c];
			o1[c] = mx * sx * iwn, o2[c] = my * sy * iwn, o0[c] = (kn - mx + my) * iwn;
			const double e1 = mx * sx * iwf, e2 = my * sy * iwf, e0 = (kf - mx + my) * iwf;
			d0[c] = e0 - o0[c], d1[c] = e1 - o1[c], d2[c] = e2 - o2[c];
		}
		const double dd = d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2] + d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2];
		const double od = o1[0] * d1[0] + o1[1] * d1[1] + o1[2] * d1[2] + o2[0] * d2[0] + o2[1] * d2[1] + o2[2] * d2[2];
		const double kappa = od / dd;
		double worst = 0.0, scale = 0.0;
		for (int c = 0; c < 3; c++)
		{
			worst = std::fmax(worst, std::fmax(std::fabs(o1[c] - kappa * d1[c]), std::fabs(o2[c] - kappa * d2[c])));
			scale = std::fmax(scale, std::fmax(std::fabs(o1[c]), std::fabs(o2[c])));
		}
		if (dd > 0.0 && kappa >= 0x1.0p-20 && kappa <= 0x1.0p20 && worst <= 1.0e-5 * scale) // (a NaN anywhere fails a comparison; kappa = near / (far - near) scales the vector the kernels normalise, so its sign and size matter)
		{
			f.pinhole = 1u;
			for (int c = 0; c < 3; c++)
			{
				// (the vector the kernels carry is kappa * (far - near) = near - eye: the near point is then eye + it, one addition)
					f.ray_d0[c] = static_cast<float>(kappa * d0[c]), f.ray_d1[c] = static_cast<float>(kappa * d1[c]), f.ray_d2[c] = static_cast<float>(kappa * d2[c]);
				f.ray_j1[c] = f.ray_d1[c] * 0x1.0p-24f, f.ray_j2[c] = f.ray_d2[c] * 0x1.0p-24f;
				f.ray_eye[c] = static_cast<float>(o0[c] - kappa * d0[c]);
			}
		}
	}

	bool rolling_items = false; // the persistent big-scene kernels draw items from a sequence whose head must start at 0
	rolling_buffers rolling;
	if (!(flags & RT_HIP_FLAG_PREVIEW))
	{
		const uint32_t variant = choose_kernel(ctx->scene, flags, f.samples_per_pixel, f.pinhole != 0 || f.eye_form == 2u, static_cast<uint64_t>(width) * f.local_rows);
		const bool big_scene = variant == RT_HIP_KERNEL_TILED || variant == RT_HIP_KERNEL_STREAMED;
		rolling_items = big_scene;
		queue_params queue = choose_queue(f.samples_per_pixel, width, f.local_rows, big_scene, host_frame, half_chunk_choice(flags), ctx->scene.n_spheres + ctx->scene.n_planes, variant == RT_HIP_KERNEL_STREAMED && ctx->scene.n_spheres >= sparse_launch_min_spheres);
		// small scenes: a pixel's chunk sums (one per 16 samples) are parked in LDS until the pixel is complete
		const uint64_t slot_bytes = big_scene ? 0u : 4ull * tile_slot_bytes(queue);
		if (slot_bytes > 48u * 1024u)
			return fail(RT_HIP_UNSUPPORTED, "rt_hip_render_device: %u samples per pixel are more than the kernels hold chunk sums for (4096; the reference clamps to 1000, src/scene.cpp:544)", f.samples_per_pixel);
		// big scenes: they meet in HBM, 16 bytes per chunk (or per sample) of this rank's rows
		size_t sums_bytes = 0, done_bytes = 0;
		rolling_buffer_bytes(queue, f.samples_per_pixel, width, f.local_rows, big_scene, sums_bytes, done_bytes);
		if (sums_bytes > ctx->item_sums.bytes && queue.halves && big_scene && ctx->item_sums.reserve(sums_bytes) != hipSuccess)
		{
			// no room for a slot per SAMPLE (up to 8 GiB): whole chunks need a sixteenth of it.  The launch code makes the
			// same choice from the same flag.
			(void)hipGetLastError();
			flags |= RT_HIP_FLAG_FORCE_WHOLE_CHUNKS;
			flags &= ~static_cast<uint32_t>(RT_HIP_FLAG_FORCE_HALF_CHUNKS);
			queue = choose_queue(f.samples_per_pixel, width, f.local_rows, big_scene, host_frame, half_chunk_choice(flags), ctx->scene.n_spheres + ctx->scene.n_planes, variant == RT_HIP_KERNEL_STREAMED && ctx->scene.n_spheres >= sparse_launch_min_spheres);
			rolling_buffer_bytes(queue, f.samples_per_pixel, width, f.local_rows, big_scene, sums_bytes, done_bytes);
		}
		if (sums_bytes > (64ull << 30))
			return fail(RT_HIP_UNSUPPORTED, "rt_hip_render_device: %ux%u at %u samples per pixel needs %zu GiB for the chunk sums of a scene of this size", width, height, f.samples_per_pixel, sums_bytes >> 30);
		if (sums_bytes)
		{
			RT_HIP_TRY(ctx->item_sums.reserve(sums_bytes));
			RT_HIP_TRY(ctx->pixel_done.reserve(done_bytes));
			// The arrival counters start every launch at zero — set here, on the launch's own stream, not left behind by the
			// previous launch: a launch that did not run to its end (a failed or aborted one) must not cost later frames
			// their pixels.  (8 MB at 1080p in front of a launch of milliseconds to seconds.)
			RT_HIP_TRY(hipMemsetAsync(ctx->pixel_done.ptr, 0, done_bytes, s));
			rolling.item_sums = ctx->item_sums.as<unsigned long long>();
			rolling.pixel_done = ctx->pixel_done.as<uint32_t>();
		}
	}
	device_counters* const counters = ctx->counters.as<device_counters>();
	if (keep_stats)
	{
		RT_HIP_TRY(hipMemsetAsync(counters, 0, sizeof(device_counters), s));
		RT_HIP_TRY(hipEventRecord(ctx->render_begin, s));
	}
	else if (rolling_items)
		RT_HIP_TRY(hipMemsetAsync(&counters->next_item, 0, sizeof(counters->next_item), s)); // (seconds-long launches: not launch-bound)
	uint32_t variant = RT_HIP_KERNEL_PREVIEW;
	if (flags & RT_HIP_FLAG_PREVIEW)
		launch_preview(f, ctx->scene, d_rgba8, d_rgb_f32, counters, s);
	else if (flags & RT_HIP_FLAG_FAST)
		variant = launch_render_fast(f, ctx->scene, ctx->small, flags | (host_frame ? launch_flag_host_frame : 0u), d_rgba8, d_rgb_f32, counters, rolling, ctx->compute_units, ctx->cache, s);
	else
		variant = launch_render(f, ctx->scene, (flags & RT_HIP_FLAG_SM_MATERIALS) ? ctx->small_sm : ctx->small, flags | (host_frame ? launch_flag_host_frame : 0u), d_rgba8, d_rgb_f32, counters, rolling, ctx->compute_units, ctx->cache, s);
	RT_HIP_TRY(hipGetLastError());
	ctx->launched = true;
	ctx->last_stream = s;
	if (keep_stats)
	{
		RT_HIP_TRY(hipEventRecord(ctx->render_end, s));
		// the counters follow the kernel to the host on the same stream: reading them later costs no transfer of its own
		RT_HIP_TRY(hipMemcpyAsync(ctx->counters_host, counters, sizeof(device_counters), hipMemcpyDeviceToHost, s));
		RT_HIP_TRY(hipEventRecord(ctx->counters_copied, s));
	}
	ctx->render_recorded = keep_stats;
	ctx->stats.kernel_variant = variant;
	ctx->stats.primary_samples = static_cast<uint64_t>(f.local_rows) * width * ((flags & RT_HIP_FLAG_PREVIEW) ? 1u : f.samples_per_pixel);
	if (!keep_stats) // the counters of this frame were not kept: nothing stale may be reported for it
	{
		ctx->stats.render_ms = 0.0f;
		ctx->stats.segments = ctx->stats.sphere_tests = ctx->stats.plane_tests = 0;
	}
	return ok();
}
}

extern "C" rt_hip_status rt_hip_assemble_device(rt_hip_ctx* ctx,
												uint32_t width,
												uint32_t height,
												uint32_t world,
												uint32_t stripe_rows,
												const uint32_t* d_gathered,
												uint32_t* d_frame,
												void* stream)
{
	if (!ctx || !d_gathered || !d_frame)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_assemble_device: NULL argument");
	if (!width || !height || !world || !stripe_rows)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_assemble_device: zero size");
	RT_HIP_TRY(hipSetDevice(ctx->device));
	const rt_hip_partition p = { 0, world, stripe_rows };
	uint32_t padded = 0;
	if (const rt_hip_status st = rt_hip_padded_local_rows(height, &p, &padded))
		return st;
	launch_assemble(width, height, world, stripe_rows, padded, d_gathered, d_frame, 0u, false, static_cast<hipStream_t>(stream));
	RT_HIP_TRY(hipGetLastError());
	return ok();
}

namespace rt_hip
{
	// synchronise with the member's last launch and read its counters into member->stats
	rt_hip_status fetch_member_stats(rt_hip_ctx* ctx)
	{
		RT_HIP_TRY(hipSetDevice(ctx->device));
		if (ctx->render_recorded)
		{
			RT_HIP_TRY(hipEventSynchronize(ctx->counters_copied));
			float ms = 0.0f;
			RT_HIP_TRY(hipEventElapsedTime(&ms, ctx->render_begin, ctx->render_end));
			ctx->stats.render_ms = ms;
			uint64_t segments = 0;
			for (const unsigned long long part : ctx->counters_host->segments)
				segments += part;
			ctx->stats.segments = segments;
			ctx->stats.sphere_tests = segments * ctx->scene.n_spheres;
			ctx->stats.plane_tests = segments * ctx->scene.n_planes;
		}
		return ok();
	}
}

namespace rt_hip
{
	rt_hip_stats stats_of_group_rank(const rt_hip_ctx* ctx, uint32_t rank)
	{
		const frame_group_rank& line = ctx->group->block->ranks[rank];
		rt_hip_stats out{};
		out.primary_samples = line.primary_samples;
		out.segments = line.segments;
		out.sphere_tests = line.sphere_tests;
		out.plane_tests = line.plane_tests;
		out.render_ms = line.render_ms;
		out.upload_ms = line.upload_ms;
		out.kernel_variant = line.kernel_variant;
		return out;
	}

	// whole-frame counters of a frame group: counts summed over the ranks, times the slowest rank's
	void sum_group_stats(const rt_hip_ctx* ctx, rt_hip_stats* out)
	{
		for (uint32_t r = 0; r < ctx->world; r++)
		{
			if (r == ctx->first_rank)
				continue;
			const rt_hip_stats other = stats_of_group_rank(ctx, r);
			out->primary_samples += other.primary_samples;
			out->segments += other.segments;
			out->sphere_tests += other.sphere_tests;
			out->plane_tests += other.plane_tests;
			out->render_ms = std::max(out->render_ms, other.render_ms);
			out->upload_ms = std::max(out->upload_ms, other.upload_ms);
		}
	}
}

extern "C" rt_hip_status rt_hip_stats_fetch(rt_hip_ctx* ctx, rt_hip_stats* out_stats)
{
	if (!ctx || !out_stats)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_stats_fetch: NULL argument");
	if (const rt_hip_status st = fetch_member_stats(ctx))
		return st;
	*out_stats = ctx->stats;
	if (ctx->group) // the other ranks' shares are in the group's block (valid between two frames)
		sum_group_stats(ctx, out_stats);
	// several GPUs: the frame's counts are the sum over the members' shares, its kernel time the slowest member's
	for (rt_hip_ctx* member : ctx->peers)
	{
		if (const rt_hip_status st = fetch_member_stats(member))
			return st;
		out_stats->primary_samples += member->stats.primary_samples;
		out_stats->segments += member->stats.segments;
		out_stats->sphere_tests += member->stats.sphere_tests;
		out_stats->plane_tests += member->stats.plane_tests;
		out_stats->render_ms = std::max(out_stats->render_ms, member->stats.render_ms);
		out_stats->upload_ms = std::max(out_stats->upload_ms, member->stats.upload_ms);
	}
	if (!ctx->peers.empty())
		RT_HIP_TRY(hipSetDevice(ctx->device));
	return ok();
}

extern "C" rt_hip_status rt_hip_member_stats(rt_hip_ctx* ctx, int rank, rt_hip_stats* out_stats)
{
	if (ctx && ctx->group && out_stats && rank >= 0 && rank < static_cast<int>(ctx->world))
	{
		*out_stats = stats_of_group_rank(ctx, static_cast<uint32_t>(rank)); // (that rank's share of the most recent frame that kept stats)
		return ok();
	}
	rt_hip_ctx* member = member_of(ctx, rank);
	if (!member || !out_stats)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_member_stats: invalid argument");
	if (const rt_hip_status st = fetch_member_stats(member))
		return st;
	*out_stats = member->stats;
	RT_HIP_TRY(hipSetDevice(ctx->device));
	return ok();
}

extern "C" rt_hip_status rt_hip_phases_fetch(rt_hip_ctx* ctx, rt_hip_phases* out_phases)
{
	if (!ctx || !out_phases)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_phases_fetch: NULL argument");
	*out_phases = ctx->phases;
	return ok();
}

extern "C" rt_hip_status rt_hip_render(rt_hip_ctx* ctx,
									   const rt_hip_scene* scene,
									   uint32_t* pixels_rgba8888,
									   uint32_t width,
									   uint32_t height,
									   uint64_t seed,
									   uint32_t flags,
									   float* rgb_f32,
									   rt_hip_stats* stats)
{
	const auto entered = std::chrono::steady_clock::now();
	if (!ctx || !scene || (!pixels_rgba8888 && !(ctx->multi && ctx->first_rank != 0) && !ctx->group))
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_render: NULL argument");
	if (!width || !height)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_render: empty frame %ux%u", width, height);
	if (flags & ~(render_flag_mask | static_cast<uint32_t>(RT_HIP_FLAG_PERSISTENT_FRAME | RT_HIP_FLAG_STATS)))
		return fail(RT_HIP_UNSUPPORTED, "rt_hip_render: unknown flag bits 0x%x", flags);
	const size_t pixels = static_cast<size_t>(width) * height;
	const size_t frame_bytes = pixels * sizeof(uint32_t);
	const bool keep_stats = stats || (flags & RT_HIP_FLAG_STATS);
	try
	{
		RT_HIP_TRY(hipSetDevice(ctx->device));
		if (ctx->group)
			return render_group(ctx, scene, pixels_rgba8888, width, height, seed, flags, rgb_f32, stats, entered);
		if (pixels_rgba8888 && ctx->multi && ctx->direct_frame && ctx->peers.size() + 1 == ctx->world)
		{
			std::vector<int> nodes(1, ctx->numa_node); // every member stores its own stripes: each stripe on its member's node
			for (const rt_hip_ctx* member : ctx->peers)
				nodes.push_back(member->numa_node);
			track_frame_buffer(ctx, pixels_rgba8888, frame_bytes, (flags & RT_HIP_FLAG_PERSISTENT_FRAME) != 0, true, &nodes, width, height);
		}
		else if (pixels_rgba8888)
			track_frame_buffer(ctx, pixels_rgba8888, frame_bytes, (flags & RT_HIP_FLAG_PERSISTENT_FRAME) != 0);
		if (ctx->multi)
			return render_multi(ctx, scene, pixels_rgba8888, width, height, seed, flags, rgb_f32, stats, entered);

		const auto scene_t0 = std::chrono::steady_clock::now();
		scene_request request;
		if (const rt_hip_status st = open_request(request, scene))
			return st;
		ctx->phases = rt_hip_phases{};
		if (const rt_hip_status st = make_resident(ctx, request))
			return st;
		ctx->stats.upload_ms = static_cast<float>(seconds_since(scene_t0) * 1e3); // the frame's whole scene check, fingerprint pass included
		// Where the kernel stores its finished pixels (4 bytes per pixel over PCIe while the rest of the frame is still being
		// traced): the caller's own buffer if it is page-locked and mapped (RT_HIP_FLAG_PERSISTENT_FRAME), else the module's
		// own page-locked frame, from which the carrier's threads take them on into the caller's buffer (frame.hip).
		uint32_t* d_frame = nullptr;
		if (ctx->pinned_frame)
		{
			void* device_view = nullptr;
			if (hipHostGetDevicePointer(&device_view, pixels_rgba8888, 0) == hipSuccess && device_view)
				d_frame = static_cast<uint32_t*>(device_view);
			else
				(void)hipGetLastError();
		}
		frame_delivery* const delivery = delivery_of(ctx);
		if (!delivery)
			return fail(RT_HIP_RUNTIME_ERROR, "rt_hip_render: out of host memory");
		struct abandon_unless_finished // any early exit below: no thread keeps copying, the staging frame is wiped before its next use
		{
			frame_delivery* delivery;
			~abandon_unless_finished()
			{
				if (delivery)
					delivery->abandon();
			}
		} staged{ nullptr };
		// Declared AFTER `staged`, so that it runs FIRST on every early exit: whatever this call has put on the stream has
		// drained before the staging frame is abandoned (wiped, or on growth unmapped, by the next begin()).  render_device can
		// fail AFTER its kernel is enqueued (hipGetLastError, the timing events, the counters' copy): without this the device
		// would go on storing pixels into a frame the zero-means-pending protocol has already handed to the next call — or
		// into unmapped memory.  (multi.hip's settle_members is the same rule for several devices.)
		struct drain_before_abandoning
		{
			hipStream_t stream;
			bool armed;
			~drain_before_abandoning()
			{
				if (armed && hipStreamSynchronize(stream) != hipSuccess)
					(void)hipGetLastError();
			}
		} drain{ ctx->stream, false };
		if (!d_frame)
		{
			if (const rt_hip_status st = delivery->begin(pixels_rgba8888, pixels, &d_frame))
				return st;
			staged.delivery = delivery;
		}
		const size_t rgb_bytes = pixels * 3 * sizeof(float);
		if (rgb_f32)
		{
			RT_HIP_TRY(ctx->frame_rgb.reserve(rgb_bytes));
			RT_HIP_TRY(ctx->staging_rgb.reserve(rgb_bytes));
		}
		drain.armed = true; // from here on the device may be storing into host memory: no return before the stream has drained
		if (const rt_hip_status st = render_device(ctx, width, height, seed, flags & render_flag_mask, nullptr, d_frame, rgb_f32 ? ctx->frame_rgb.as<float>() : nullptr, ctx->stream, false, keep_stats, true))
			return st;
		if (staged.delivery)
			staged.delivery->launched();
		hipError_t e = hipSuccess;
		if (rgb_f32) // the float mean lands in the module's own page-locked buffer and is copied on from there
			e = hipMemcpyAsync(ctx->staging_rgb.ptr, ctx->frame_rgb.ptr, rgb_bytes, hipMemcpyDeviceToHost, ctx->stream);
		const auto issued = std::chrono::steady_clock::now();
		if (e == hipSuccess && keep_stats)
			e = hipEventSynchronize(ctx->render_end);
		const auto t0 = std::chrono::steady_clock::now();
		const hipError_t drained = hipStreamSynchronize(ctx->stream);
		drain.armed = drained != hipSuccess; // (a failed wait is tried once more on the way out)
		RT_HIP_TRY(e);
		RT_HIP_TRY(drained);
		if (staged.delivery)
		{
			staged.delivery->finish(); // the last tiles' pixels; returns with the whole frame in the caller's buffer
			ctx->phases.carrier_bands = static_cast<uint32_t>(staged.delivery->carrier.bands());
			ctx->phases.carrier_bands_early = static_cast<uint32_t>(staged.delivery->carrier.early_bands());
			ctx->phases.carrier_helpers = staged.delivery->carrier.helpers();
			staged.delivery = nullptr;
		}
		if (rgb_f32)
			delivery->carrier.copy(rgb_f32, ctx->staging_rgb.ptr, rgb_bytes);
		ctx->stats.readback_ms = keep_stats ? static_cast<float>(seconds_since(t0) * 1e3) : 0.0f;
		ctx->phases.host_issue_ms = static_cast<float>(std::chrono::duration<double>(issued - entered).count() * 1e3);
		ctx->phases.host_wait_ms = static_cast<float>(seconds_since(issued) * 1e3);
		if (keep_stats)
			ctx->phases.render_ms = elapsed_or_zero(ctx->render_begin, ctx->render_end);
		if (stats)
			return rt_hip_stats_fetch(ctx, stats);
		return ok();
	}
	catch (const std::exception& e) // nothing may propagate through the C boundary
	{
		return fail(RT_HIP_RUNTIME_ERROR, "rt_hip_render: %s", e.what());
	}
}
