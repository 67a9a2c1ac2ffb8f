// rt_amd/csrc/internal.hpp — what the translation units of librt_hip.so share: the context, error translation, the
// grow-only device / page-locked buffers, and the functions one unit offers the others.
//
//   context.hip   library entry points, create / destroy, multi-GPU and rank contexts, RCCL and frame-group joins
//   scene.hip     the caller's columns: checks, fingerprint, the HBM image, residency
//   frame.hip     the caller's frame buffer: the module-owned page-locked frame and its delivery threads (the default),
//                 the opt-in page-lock of the caller's own buffer, NUMA placement
//   render.hip    one launch (rt_hip_render_device), the work counters, the single-GPU drop-in rt_hip_render
//   multi.hip     rt_hip_render on several GPUs: stripes, one gather, assemble
//   group.hip     rt_hip_render as one rank of a frame group (rank processes storing into one shared back buffer)
//   kernels.hip   the gfx950 kernels (compiled twice: parity contract and RT_HIP_FLAG_FAST arithmetic)
//
// Everything here has hidden visibility: the library exports the C entry points of include/rt_hip.h and nothing else.
// The test-only library librt_hip_kat.so (kat.hip, kat_kernels.hip) includes this header for the LAYOUT of rt_hip_ctx —
// it reads a context's device and resident scene — and calls nothing of librt_hip.so that is not in rt_hip.h.
#pragma once

#include "../../include/rt_hip.h"
#include "contract.hpp"
#include "delivery.hpp"
#include "frame_group.hpp"
#include "kernels.hpp"

#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace rt_hip
{
	// ---- error translation: no exception and no HIP error code crosses the C boundary (src/renderer.hpp:11: noexcept) ----
	rt_hip_status fail(rt_hip_status status, const char* format, ...) __attribute__((format(printf, 2, 3)));
	const std::string& last_error();
	inline rt_hip_status ok() { return RT_HIP_OK; }

#define RT_HIP_TRY_NCCL(expr)                                                                                          \
	do                                                                                                                 \
	{                                                                                                                  \
		const ncclResult_t rt_hip_try_res = (expr);                                                                    \
		if (rt_hip_try_res != ncclSuccess)                                                                             \
			return ::rt_hip::fail(RT_HIP_RUNTIME_ERROR, "%s failed: %s", #expr, ncclGetErrorString(rt_hip_try_res));   \
	}                                                                                                                  \
	while (false)

#define RT_HIP_TRY(expr)                                                                                               \
	do                                                                                                                 \
	{                                                                                                                  \
		const hipError_t rt_hip_try_err = (expr);                                                                      \
		if (rt_hip_try_err != hipSuccess)                                                                              \
			return ::rt_hip::fail(RT_HIP_RUNTIME_ERROR, "%s failed: %s", #expr, hipGetErrorString(rt_hip_try_err));    \
	}                                                                                                                  \
	while (false)

	// grow-only device allocation
	struct device_buffer
	{
		void* ptr = nullptr;
		size_t bytes = 0;

		hipError_t reserve(size_t wanted)
		{
			if (wanted <= bytes)
				return hipSuccess;
			if (ptr)
			{
				(void)hipFree(ptr);
				ptr = nullptr;
				bytes = 0;
			}
			const hipError_t e = hipMalloc(&ptr, wanted);
			if (e == hipSuccess)
				bytes = wanted;
			return e;
		}
		void release()
		{
			if (ptr)
				(void)hipFree(ptr);
			ptr = nullptr;
			bytes = 0;
		}
		template <typename T>
		T* as() const
		{
			return static_cast<T*>(ptr);
		}
	};

	// grow-only page-locked HOST allocation owned by the module (hipHostMalloc: mapped into every device's address space).
	// Whatever the HIP runtime reads from or writes to host memory on this module's behalf is one of these — never the
	// caller's pageable memory, and never a pageable block of the module's own (see frame.hip, "Why the module stages").
	struct pinned_buffer
	{
		void* ptr = nullptr;
		size_t bytes = 0;

		hipError_t reserve(size_t wanted)
		{
			if (wanted <= bytes)
				return hipSuccess;
			release();
			// a little slack, so that a window dragged larger pixel row by pixel row does not re-allocate every frame
			const size_t rounded = (wanted + wanted / 8u + 4095u) & ~static_cast<size_t>(4095u);
			const hipError_t e = hipHostMalloc(&ptr, rounded, hipHostMallocMapped | hipHostMallocPortable);
			if (e == hipSuccess)
				bytes = rounded;
			else
				ptr = nullptr;
			return e;
		}
		void release()
		{
			if (ptr)
				(void)hipHostFree(ptr);
			ptr = nullptr;
			bytes = 0;
		}
		template <typename T>
		T* as() const
		{
			return static_cast<T*>(ptr);
		}
	};

	inline double seconds_since(std::chrono::steady_clock::time_point t0)
	{
		return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	}

	inline bool valid_partition(const rt_hip_partition& p)
	{
		return p.world && p.rank < p.world && p.stripe_rows;
	}

	inline uint32_t local_rows_of(uint32_t height, uint32_t rank, uint32_t world, uint32_t stripe_rows)
	{
		const uint32_t stripes = (height + stripe_rows - 1) / stripe_rows;
		uint32_t rows = 0;
		for (uint32_t b = rank; b < stripes; b += world)
		{
			const uint32_t y0 = b * stripe_rows;
			rows += (height - y0 < stripe_rows) ? height - y0 : stripe_rows;
		}
		return rows;
	}

	inline float elapsed_or_zero(hipEvent_t from, hipEvent_t to)
	{
		float ms = 0.0f;
		if (hipEventElapsedTime(&ms, from, to) != hipSuccess)
		{
			(void)hipGetLastError();
			ms = 0.0f;
		}
		return ms;
	}

	struct frame_delivery; // below: the module-owned frame and the threads that carry it to the caller's buffer

	// (RT_HIP_FLAG_PERSISTENT_FRAME and RT_HIP_FLAG_STATS are rt_hip_render's own: they do not travel to the launch)
	constexpr uint32_t render_flag_mask = RT_HIP_FLAG_FORCE_TILED | RT_HIP_FLAG_FORCE_RESIDENT | RT_HIP_FLAG_SM_MATERIALS | RT_HIP_FLAG_PREVIEW | RT_HIP_FLAG_FORCE_STREAMED | RT_HIP_FLAG_FAST | RT_HIP_FLAG_FORCE_HALF_CHUNKS | RT_HIP_FLAG_FORCE_WHOLE_CHUNKS;

}

struct rt_hip_ctx
{
	int device = 0;
	uint32_t compute_units = 256;
	int numa_node = -1; // host NUMA node the GPU hangs off (sysfs), -1 = unknown

	// the scene, resident in HBM: one buffer holding every column back to back (256-byte aligned starts)
	rt_hip::device_buffer scene_columns;
	rt_hip::pinned_buffer scene_staging; // host image of that block on its way up (built here, copied from here)
	rt_hip::device_scene scene{};
	rt_hip::small_scene small{}, small_sm{}; // host copies of the kernel-argument scene of the `small` kernel (mg / sm scatter tables)
	bool have_scene = false;
	uint32_t samples_per_pixel = 0, max_bounces = 0;
	float inverse_view_projection[16]{};

	// big scenes: chunk sums on their way between waves and the pixels' arrival counters (kernels.hpp, rolling_buffers)
	rt_hip::device_buffer item_sums, pixel_done;
	rt_hip::device_buffer counters;
	rt_hip::device_counters* counters_host = nullptr; // page-locked; filled by an asynchronous copy right behind every launch that keeps stats
	hipEvent_t render_begin = nullptr, render_end = nullptr, counters_copied = nullptr;
	bool render_recorded = false; // the most recent launch was bracketed by the events above and its counters were read back
	bool launched = false;		  // something has been launched on last_stream
	hipStream_t last_stream = nullptr;
	// exchange phases of a multi-GPU frame on the root's stream (frames that keep stats): stripes on the root, frame
	// assembled, frame copied to the host
	hipEvent_t gathered = nullptr, assembled = nullptr, copied = nullptr;
	rt_hip_phases phases{};
	rt_hip::launch_cache cache; // what the launch code remembers per context (occupancy of the persistent kernels)

	// the context's own stream: everything rt_hip_render() enqueues goes here (never the process-wide null stream)
	hipStream_t stream = nullptr;

	// ---- several GPUs behind one render() (rt_hip_create_multi) ----
	// The context the caller holds is member 0 (the root); it owns members 1 .. n-1, one per further device.
	bool multi = false;
	bool peer_copy = false;			  // RT_HIP_MULTI_PEER_COPY
	bool direct_frame = false;		  // RT_HIP_MULTI_DIRECT_FRAME
	// the renderer's ranks: `world` in all, of which this process holds first_rank .. first_rank + members - 1
	// (rt_hip_create_multi: all of them; rt_hip_create_rank: exactly one, the others live in other processes)
	uint32_t world = 1, first_rank = 0;
	std::vector<rt_hip_ctx*> peers;	  // members 1 .. n-1
	std::vector<ncclComm_t> comms;	  // one communicator per member, rank order (empty with peer_copy)
	rt_hip::device_buffer stripes_rgba, stripes_rgb;   // this member's compact stripe buffers (the gather's send side)
	rt_hip::device_buffer gathered_rgba, gathered_rgb; // root: n x padded stripes, rank order (the gather's receive side)
	hipEvent_t stripes_ready = nullptr;				   // recorded on `stream` after this member's launch (peer copies wait for it)
	// ---- or: one rank of a renderer whose ranks are processes that all map the caller's back buffer (rt_hip_join_frame_group)
	std::unique_ptr<rt_hip::frame_group> group;

	// ---- the frame's way to the caller (frame.hip) ----
	// Default: the module's OWN page-locked, mapped frame — the kernels store finished pixels straight into it — and a few
	// host threads that carry the pixels on into the caller's buffer while the rest of the frame is still being traced.
	std::unique_ptr<rt_hip::frame_delivery> delivery;
	rt_hip::device_buffer frame_rgb;	 // the float mean of the whole frame in HBM (optional second output)
	rt_hip::pinned_buffer staging_rgb;	 // ... and its landing place on the host
	uint64_t scene_fingerprint = 0;		 // of the caller's columns last uploaded: rt has no scene version counter (src/main.cpp:233-311)
	size_t scene_bytes = 0;				 // size of the resident block
	// Opt-in (RT_HIP_FLAG_PERSISTENT_FRAME, frame groups): the CALLER's buffer, page-locked and mapped into the GPU's address
	// space while it keeps arriving at the same address: the kernel renders straight into it.  The caller then owes the
	// module rt_hip_forget_frame before that memory is unmapped (rt_hip.h).
	void* pinned_frame = nullptr;
	size_t pinned_bytes = 0;
	void* refused_frame = nullptr; // a buffer whose page-lock failed: not tried again while it keeps arriving
	size_t refused_bytes = 0;

	const void* asked_pointer = nullptr; // rt_hip_render_device: the last output pointer and whether it is host memory
	bool asked_pointer_is_host = false;

	rt_hip_stats stats{};

	rt_hip_ctx() = default;
	rt_hip_ctx(const rt_hip_ctx&) = delete;
	rt_hip_ctx& operator=(const rt_hip_ctx&) = delete;
};

namespace rt_hip
{
	// ---- context.hip ----
	rt_hip_ctx* member_of(rt_hip_ctx* ctx, int rank);

	// ---- scene.hip ----
	// layout of the single HBM block: every column starts on a 256-byte boundary.  A function of the four counts alone.
	struct scene_layout
	{
		size_t scx, scy, scz, sr, sm;				 // sphere columns
		size_t pnx, pny, pnz, pd, pm;				 // plane columns
		size_t shading, type;						 // per material
		size_t geometry, prim_shading, prim_metal;	 // derived per-primitive tables (spheres, then planes)
		size_t prim_shading_sm, prim_scatter_sm;	 // the same under sm_ray_tracer's scatter table
		size_t box_bounds, albedo;					 // what only the preview reads
		size_t total;
	};
	// One render()'s view of the caller's scene: pointers checked, fingerprint taken once — shared by all members of a
	// multi-GPU context — and the image of the HBM block, which is built (in the first member's page-locked staging
	// buffer) only if some member turns out not to hold these columns yet.
	struct scene_request
	{
		const rt_hip_scene* scene = nullptr;
		scene_layout layout{};
		uint64_t print = 0;
		bool indices_checked = false;
		const unsigned char* image = nullptr; // page-locked host image of the block (one H2D copy per member that needs it)
		small_scene small{}, small_sm{};	  // the kernel-argument scene of the `small` kernel (mg / sm scatter tables)
	};
	rt_hip_status open_request(scene_request& r, const rt_hip_scene* scene);
	// Make `ctx` hold the request's scene (leaves ctx->device current)
	rt_hip_status make_resident(rt_hip_ctx* ctx, scene_request& r);

	// ---- render.hip ----
	// whole_frame_buffers: d_rgba8 / d_rgb_f32 are the whole width x height frame and every pixel goes to its image row
	// (several GPUs rendering into one host frame); otherwise the rank's compact stripe buffer, as the public call documents.
	// keep_stats: bracket the launch with timing events, zero the work counters before it and read them back after it.  Without
	// it NOTHING but the kernel is enqueued (the plug-in's call: rt_hip_render with stats == NULL).
	// host_frame: d_rgba8 is page-locked host memory (a mapped frame): the tiles are cut for PCIe writes (choose_queue).
	rt_hip_status render_device(rt_hip_ctx* ctx, uint32_t width, uint32_t height, uint64_t seed, uint32_t flags, const rt_hip_partition* part, uint32_t* d_rgba8, float* d_rgb_f32, void* stream, bool whole_frame_buffers, bool keep_stats, bool host_frame);
	rt_hip_status fetch_member_stats(rt_hip_ctx* ctx);
	rt_hip_stats stats_of_group_rank(const rt_hip_ctx* ctx, uint32_t rank);
	void sum_group_stats(const rt_hip_ctx* ctx, rt_hip_stats* out);

	// ---- multi.hip / group.hip ----
	rt_hip_status render_multi(rt_hip_ctx* root, const rt_hip_scene* scene, uint32_t* pixels_rgba8888, uint32_t width, uint32_t height, uint64_t seed, uint32_t flags, float* rgb_f32, rt_hip_stats* stats, std::chrono::steady_clock::time_point entered);
	rt_hip_status render_group(rt_hip_ctx* ctx, const rt_hip_scene* scene, uint32_t* pixels_rgba8888, uint32_t width, uint32_t height, uint64_t seed, uint32_t flags, float* rgb_f32, rt_hip_stats* stats, std::chrono::steady_clock::time_point entered);

	// ---- frame.hip ----
	int numa_node_of(int device);
	bool debug_frame();
	void place_stripes(void* ptr, size_t bytes, uint32_t width, uint32_t height, uint32_t stripe_rows, const std::vector<int>& node_of_rank);
	void unpin_frame(rt_hip_ctx* ctx);
	// `pin`: RT_HIP_FLAG_PERSISTENT_FRAME — page-lock the caller's buffer on first sight, keep the lock while the same buffer
	// keeps arriving; any other buffer (or no flag) first drops the old registration.  `stripe_nodes` (with the frame's
	// shape): place the row stripes on their owners' nodes instead of the whole buffer on this context's.
	void track_frame_buffer(rt_hip_ctx* ctx, uint32_t* pixels, size_t bytes, bool pin, bool may_move_pages = true, const std::vector<int>* stripe_nodes = nullptr, uint32_t width = 0, uint32_t height = 0);
	// page-locks on callers' memory this process holds right now (rt_hip_live_frame_locks)
	uint32_t live_frame_locks();

	// The module-owned frame of a context and its delivery to the caller's buffer (delivery.hpp has the idea).  Per frame:
	//   begin    the staging frame is in place (all zero) and the carrier's threads are looking at it; returns the frame as
	//            the current device sees it (the kernels' output pointer)
	//   ... launches that store packed pixels into it, hipStreamSynchronize ...
	//   finish   everything the device stored is visible: the rest of the frame is carried over; returns when every pixel
	//            is in the caller's buffer
	//   abandon  instead of finish, on any failure: nothing further is copied, the staging frame is wiped before its next use
	// The module-owned frame itself: anonymous memory of the module's own, placed on the host NUMA node the GPU hangs off
	// BEFORE it is first touched, then page-locked and mapped for every device (hipHostRegister: portable).  Not hipHostMalloc:
	// that puts the pages wherever the runtime likes; the module knows the node (sysfs), owns the memory and keeps the
	// carrier's helper threads on that node's CPUs, so frame, pollers and GPU sit on one socket (with the helpers left to the
	// scheduler the default mode cost the headline call + 0.02-0.03 ms and the 4K frame + 0.04-0.08 ms over the zero-copy
	// mode; pinned, + 0.00-0.02: profiles/r04/carrier_pinning_ab.txt).
	struct staging_frame
	{
		void* ptr = nullptr;
		size_t bytes = 0;
		bool registered = false;
		int view_device = -1;	   // the device `view` is the address on (the one that was current when it was asked for)
		void* view = nullptr;
		hipError_t reserve(size_t wanted, int numa_node); // grows (contents lost: the caller wipes); ptr all zero after a successful growth
		void release();
		~staging_frame() { release(); }
		template <typename T>
		T* as() const
		{
			return static_cast<T*>(ptr);
		}
	};

	struct frame_delivery
	{
		staging_frame frame; // uint32 per pixel; all zero between frames
		int numa_node = -1;	 // where the frame's pages should live
		bool dirty = false;	 // not all zero (a failed frame): wiped by the next begin
		pixel_carrier carrier;

		frame_delivery(unsigned helpers, int node);
		// before the launch: the frame in place and all zero, its address as the current device sees it.  The carrier's threads
		// are not told yet —
		rt_hip_status begin(uint32_t* caller_pixels, size_t pixels, uint32_t** out_device_view, bool bottom_first = true);
		// — but right BEHIND the launch: waking them (a futex call, and however long the host's scheduler takes) then runs
		// beside the kernel instead of in front of it
		void launched()
		{
			carrier.announce();
		}
		// the same frame as `device` sees it (direct frames of several GPUs); leaves that device current; NULL on failure
		uint32_t* view_on(int device);
		void finish();
		void abandon()
		{
			carrier.abandon();
			dirty = true;
		}
	};
	// the context's delivery, made on first use (NULL: out of memory)
	frame_delivery* delivery_of(rt_hip_ctx* ctx);

	// After the first launch of a multi-GPU frame nothing may return before every member's stream has drained: a member
	// that is still storing into the frame (or into stripe buffers a later call would re-use) must not outlive the call
	// that reported the failure.  Also puts the root's device back as the current one.
	struct settle_members
	{
		rt_hip_ctx* root;
		bool armed = false;
		explicit settle_members(rt_hip_ctx* r) : root(r) {}
		settle_members(const settle_members&) = delete;
		settle_members& operator=(const settle_members&) = delete;
		~settle_members()
		{
			if (!armed)
				return;
			for (rt_hip_ctx* member : root->peers)
				if (hipSetDevice(member->device) == hipSuccess)
					(void)hipStreamSynchronize(member->stream);
			if (hipSetDevice(root->device) == hipSuccess)
				(void)hipStreamSynchronize(root->stream);
			(void)hipGetLastError();
		}
	};
}
