// rt_amd/csrc/frame.hip — the frame's way from the GPU into the caller's buffer.
//
// The reference's render() is handed a pageable host buffer (image_view: muu::aligned_alloc, src/image.cpp:9-13), fills it
// and returns; the caller presents it at once (src/window.cpp:215-216) and may free or re-create it at any time between two
// calls (resize: src/window.cpp:198-203).  Two ways here:
//
// DEFAULT — the module's own frame.  The kernels store finished pixels (system-scope, written through) into a page-locked,
// mapped frame that the MODULE allocated (its own anonymous mapping, first touched on the GPU's host node and registered with
// hipHostRegister: staging_frame below), and a few host threads carry them on into the caller's buffer
// while the rest of the frame is still being traced (delivery.hpp).  The caller's memory is only ever touched by ordinary
// CPU stores issued inside the call: nothing of it is registered, locked, mapped or remembered, so whatever the caller does
// with it between two calls — free it, get the same address back, hand it to somebody else — cannot reach this module.
//
// OPT-IN — RT_HIP_FLAG_PERSISTENT_FRAME.  The caller promises the buffer's lifetime (rt_hip.h) and the module page-locks and
// maps the CALLER's buffer: the kernel stores straight into it, no second copy of the frame exists anywhere.  Saves the
// carrier's last few tens of microseconds per frame; costs the caller a call to rt_hip_forget_frame before it unmaps the
// buffer (a mapping re-created at the same address is a GPU memory fault otherwise: profiles/r03/remap_without_forget.txt).
//
// Why the module stages (round 4).  Up to round 3 the default was a frame in HBM followed by
// hipMemcpyAsync(caller's pageable buffer, ..., hipMemcpyDeviceToHost).  What the HIP runtime does with a pageable
// destination is its own business and not a stable one: copies below a size threshold are staged through a runtime-owned
// buffer and finished by a CPU memcpy on a runtime thread, larger ones page-lock the caller's pages on the fly
// (hsa_amd_memory_lock) and keep such locks cached, keyed by host ADDRESS and size, after the copy has completed
// (GPU_PINNED_XFER_SIZE / GPU_PINNED_MIN_XFER_SIZE / GPU_STAGING_BUFFER_SIZE; "HSA Copy Using Pinned resource" vs
// "... Staging resource" in the runtime's log: profiles/r04/pageable_diag.txt has what this installation did).  Either way
// something outside this module holds a handle on caller memory AFTER rt_hip_render has returned, while a numpy caller
// munmap()s a 8 MB frame the moment it drops it and mmap()s the next array at the same address: one of round 3's test runs
// found a float64 array of the test itself full of packed RGBA8 words.  The module no longer takes part in that: every byte
// the HIP runtime moves to or from host memory on its behalf lives in page-locked memory the module itself allocated — frames
// (frame_delivery), the float mean (staging_rgb), the scene image (scene_staging), the work counters.
#include "internal.hpp"

#include <sched.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <new>

using namespace rt_hip;

namespace rt_hip
{
	// The host NUMA node of a GPU, from sysfs (-1 if it cannot be told).
	int numa_node_of(int device)
	{
		char bus_id[64] = {};
		if (hipDeviceGetPCIBusId(bus_id, sizeof(bus_id), device) != hipSuccess)
		{
			(void)hipGetLastError();
			return -1;
		}
		for (char* c = bus_id; *c; c++)
			if (*c >= 'A' && *c <= 'F')
				*c = static_cast<char>(*c - 'A' + 'a'); // sysfs spells the address in lower case
		char path[160];
		std::snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus_id);
		int node = -1;
		if (FILE* f = std::fopen(path, "r"))
		{
			if (std::fscanf(f, "%d", &node) != 1)
				node = -1;
			std::fclose(f);
		}
		return node;
	}

	// RT_HIP_DEBUG_FRAME=1: one line on stderr per page-lock event (diagnostics for integrators; off by default)
	bool debug_frame()
	{
		static const bool on = [] {
			const char* knob = std::getenv("RT_HIP_DEBUG_FRAME");
			return knob && knob[0] == '1';
		}();
		return on;
	}

	// Ask the kernel to move the pages of [ptr, ptr + bytes) to NUMA node `node` (mbind with MPOL_MF_MOVE; the raw
	// system call, so that nothing links libnuma).  The kernels render straight into the caller's back buffer: on a
	// two-socket host a buffer on the far socket makes every pixel store cross the socket interconnect — measured on an
	// MI355X box: 3.23 ms per frame and noisy against 2.96 ms and steady (profiles/r02/numa_probe.txt).  Addresses stay
	// what they are; only the physical placement changes.  Only the pages that lie WHOLLY inside the buffer are touched
	// (begin rounded up, end rounded down): a partial first or last page may hold the caller's neighbouring allocations,
	// whose placement is none of this module's business.  RT_HIP_NUMA_MOVE=0 in the environment turns the move off.
	// Best effort: any failure leaves the buffer where it was.
	static void place_on_node(void* ptr, size_t bytes, int node)
	{
#ifdef SYS_mbind
		if (node < 0 || node >= 1024 || !ptr || !bytes)
			return;
		if (const char* knob = std::getenv("RT_HIP_NUMA_MOVE"))
			if (knob[0] == '0' && knob[1] == '\0')
				return;
		const long page = sysconf(_SC_PAGESIZE);
		if (page <= 0)
			return;
		const uintptr_t mask_low = static_cast<uintptr_t>(page - 1);
		const uintptr_t begin = (reinterpret_cast<uintptr_t>(ptr) + mask_low) & ~mask_low;
		const uintptr_t end = (reinterpret_cast<uintptr_t>(ptr) + bytes) & ~mask_low;
		if (end <= begin)
			return; // the buffer owns no whole page
		unsigned long mask[1024 / (8 * sizeof(unsigned long))] = {};
		mask[static_cast<size_t>(node) / (8 * sizeof(unsigned long))] |= 1ul << (static_cast<size_t>(node) % (8 * sizeof(unsigned long)));
		constexpr int mpol_preferred = 1, mpol_mf_move = 2;
		(void)syscall(SYS_mbind, begin, end - begin, mpol_preferred, mask, 1024ul + 1ul, mpol_mf_move);
#else
		(void)ptr, (void)bytes, (void)node;
#endif
	}

	// The same for a frame whose row stripes are stored by DIFFERENT GPUs (direct-frame members, the ranks of a frame
	// group): every whole page goes to the node of the GPU that owns the stripe the page begins in.  On the two-socket
	// GPU hosts half of the GPUs hang off each socket; with the whole frame on one node the other half store across
	// the socket interconnect (+23..37 % kernel time for a GPU that stores a whole frame into far memory,
	// profiles/r03/shared_frame_numa.txt).  One move_pages(2) call with a target node per page: no memory policy is
	// installed and no mapping is split.  Pages have to be present to be moved, so they are touched first (the content
	// stays; nobody else writes the frame at this point: the caller is inside rt_hip_render and no launch has been
	// made); pages that are page-locked or mapped by another process stay where they are.  Best effort.
	void place_stripes(void* ptr, size_t bytes, uint32_t width, uint32_t height, uint32_t stripe_rows, const std::vector<int>& node_of_rank)
	{
#ifdef SYS_move_pages
		if (!ptr || !bytes || node_of_rank.empty() || !width || !height || !stripe_rows)
			return;
		if (const char* knob = std::getenv("RT_HIP_NUMA_MOVE"))
			if (knob[0] == '0' && knob[1] == '\0')
				return;
		bool any = false;
		for (const int node : node_of_rank)
			any = any || (node >= 0 && node < 1024);
		if (!any)
			return;
		const long page = sysconf(_SC_PAGESIZE);
		if (page <= 0)
			return;
		const uintptr_t mask_low = static_cast<uintptr_t>(page - 1);
		const uintptr_t base = reinterpret_cast<uintptr_t>(ptr);
		const uintptr_t begin = (base + mask_low) & ~mask_low;
		const uintptr_t end = (base + bytes) & ~mask_low;
		if (end <= begin)
			return;
		const size_t count = (end - begin) / static_cast<size_t>(page);
		std::vector<void*> pages;
		std::vector<int> nodes;
		pages.reserve(count);
		nodes.reserve(count);
		const size_t row_bytes = static_cast<size_t>(width) * sizeof(uint32_t);
		for (size_t i = 0; i < count; i++)
		{
			const uintptr_t address = begin + i * static_cast<size_t>(page);
			const size_t row = std::min<size_t>((address - base) / row_bytes, height - 1u);
			const int node = node_of_rank[(row / stripe_rows) % node_of_rank.size()];
			if (node < 0 || node >= 1024)
				continue;
			// present — and this process's own — from here on: the page's first byte is written back as it was read.  A read
			// alone settles for the kernel's shared zero page on memory that was allocated and never written, and an atomic
			// OR of nothing is turned into a fence by the compiler (no access at all).
			volatile unsigned char* const first_byte = reinterpret_cast<volatile unsigned char*>(address);
			*first_byte = *first_byte;
			pages.push_back(reinterpret_cast<void*>(address));
			nodes.push_back(node);
		}
		if (pages.empty())
			return;
		std::vector<int> status(pages.size(), 0);
		constexpr int mpol_mf_move = 2;
		const long rc = syscall(SYS_move_pages, 0, static_cast<unsigned long>(pages.size()), pages.data(), nodes.data(), status.data(), mpol_mf_move);
		if (debug_frame())
		{
			size_t arrived = 0, busy = 0, other = 0;
			for (size_t i = 0; i < pages.size(); i++)
				(status[i] == nodes[i] ? arrived : status[i] == -EBUSY ? busy : other)++;
			std::fprintf(stderr, "rt_hip: placed the stripes of back buffer %p: move_pages returned %ld (errno %d); %zu of %zu pages on their node, %zu busy, %zu other (first status %d)\n", ptr, rc, rc ? errno : 0, arrived, pages.size(), busy, other, status[0]);
		}
#else
		(void)ptr, (void)bytes, (void)width, (void)height, (void)stripe_rows, (void)node_of_rank;
#endif
	}

	// page-locks on callers' memory this process holds: + 1 per successful hipHostRegister, - 1 per unpin_frame of one
	static std::atomic<uint32_t> g_live_frame_locks{ 0 };
	uint32_t live_frame_locks()
	{
		return g_live_frame_locks.load(std::memory_order_relaxed);
	}

	// Drop the page-lock.  hipHostUnregister fails when the caller has already unmapped the buffer (the driver dropped
	// the registration with the mapping): either way the registration is gone afterwards, and the sticky error it may
	// leave behind is cleared so that the next HIP call of this thread does not report it.
	void unpin_frame(rt_hip_ctx* ctx)
	{
		if (ctx->pinned_frame)
		{
			const hipError_t e = hipHostUnregister(ctx->pinned_frame);
			(void)hipGetLastError();
			g_live_frame_locks.fetch_sub(1, std::memory_order_relaxed);
			if (debug_frame())
				std::fprintf(stderr, "rt_hip: device %d unregistered back buffer %p (%zu bytes): %s\n", ctx->device, ctx->pinned_frame, ctx->pinned_bytes, hipGetErrorString(e));
		}
		ctx->pinned_frame = nullptr;
		ctx->pinned_bytes = 0;
	}

	// image_view memory is ordinary pageable host memory (reference src/image.cpp:9-13).  With
	// RT_HIP_FLAG_PERSISTENT_FRAME it is page-locked on first sight and stays so while the same buffer keeps arriving;
	// any other buffer (or no flag) first drops the old registration — before anything else touches host memory.
	// A buffer whose page-lock was refused (registered by somebody else, not lockable) is remembered and not tried again
	// while it keeps arriving: neither the mbind nor the failing hipHostRegister is repeated every frame.
	// `stripe_nodes` (with the frame's shape): place the row stripes on their owners' nodes instead of the whole buffer on
	// this context's (place_stripes).
	void track_frame_buffer(rt_hip_ctx* ctx, uint32_t* pixels, size_t bytes, bool pin, bool may_move_pages, const std::vector<int>* stripe_nodes, uint32_t width, uint32_t height)
	{
		if (ctx->pinned_frame && (!pin || ctx->pinned_frame != pixels || ctx->pinned_bytes != bytes))
			unpin_frame(ctx);
		if (ctx->refused_frame && (!pin || ctx->refused_frame != pixels || ctx->refused_bytes != bytes))
		{
			ctx->refused_frame = nullptr;
			ctx->refused_bytes = 0;
		}
		if (pin && !ctx->pinned_frame && !ctx->refused_frame)
		{
			if (may_move_pages && stripe_nodes)
				place_stripes(pixels, bytes, width, height, RT_HIP_DEFAULT_STRIPE_ROWS, *stripe_nodes); // before the pages are locked where they are
			else if (may_move_pages)
				place_on_node(pixels, bytes, ctx->numa_node);
			const hipError_t e = hipHostRegister(pixels, bytes, hipHostRegisterMapped | (ctx->direct_frame ? hipHostRegisterPortable : 0u));
			if (debug_frame())
				std::fprintf(stderr, "rt_hip: device %d registered back buffer %p (%zu bytes): %s\n", ctx->device, static_cast<void*>(pixels), bytes, hipGetErrorString(e));
			if (e == hipSuccess)
			{
				ctx->pinned_frame = pixels;
				ctx->pinned_bytes = bytes;
				g_live_frame_locks.fetch_add(1, std::memory_order_relaxed);
			}
			else
			{
				(void)hipGetLastError(); // not fatal: the frame then takes the module's own page-locked frame and the carrier
				ctx->refused_frame = pixels;
				ctx->refused_bytes = bytes;
			}
		}
	}

	// ---- the module's own frame --------------------------------------------------------------------------------------------
	namespace
	{
		// CPUs' worth of time per period the process's cgroup grants it (v2: cpu.max, v1: cpu.cfs_quota_us / cpu.cfs_period_us);
		// 0 = no limit, or none that can be read
		double granted_cpu_time()
		{
			if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r"))
			{
				char quota[32] = {};
				double period = 0.0;
				const int got = std::fscanf(f, "%31s %lf", quota, &period);
				std::fclose(f);
				if (got == 2 && period > 0.0 && quota[0] >= '0' && quota[0] <= '9')
					return std::strtod(quota, nullptr) / period;
				return 0.0; // ("max": no limit)
			}
			double quota = 0.0, period = 0.0;
			if (FILE* f = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"))
			{
				if (std::fscanf(f, "%lf", &quota) != 1)
					quota = 0.0;
				std::fclose(f);
			}
			if (FILE* f = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r"))
			{
				if (std::fscanf(f, "%lf", &period) != 1)
					period = 0.0;
				std::fclose(f);
			}
			return quota > 0.0 && period > 0.0 ? quota / period : 0.0;
		}

		unsigned carrier_helpers(bool several_gpus)
		{
			// threads besides the caller's own.  They follow the kernel's progress and share what is left when the stream has
			// drained.  How many decides how SHORT a frame they can keep up with: 8.3 MB of pixels in the headline kernel's
			// 2.6 ms is 3 GB/s (one thread would do), in config 2's 0.70 ms it is 12 GB/s, and a thread carries 4-5 GB/s (a
			// line is read, checked, written and zeroed: 320 bytes of memory traffic per 64 bytes of pixels) — with three
			// helpers config 2's call took + 0.08 ms over the zero-copy mode, with seven it does not.  Several GPUs,
			// gathered: 7/8 of the frame arrives within the assemble kernel's 0.1-0.2 ms at the very end, at PCIe speed.
			// Helpers sleep between frames; while a frame is in flight they spin, as the reference's own renderer keeps
			// every core busy inside render() (src/renderers/mg_ray_tracer.cpp:203-204).
			(void)several_gpus;
			long wanted = 7;
			if (const char* knob = std::getenv("RT_HIP_COPY_THREADS"))
			{
				char* end = nullptr;
				const long v = std::strtol(knob, &end, 10);
				if (end != knob && v >= 0 && v <= 32)
					wanted = v;
			}
			// (the CPUs this PROCESS may use, not the machine's: a container's share can be a fraction of the host — by the set
			// of CPUs it may run on, or by the CPU TIME its cgroup grants: a GPU box of the pool this was measured on shows
			// all 256 hardware threads to a job that is granted the time of 16, cpu.max = "1600000 100000")
			unsigned cores = std::thread::hardware_concurrency();
			cpu_set_t allowed;
			CPU_ZERO(&allowed);
			if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0 && CPU_COUNT(&allowed) > 0)
				cores = static_cast<unsigned>(CPU_COUNT(&allowed));
			const double granted = granted_cpu_time();
			if (granted > 0.0 && granted < static_cast<double>(cores))
				cores = static_cast<unsigned>(std::max(1.0, std::ceil(granted)));
			if (cores && static_cast<unsigned>(wanted) + 2u > cores)
				wanted = static_cast<long>(cores) - 2; // one for the caller's thread, one for whatever else the host runs
			return static_cast<unsigned>(std::max(wanted, 0l));
		}
	}

	frame_delivery::frame_delivery(unsigned helpers, int node) : numa_node(node), carrier(helpers, node) {}

	hipError_t staging_frame::reserve(size_t wanted, int numa_node)
	{
		if (wanted <= bytes)
			return hipSuccess;
		release();
		// a little slack, so that a window dragged larger pixel row by pixel row does not re-allocate every frame
		const size_t rounded = (wanted + wanted / 8u + (2u << 20) - 1u) & ~static_cast<size_t>((2u << 20) - 1u);
		void* const mapping = mmap(nullptr, rounded, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
		if (mapping == MAP_FAILED)
			return hipErrorOutOfMemory;
#ifdef SYS_mbind
		if (numa_node >= 0 && numa_node < 1024 && !(std::getenv("RT_HIP_NUMA_MOVE") && std::getenv("RT_HIP_NUMA_MOVE")[0] == '0'))
		{
			unsigned long mask[1024 / (8 * sizeof(unsigned long))] = {};
			mask[static_cast<size_t>(numa_node) / (8 * sizeof(unsigned long))] |= 1ul << (static_cast<size_t>(numa_node) % (8 * sizeof(unsigned long)));
			constexpr int mpol_preferred = 1;
			(void)syscall(SYS_mbind, mapping, rounded, mpol_preferred, mask, 1024ul + 1ul, 0); // (nothing to move: no page exists yet)
		}
#endif
		std::memset(mapping, 0, rounded); // first touch: the pages come into being on that node
		const hipError_t e = hipHostRegister(mapping, rounded, hipHostRegisterMapped | hipHostRegisterPortable);
		if (e != hipSuccess)
		{
			(void)munmap(mapping, rounded);
			return e;
		}
		ptr = mapping;
		bytes = rounded;
		registered = true;
		if (debug_frame())
			std::fprintf(stderr, "rt_hip: the module's own frame: %zu bytes at %p on host node %d, page-locked and mapped\n", rounded, mapping, numa_node);
		return hipSuccess;
	}

	void staging_frame::release()
	{
		if (ptr)
		{
			if (registered)
				(void)hipHostUnregister(ptr);
			(void)hipGetLastError();
			(void)munmap(ptr, bytes);
		}
		ptr = nullptr;
		bytes = 0;
		registered = false;
		view = nullptr;
		view_device = -1;
	}

	rt_hip_status frame_delivery::begin(uint32_t* caller_pixels, size_t pixels, uint32_t** out_device_view, bool bottom_first)
	{
		const size_t bytes = pixels * sizeof(uint32_t);
		if (frame.bytes < bytes)
		{
			RT_HIP_TRY(frame.reserve(bytes, numa_node));
			dirty = false; // (a fresh frame is all zero)
		}
		if (dirty)
		{
			std::memset(frame.ptr, 0, frame.bytes);
			dirty = false;
		}
		int device = -1;
		RT_HIP_TRY(hipGetDevice(&device));
		if (!frame.view || frame.view_device != device) // (asked once per frame buffer and device, not once per frame)
		{
			frame.view = nullptr;
			RT_HIP_TRY(hipHostGetDevicePointer(&frame.view, frame.ptr, 0));
			frame.view_device = device;
		}
		*out_device_view = static_cast<uint32_t*>(frame.view);
		carrier.begin(frame.as<uint32_t>(), caller_pixels, pixels, bottom_first, false); // (announced by launched())
		return ok();
	}

	void frame_delivery::finish()
	{
		const auto t0 = std::chrono::steady_clock::now();
		carrier.finish();
		if (debug_frame())
			std::fprintf(stderr, "rt_hip: frame delivered: %zu of its 64 KB bands had been carried over before the stream drained; the rest took %.1f us\n", carrier.early_bands(), seconds_since(t0) * 1e6);
	}

	uint32_t* frame_delivery::view_on(int device)
	{
		void* view = nullptr;
		if (hipSetDevice(device) != hipSuccess || hipHostGetDevicePointer(&view, frame.ptr, 0) != hipSuccess)
		{
			(void)hipGetLastError();
			return nullptr;
		}
		return static_cast<uint32_t*>(view);
	}

	frame_delivery* delivery_of(rt_hip_ctx* ctx)
	{
		if (!ctx->delivery)
		{
			ctx->delivery.reset(new (std::nothrow) frame_delivery(carrier_helpers(ctx->multi && ctx->world > 1), ctx->numa_node));
			if (ctx->delivery && debug_frame())
				std::fprintf(stderr, "rt_hip: device %d: %u helper threads carry frames to the caller's buffer (the cgroup grants the time of %.1f CPUs; 0 = no limit)\n", ctx->device, ctx->delivery->carrier.helpers(), granted_cpu_time());
		}
		return ctx->delivery.get();
	}
}

extern "C" void rt_hip_forget_frame(rt_hip_ctx* ctx)
{
	if (!ctx)
		return;
	(void)hipSetDevice(ctx->device);
	if (ctx->stream)
		(void)hipStreamSynchronize(ctx->stream);
	unpin_frame(ctx);
	ctx->refused_frame = nullptr;
	ctx->refused_bytes = 0;
}
