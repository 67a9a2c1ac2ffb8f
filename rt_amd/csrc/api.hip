// rt_amd/csrc/api.hip — the C ABI of include/rt_hip.h on top of the gfx950 kernels.
//
// Host-side only: argument checking, HBM residency of the scene columns, launches, the device-to-host frame
// copy of the drop-in rt_hip_render(), error translation.  No exceptions leave this file; every HIP error is
// turned into RT_HIP_RUNTIME_ERROR + a message (the reference's render() is noexcept, src/renderer.hpp:11).
#include "../../include/rt_hip.h"
#include "contract.hpp"
#include "frame_group.hpp"
#include "kernels.hpp"

#include <rccl/rccl.h>

#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

using namespace rt_hip;

namespace
{
	thread_local std::string g_last_error;

	rt_hip_status fail(rt_hip_status status, const char* format, ...)
	{
		char buffer[512];
		va_list args;
		va_start(args, format);
		std::vsnprintf(buffer, sizeof(buffer), format, args);
		va_end(args);
		g_last_error = buffer;
		return status;
	}

	rt_hip_status ok()
	{
		return RT_HIP_OK;
	}

#define RT_HIP_TRY_NCCL(expr)                                                                                          \
	do                                                                                                                 \
	{                                                                                                                  \
		const ncclResult_t rt_hip_try_res = (expr);                                                                    \
		if (rt_hip_try_res != ncclSuccess)                                                                             \
			return fail(RT_HIP_RUNTIME_ERROR, "%s failed: %s", #expr, ncclGetErrorString(rt_hip_try_res));             \
	}                                                                                                                  \
	while (false)

#define RT_HIP_TRY(expr)                                                                                               \
	do                                                                                                                 \
	{                                                                                                                  \
		const hipError_t rt_hip_try_err = (expr);                                                                      \
		if (rt_hip_try_err != hipSuccess)                                                                              \
			return fail(RT_HIP_RUNTIME_ERROR, "%s failed: %s", #expr, hipGetErrorString(rt_hip_try_err));              \
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

	double seconds_since(std::chrono::steady_clock::time_point t0)
	{
		return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	}

	bool valid_partition(const rt_hip_partition& p)
	{
		return p.world && p.rank < p.world && p.stripe_rows;
	}

	uint32_t local_rows_of(uint32_t height, uint32_t rank, uint32_t world, uint32_t stripe_rows)
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
}

struct rt_hip_ctx
{
	int device = 0;
	uint32_t compute_units = 256;
	int numa_node = -1; // host NUMA node the GPU hangs off (sysfs), -1 = unknown

	// the scene, resident in HBM: one buffer holding every column back to back (256-byte aligned starts)
	device_buffer scene_columns;
	device_scene scene{};
	small_scene small{}, small_sm{}; // host copies of the kernel-argument scene of the `small` kernel (mg / sm scatter tables)
	bool have_scene = false;
	uint32_t samples_per_pixel = 0, max_bounces = 0;
	float inverse_view_projection[16]{};

	// big scenes: chunk sums on their way between waves and the pixels' arrival counters (kernels.hpp, rolling_buffers)
	device_buffer item_sums, pixel_done;
	device_buffer counters;
	device_counters* counters_host = nullptr; // page-locked; filled by an asynchronous copy right behind every launch that keeps stats
	hipEvent_t render_begin = nullptr, render_end = nullptr, counters_copied = nullptr;
	bool render_recorded = false; // the most recent launch was bracketed by the events above and its counters were read back
	bool launched = false;		  // something has been launched on last_stream
	hipStream_t last_stream = nullptr;
	// exchange phases of a multi-GPU frame on the root's stream (frames that keep stats): stripes on the root, frame
	// assembled, frame copied to the host
	hipEvent_t gathered = nullptr, assembled = nullptr, copied = nullptr;
	rt_hip_phases phases{};
	launch_cache cache; // what the launch code remembers per context (occupancy of the persistent kernels)

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
	device_buffer stripes_rgba, stripes_rgb;   // this member's compact stripe buffers (the gather's send side)
	device_buffer gathered_rgba, gathered_rgb; // root: n x padded stripes, rank order (the gather's receive side)
	hipEvent_t stripes_ready = nullptr;		   // recorded on `stream` after this member's launch (peer copies wait for it)
	// ---- or: one rank of a renderer whose ranks are processes that all map the caller's back buffer (rt_hip_join_frame_group)
	std::unique_ptr<frame_group> group;

	// staging for the drop-in rt_hip_render()
	device_buffer frame_rgba, frame_rgb;
	uint64_t scene_fingerprint = 0; // of the caller's columns last uploaded: rt has no scene version counter (src/main.cpp:233-311)
	size_t scene_bytes = 0;			// size of the resident block
	// the caller's frame buffer, page-locked and mapped into the GPU's address space while it keeps arriving at the same
	// address (the reference allocates its back buffer once per window size, src/window.cpp:61-64): the kernel renders
	// straight into it (one GPU, direct-frame members), or the assembled frame arrives by one DMA (gathered)
	void* pinned_frame = nullptr;
	size_t pinned_bytes = 0;
	void* refused_frame = nullptr; // a buffer whose page-lock failed: not tried again while it keeps arriving
	size_t refused_bytes = 0;

	const void* asked_pointer = nullptr; // rt_hip_render_device: the last output pointer and whether it is host memory
	bool asked_pointer_is_host = false;

	// KAT scratch
	device_buffer kat_in, kat_out;

	rt_hip_stats stats{};
};

namespace
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
	void place_on_node(void* ptr, size_t bytes, int node)
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

	// Drop the page-lock.  hipHostUnregister fails when the caller has already unmapped the buffer (the driver dropped
	// the registration with the mapping): either way the registration is gone afterwards, and the sticky error it may
	// leave behind is cleared so that the next HIP call of this thread does not report it.
	void unpin_frame(rt_hip_ctx* ctx)
	{
		if (ctx->pinned_frame)
		{
			const hipError_t e = hipHostUnregister(ctx->pinned_frame);
			(void)hipGetLastError();
			if (debug_frame())
				std::fprintf(stderr, "rt_hip: device %d unregistered back buffer %p (%zu bytes): %s\n", ctx->device, ctx->pinned_frame, ctx->pinned_bytes, hipGetErrorString(e));
		}
		ctx->pinned_frame = nullptr;
		ctx->pinned_bytes = 0;
	}
}

extern "C" uint32_t rt_hip_abi_version(void)
{
	return RT_HIP_ABI_VERSION;
}

extern "C" const char* rt_hip_last_error(void)
{
	return g_last_error.c_str();
}

extern "C" rt_hip_status rt_hip_device_count(int* count)
{
	if (!count)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_device_count: count is NULL");
	*count = 0;
	const hipError_t e = hipGetDeviceCount(count);
	if (e == hipErrorNoDevice)
	{
		*count = 0;
		return ok();
	}
	RT_HIP_TRY(e);
	return ok();
}

extern "C" rt_hip_status rt_hip_create(rt_hip_ctx** out_ctx, int device)
{
	if (!out_ctx)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_create: out_ctx is NULL");
	*out_ctx = nullptr;
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
		return fail(RT_HIP_NO_DEVICE, "rt_hip_create: no HIP device is visible");
	if (device < 0 || device >= count)
		return fail(RT_HIP_NO_DEVICE, "rt_hip_create: device %d out of range (%d visible)", device, count);
	RT_HIP_TRY(hipSetDevice(device));
	hipDeviceProp_t props{};
	RT_HIP_TRY(hipGetDeviceProperties(&props, device));
	if (std::strncmp(props.gcnArchName, "gfx950", 6) != 0)
		return fail(RT_HIP_NO_DEVICE, "rt_hip_create: device %d is %s; this module is built for gfx950 only", device, props.gcnArchName);

	rt_hip_ctx* ctx = new (std::nothrow) rt_hip_ctx;
	if (!ctx)
		return fail(RT_HIP_RUNTIME_ERROR, "rt_hip_create: out of host memory");
	ctx->device = device;
	ctx->compute_units = props.multiProcessorCount > 0 ? static_cast<uint32_t>(props.multiProcessorCount) : 256u;
	ctx->numa_node = numa_node_of(device);
	if (const char* knob = std::getenv("RT_HIP_NUMA_NODE")) // (tests and odd hosts: say which node the back buffer should live on)
	{
		char* end = nullptr;
		const long v = std::strtol(knob, &end, 10);
		if (end != knob && v >= -1 && v < 1024)
			ctx->numa_node = static_cast<int>(v);
	}
	hipError_t e = ctx->counters.reserve(sizeof(device_counters));
	if (e == hipSuccess)
		e = hipEventCreate(&ctx->render_begin);
	if (e == hipSuccess)
		e = hipEventCreate(&ctx->render_end);
	if (e == hipSuccess)
		e = hipEventCreateWithFlags(&ctx->stripes_ready, hipEventDisableTiming);
	if (e == hipSuccess)
		e = hipEventCreateWithFlags(&ctx->counters_copied, hipEventDisableTiming);
	if (e == hipSuccess)
		e = hipEventCreate(&ctx->gathered);
	if (e == hipSuccess)
		e = hipEventCreate(&ctx->assembled);
	if (e == hipSuccess)
		e = hipEventCreate(&ctx->copied);
	if (e == hipSuccess)
		e = hipHostMalloc(reinterpret_cast<void**>(&ctx->counters_host), sizeof(device_counters), hipHostMallocDefault);
	if (e == hipSuccess)
		e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
	if (e != hipSuccess)
	{
		rt_hip_destroy(ctx);
		return fail(RT_HIP_RUNTIME_ERROR, "rt_hip_create: %s", hipGetErrorString(e));
	}
	*out_ctx = ctx;
	return ok();
}

extern "C" rt_hip_status rt_hip_create_multi(rt_hip_ctx** out_ctx, const int* devices, int n_devices, uint32_t multi_flags)
{
	if (!out_ctx)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_create_multi: out_ctx is NULL");
	*out_ctx = nullptr;
	if (n_devices < 1 || n_devices > 64)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_create_multi: %d devices (1 .. 64 supported)", n_devices);
	if (multi_flags & ~static_cast<uint32_t>(RT_HIP_MULTI_PEER_COPY | RT_HIP_MULTI_DIRECT_FRAME))
		return fail(RT_HIP_UNSUPPORTED, "rt_hip_create_multi: unknown flag bits 0x%x", multi_flags);
	const bool peer_copy = (multi_flags & RT_HIP_MULTI_PEER_COPY) != 0;
	try
	{
		std::vector<int> ordinals(static_cast<size_t>(n_devices));
		for (int r = 0; r < n_devices; r++)
			ordinals[static_cast<size_t>(r)] = devices ? devices[r] : r;
		if (!peer_copy) // RCCL would fail later and less clearly
			for (int a = 0; a < n_devices; a++)
				for (int b = a + 1; b < n_devices; b++)
					if (ordinals[static_cast<size_t>(a)] == ordinals[static_cast<size_t>(b)])
						return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_create_multi: device %d is named twice (an RCCL communicator takes each device once; RT_HIP_MULTI_PEER_COPY allows it)", ordinals[static_cast<size_t>(a)]);

		rt_hip_ctx* created = nullptr;
		if (const rt_hip_status st = rt_hip_create(&created, ordinals[0]))
			return st;
		// owns the root (and through it the members pushed so far) until the very end: no exit below can leak it
		std::unique_ptr<rt_hip_ctx, void (*)(rt_hip_ctx*)> owner(created, rt_hip_destroy);
		rt_hip_ctx* const root = created;
		root->multi = true;
		root->peer_copy = peer_copy;
		root->direct_frame = (multi_flags & RT_HIP_MULTI_DIRECT_FRAME) != 0;
		root->world = static_cast<uint32_t>(n_devices);
		root->first_rank = 0;
		root->peers.reserve(static_cast<size_t>(n_devices));
		for (int r = 1; r < n_devices; r++)
		{
			rt_hip_ctx* member = nullptr;
			if (const rt_hip_status st = rt_hip_create(&member, ordinals[static_cast<size_t>(r)]))
				return st;
			root->peers.push_back(member); // (capacity reserved above: cannot throw)
		}
		if (peer_copy)
		{
			// the root pulls the stripes itself: it needs access to the other members' memory
			(void)hipSetDevice(root->device);
			for (const rt_hip_ctx* member : root->peers)
				if (member->device != root->device)
				{
					int can = 0;
					(void)hipDeviceCanAccessPeer(&can, root->device, member->device);
					if (can)
					{
						const hipError_t e = hipDeviceEnablePeerAccess(member->device, 0);
						if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
							return fail(RT_HIP_RUNTIME_ERROR, "rt_hip_create_multi: hipDeviceEnablePeerAccess(%d) failed: %s", member->device, hipGetErrorString(e));
						(void)hipGetLastError();
					}
					// without peer access hipMemcpyPeerAsync stages through the host: slower, still correct
				}
		}
		else
		{
			// one communicator per member of THIS process (rccl.h: ncclCommInitAll); rank r = member r = ordinals[r]
			root->comms.assign(static_cast<size_t>(n_devices), nullptr);
			const ncclResult_t res = ncclCommInitAll(root->comms.data(), n_devices, ordinals.data());
			if (res != ncclSuccess)
			{
				root->comms.clear();
				return fail(RT_HIP_RUNTIME_ERROR, "rt_hip_create_multi: ncclCommInitAll over %d device(s) failed: %s", n_devices, ncclGetErrorString(res));
			}
		}
		(void)hipSetDevice(root->device);
		*out_ctx = owner.release();
		return ok();
	}
	catch (const std::exception& e)
	{
		return fail(RT_HIP_RUNTIME_ERROR, "rt_hip_create_multi: %s", e.what());
	}
}

extern "C" rt_hip_status rt_hip_unique_id(char out_id[RT_HIP_UNIQUE_ID_BYTES])
{
	static_assert(sizeof(ncclUniqueId) == RT_HIP_UNIQUE_ID_BYTES, "rt_hip.h must match rccl.h's ncclUniqueId");
	if (!out_id)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_unique_id: NULL argument");
	ncclUniqueId id;
	RT_HIP_TRY_NCCL(ncclGetUniqueId(&id));
	std::memcpy(out_id, &id, sizeof(id));
	return ok();
}

namespace
{
	// ncclCommInitRank blocks until every rank has called it.  It runs on a helper thread so that the caller can give up
	// after a deadline: a rank that died between the launcher's vote and this call must not take the others with it.
	struct join_state
	{
		std::mutex mutex;
		std::condition_variable changed;
		bool done = false;
		bool abandoned = false; // the caller stopped waiting: the helper disposes of whatever it still gets
		ncclResult_t result = ncclSuccess;
		ncclComm_t comm = nullptr;
	};
}

extern "C" rt_hip_status rt_hip_join_ranks(rt_hip_ctx* ctx, int rank, int world, const char id[RT_HIP_UNIQUE_ID_BYTES], uint32_t timeout_ms)
{
	if (!ctx || !id)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_join_ranks: NULL argument");
	if (world < 1 || world > 4096 || rank < 0 || rank >= world)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_join_ranks: invalid rank %d of %d", rank, world);
	if (ctx->multi || ctx->group)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_join_ranks: the context already belongs to a multi-GPU renderer");
	if (!timeout_ms)
	{
		timeout_ms = 120000u;
		if (const char* knob = std::getenv("RT_HIP_JOIN_TIMEOUT_MS"))
		{
			const long v = std::strtol(knob, nullptr, 10);
			if (v > 0 && v < 0x7FFFFFFFl)
				timeout_ms = static_cast<uint32_t>(v);
		}
	}
	try
	{
		ncclUniqueId unique;
		std::memcpy(&unique, id, sizeof(unique));
		const auto state = std::make_shared<join_state>();
		const int device = ctx->device;
		std::thread(
			[state, unique, world, rank, device]
			{
				(void)hipSetDevice(device); // the current device is per host thread
				ncclComm_t comm = nullptr;
				// collective: returns when every rank of the renderer has called it (rccl.h: ncclCommInitRank)
				const ncclResult_t result = ncclCommInitRank(&comm, world, unique, rank);
				bool abandoned;
				{
					const std::lock_guard<std::mutex> lock(state->mutex);
					state->comm = comm;
					state->result = result;
					state->done = true;
					abandoned = state->abandoned;
				}
				state->changed.notify_all();
				if (abandoned && result == ncclSuccess && comm)
					(void)ncclCommAbort(comm);
			})
			.detach();
		std::unique_lock<std::mutex> lock(state->mutex);
		if (!state->changed.wait_for(lock, std::chrono::milliseconds(timeout_ms), [&] { return state->done; }))
		{
			state->abandoned = true;
			return fail(RT_HIP_TIMEOUT, "rt_hip_join_ranks: rank %d of %d waited %u ms in ncclCommInitRank for the other ranks", rank, world, timeout_ms);
		}
		if (state->result != ncclSuccess)
			return fail(RT_HIP_RUNTIME_ERROR, "rt_hip_join_ranks: ncclCommInitRank(rank %d of %d) failed: %s", rank, world, ncclGetErrorString(state->result));
		ctx->comms.assign(1, state->comm);
		ctx->multi = true;
		ctx->world = static_cast<uint32_t>(world);
		ctx->first_rank = static_cast<uint32_t>(rank);
		return ok();
	}
	catch (const std::exception& e)
	{
		return fail(RT_HIP_RUNTIME_ERROR, "rt_hip_join_ranks: %s", e.what());
	}
}

extern "C" rt_hip_status rt_hip_create_rank(rt_hip_ctx** out_ctx, int device, int rank, int world, const char id[RT_HIP_UNIQUE_ID_BYTES])
{
	if (!out_ctx)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_create_rank: out_ctx is NULL");
	*out_ctx = nullptr;
	if (!id || world < 1 || world > 4096 || rank < 0 || rank >= world)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_create_rank: invalid rank %d of %d", rank, world);
	rt_hip_ctx* ctx = nullptr;
	if (const rt_hip_status st = rt_hip_create(&ctx, device)) // the half that can fail on this rank alone
		return st;
	if (const rt_hip_status st = rt_hip_join_ranks(ctx, rank, world, id, 0)) // the collective half
	{
		rt_hip_destroy(ctx); // (keeps the message: rt_hip_destroy does not touch it)
		return st;
	}
	*out_ctx = ctx;
	return ok();
}

extern "C" rt_hip_status rt_hip_join_frame_group(rt_hip_ctx* ctx, int rank, int world, const char* name, uint32_t timeout_ms)
{
	if (!ctx || !name)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_join_frame_group: NULL argument");
	if (world < 1 || world > static_cast<int>(frame_group_max_ranks) || rank < 0 || rank >= world)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_join_frame_group: invalid rank %d of %d (at most %u ranks)", rank, world, frame_group_max_ranks);
	if (ctx->multi || ctx->group)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_join_frame_group: the context already belongs to a multi-GPU renderer");
	if (!timeout_ms)
	{
		timeout_ms = 120000u;
		if (const char* knob = std::getenv("RT_HIP_JOIN_TIMEOUT_MS"))
		{
			const long v = std::strtol(knob, nullptr, 10);
			if (v > 0 && v < 0x7FFFFFFFl)
				timeout_ms = static_cast<uint32_t>(v);
		}
	}
	try
	{
		auto group = std::make_unique<frame_group>();
		if (const char* knob = std::getenv("RT_HIP_GROUP_DEADLINE_MS"))
		{
			const long v = std::strtol(knob, nullptr, 10);
			if (v > 0 && v < 0x7FFFFFFFl)
				group->deadline_ms = static_cast<uint32_t>(v);
		}
		const frame_group::outcome joined = group->join(name, static_cast<uint32_t>(rank), static_cast<uint32_t>(world), timeout_ms);
		if (joined == frame_group::outcome::timed_out)
			return fail(RT_HIP_TIMEOUT, "rt_hip_join_frame_group: %s", group->error.c_str());
		if (joined != frame_group::outcome::ok)
			return fail(joined == frame_group::outcome::failed && !group->block ? RT_HIP_INVALID_ARGUMENT : RT_HIP_RUNTIME_ERROR, "rt_hip_join_frame_group: %s", group->error.c_str());
		group->block->ranks[rank].device = ctx->device;
		group->block->ranks[rank].numa_node = ctx->numa_node;
		ctx->group = std::move(group);
		ctx->world = static_cast<uint32_t>(world);
		ctx->first_rank = static_cast<uint32_t>(rank);
		return ok();
	}
	catch (const std::exception& e)
	{
		return fail(RT_HIP_RUNTIME_ERROR, "rt_hip_join_frame_group: %s", e.what());
	}
}

extern "C" rt_hip_status rt_hip_comm_info(const rt_hip_ctx* ctx, int member, int* out_ranks, int* out_rank, int* out_device, uint32_t* out_transport)
{
	if (!ctx || member < 0 || member > static_cast<int>(ctx->peers.size()))
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_comm_info: invalid argument");
	const rt_hip_ctx* const m = member == 0 ? ctx : ctx->peers[static_cast<size_t>(member - 1)];
	int ranks = static_cast<int>(ctx->world), rank = static_cast<int>(ctx->first_rank) + member, device = m->device;
	uint32_t transport = RT_HIP_TRANSPORT_NONE;
	if (ctx->multi)
		transport = ctx->peer_copy ? RT_HIP_TRANSPORT_PEER_COPY : RT_HIP_TRANSPORT_RCCL_GATHER;
	if (ctx->multi && ctx->phases.transport == RT_HIP_TRANSPORT_DIRECT_FRAME)
		transport = RT_HIP_TRANSPORT_DIRECT_FRAME;
	if (ctx->group)
		transport = RT_HIP_TRANSPORT_SHARED_FRAME;
	if (static_cast<size_t>(member) < ctx->comms.size() && ctx->comms[static_cast<size_t>(member)])
	{
		// what RCCL itself says about the communicator this member talks through
		const ncclComm_t comm = ctx->comms[static_cast<size_t>(member)];
		RT_HIP_TRY_NCCL(ncclCommCount(comm, &ranks));
		RT_HIP_TRY_NCCL(ncclCommUserRank(comm, &rank));
		RT_HIP_TRY_NCCL(ncclCommCuDevice(comm, &device));
	}
	if (out_ranks)
		*out_ranks = ranks;
	if (out_rank)
		*out_rank = rank;
	if (out_device)
		*out_device = device;
	if (out_transport)
		*out_transport = transport;
	return ok();
}

extern "C" rt_hip_status rt_hip_member_count(const rt_hip_ctx* ctx, int* out_count)
{
	if (!ctx || !out_count)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_member_count: NULL argument");
	*out_count = 1 + static_cast<int>(ctx->peers.size());
	return ok();
}

namespace
{
	rt_hip_ctx* member_of(rt_hip_ctx* ctx, int rank)
	{
		if (!ctx || rank < 0 || rank > static_cast<int>(ctx->peers.size()))
			return nullptr;
		return rank == 0 ? ctx : ctx->peers[static_cast<size_t>(rank - 1)];
	}
}

extern "C" rt_hip_status rt_hip_member_device(const rt_hip_ctx* ctx, int rank, int* out_device)
{
	if (ctx && ctx->group && out_device && rank >= 0 && rank < static_cast<int>(ctx->world))
	{
		*out_device = ctx->group->block->ranks[rank].device; // (the ordinal as that rank's process counts its devices)
		return ok();
	}
	const rt_hip_ctx* member = member_of(const_cast<rt_hip_ctx*>(ctx), rank);
	if (!member || !out_device)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_member_device: invalid argument");
	*out_device = member->device;
	return ok();
}

extern "C" void rt_hip_destroy(rt_hip_ctx* ctx)
{
	if (!ctx)
		return;
	ctx->group.reset(); // leaves the group: the other ranks stop waiting for this one at once
	// communicators first (they hold resources on every member's device), then the members, then the root
	for (const ncclComm_t comm : ctx->comms)
		if (comm)
			(void)ncclCommDestroy(comm);
	ctx->comms.clear();
	for (rt_hip_ctx* member : ctx->peers)
		rt_hip_destroy(member);
	ctx->peers.clear();
	(void)hipSetDevice(ctx->device);
	(void)hipDeviceSynchronize();
	unpin_frame(ctx);
	ctx->scene_columns.release();
	ctx->item_sums.release();
	ctx->pixel_done.release();
	ctx->counters.release();
	ctx->frame_rgba.release();
	ctx->frame_rgb.release();
	ctx->stripes_rgba.release();
	ctx->stripes_rgb.release();
	ctx->gathered_rgba.release();
	ctx->gathered_rgb.release();
	ctx->kat_in.release();
	ctx->kat_out.release();
	if (ctx->render_begin)
		(void)hipEventDestroy(ctx->render_begin);
	if (ctx->render_end)
		(void)hipEventDestroy(ctx->render_end);
	if (ctx->stripes_ready)
		(void)hipEventDestroy(ctx->stripes_ready);
	if (ctx->counters_copied)
		(void)hipEventDestroy(ctx->counters_copied);
	for (const hipEvent_t event : { ctx->gathered, ctx->assembled, ctx->copied })
		if (event)
			(void)hipEventDestroy(event);
	if (ctx->counters_host)
		(void)hipHostFree(ctx->counters_host);
	if (ctx->stream)
		(void)hipStreamDestroy(ctx->stream);
	delete ctx;
}

extern "C" rt_hip_status rt_hip_local_rows(uint32_t height, const rt_hip_partition* part, uint32_t* out_rows)
{
	if (!out_rows || !part || !valid_partition(*part))
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_local_rows: invalid partition");
	*out_rows = local_rows_of(height, part->rank, part->world, part->stripe_rows);
	return ok();
}

extern "C" rt_hip_status rt_hip_padded_local_rows(uint32_t height, const rt_hip_partition* part, uint32_t* out_rows)
{
	if (!out_rows || !part || !part->world || !part->stripe_rows)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_padded_local_rows: invalid partition");
	// rank 0 always owns the most rows: it gets the first stripe of every round
	*out_rows = local_rows_of(height, 0, part->world, part->stripe_rows);
	// ...except that a ragged LAST stripe may land on rank 0 while another rank holds a full one
	for (uint32_t r = 1; r < part->world; r++)
	{
		const uint32_t rows = local_rows_of(height, r, part->world, part->stripe_rows);
		if (rows > *out_rows)
			*out_rows = rows;
	}
	return ok();
}

namespace
{
	// the part of the scene check that is cheap enough for every call: counts against NULL pointers
	rt_hip_status check_scene_pointers(const rt_hip_scene& s)
	{
		if (!s.samples_per_pixel || !s.max_bounces)
			return fail(RT_HIP_INVALID_ARGUMENT, "scene: samples_per_pixel and max_bounces must be >= 1");
		if (s.n_spheres && (!s.sphere_center_x || !s.sphere_center_y || !s.sphere_center_z || !s.sphere_radius || !s.sphere_material))
			return fail(RT_HIP_INVALID_ARGUMENT, "scene: %u spheres but a sphere column is NULL", s.n_spheres);
		if (s.n_planes && (!s.plane_normal_x || !s.plane_normal_y || !s.plane_normal_z || !s.plane_d || !s.plane_material))
			return fail(RT_HIP_INVALID_ARGUMENT, "scene: %u planes but a plane column is NULL", s.n_planes);
		if (!s.n_materials && (s.n_spheres || s.n_planes || s.n_boxes))
			return fail(RT_HIP_INVALID_ARGUMENT, "scene: primitives present but no materials");
		if (s.n_materials && (!s.material_type || !s.material_albedo || !s.material_roughness || !s.material_reflectivity))
			return fail(RT_HIP_INVALID_ARGUMENT, "scene: %u materials but a material column is NULL", s.n_materials);
		if (s.n_boxes && (!s.box_center_x || !s.box_center_y || !s.box_center_z || !s.box_extents_x || !s.box_extents_y || !s.box_extents_z || !s.box_material))
			return fail(RT_HIP_INVALID_ARGUMENT, "scene: %u boxes but a box column is NULL", s.n_boxes);
		return ok();
	}

	// the part that walks the columns: needed only when their content is new (an unchanged fingerprint means these very
	// bytes passed before).  The reference's loader rejects out-of-range material indices (src/scene.cpp:568-574); an index
	// that got past it would read out of bounds on the device, so it is refused here too.
	rt_hip_status check_scene_indices(const rt_hip_scene& s)
	{
		for (uint32_t i = 0; i < s.n_spheres; i++)
			if (s.sphere_material[i] >= s.n_materials)
				return fail(RT_HIP_INVALID_ARGUMENT, "scene: sphere %u has material index %u out-of-range", i, s.sphere_material[i]);
		for (uint32_t i = 0; i < s.n_planes; i++)
			if (s.plane_material[i] >= s.n_materials)
				return fail(RT_HIP_INVALID_ARGUMENT, "scene: plane %u has material index %u out-of-range", i, s.plane_material[i]);
		for (uint32_t i = 0; i < s.n_boxes; i++)
			if (s.box_material[i] >= s.n_materials)
				return fail(RT_HIP_INVALID_ARGUMENT, "scene: box %u has material index %u out-of-range", i, s.box_material[i]);
		return ok();
	}

	constexpr size_t align_up(size_t v, size_t a)
	{
		return (v + a - 1) / a * a;
	}

	// A fingerprint of the caller's scene columns — FNV-1a style over 8-byte words, not a cryptographic hash.  rt has no
	// scene version counter (src/main.cpp:233-311), so every render() has to decide whether the columns in HBM are
	// still the caller's.  The columns are hashed WHERE THEY LIE, before anything is staged or copied: an unchanged scene
	// costs one pass over its bytes and nothing else.  Four interleaved lanes: one multiply per word is a dependency chain
	// of 3-4 cycles, and a single lane would spend 0.3 ms on the 2 MB of a 100 000-sphere scene.
	struct fingerprinter
	{
		static constexpr uint64_t prime = 0x100000001B3ull;
		uint64_t lane[4] = { 0xCBF29CE484222325ull, 0x9E3779B97F4A7C15ull, 0xBF58476D1CE4E5B9ull, 0x94D049BB133111EBull };

		void add(const void* data, size_t bytes)
		{
			const unsigned char* p = static_cast<const unsigned char*>(data);
			const size_t total = bytes;
			for (; bytes >= 32; bytes -= 32, p += 32)
			{
				uint64_t w[4];
				std::memcpy(w, p, 32);
				for (int k = 0; k < 4; k++)
					lane[k] = (lane[k] ^ w[k]) * prime;
			}
			for (; bytes >= 8; bytes -= 8, p += 8)
			{
				uint64_t w;
				std::memcpy(&w, p, 8);
				lane[0] = (lane[0] ^ w) * prime;
			}
			for (; bytes; bytes--, p++)
				lane[1] = (lane[1] ^ *p) * prime;
			lane[2] = (lane[2] ^ total) * prime; // the column's length: moving a row from one column to the next changes the print
		}
		void add_count(uint32_t n) { lane[3] = (lane[3] ^ n) * prime; }
		uint64_t value() const
		{
			uint64_t h = lane[0];
			for (int k = 1; k < 4; k++)
				h = (h ^ (lane[k] + (h << 6) + (h >> 2))) * prime;
			return h;
		}
	};

	uint64_t fingerprint_of(const rt_hip_scene& s)
	{
		fingerprinter f;
		f.add_count(s.n_spheres), f.add_count(s.n_planes), f.add_count(s.n_materials), f.add_count(s.n_boxes);
		const size_t sphere_bytes = static_cast<size_t>(s.n_spheres) * 4, plane_bytes = static_cast<size_t>(s.n_planes) * 4;
		const size_t material_bytes = static_cast<size_t>(s.n_materials) * 4, box_bytes = static_cast<size_t>(s.n_boxes) * 4;
		for (const void* column : { static_cast<const void*>(s.sphere_center_x), static_cast<const void*>(s.sphere_center_y), static_cast<const void*>(s.sphere_center_z),
									static_cast<const void*>(s.sphere_radius), static_cast<const void*>(s.sphere_material) })
			f.add(column, sphere_bytes);
		for (const void* column : { static_cast<const void*>(s.plane_normal_x), static_cast<const void*>(s.plane_normal_y), static_cast<const void*>(s.plane_normal_z),
									static_cast<const void*>(s.plane_d), static_cast<const void*>(s.plane_material) })
			f.add(column, plane_bytes);
		f.add(s.material_type, material_bytes);
		f.add(s.material_albedo, material_bytes * 4);
		f.add(s.material_roughness, material_bytes);
		f.add(s.material_reflectivity, material_bytes);
		for (const void* column : { static_cast<const void*>(s.box_center_x), static_cast<const void*>(s.box_center_y), static_cast<const void*>(s.box_center_z),
									static_cast<const void*>(s.box_extents_x), static_cast<const void*>(s.box_extents_y), static_cast<const void*>(s.box_extents_z),
									static_cast<const void*>(s.box_material) })
			f.add(column, box_bytes);
		return f.value();
	}

	// layout of the single HBM block: every column starts on a 256-byte boundary.  A function of the four counts alone.
	struct scene_layout
	{
		size_t scx, scy, scz, sr, sm;					 // sphere columns
		size_t pnx, pny, pnz, pd, pm;					 // plane columns
		size_t shading, type;							 // per material
		size_t geometry, prim_shading, prim_metal;		 // derived per-primitive tables (spheres, then planes)
		size_t prim_shading_sm, prim_scatter_sm;		 // the same under sm_ray_tracer's scatter table
		size_t box_bounds, albedo;						 // what only the preview reads
		size_t total;
	};

	scene_layout layout_of(const rt_hip_scene& s)
	{
		constexpr size_t column_alignment = 256;
		size_t offset = 0;
		const auto place = [&](size_t bytes)
		{
			const size_t at = offset;
			offset = align_up(offset + bytes, column_alignment);
			return at;
		};
		const size_t sphere_bytes = static_cast<size_t>(s.n_spheres) * 4;
		const size_t plane_bytes = static_cast<size_t>(s.n_planes) * 4;
		const size_t n_primitives = static_cast<size_t>(s.n_spheres) + s.n_planes;
		scene_layout L{};
		L.scx = place(sphere_bytes), L.scy = place(sphere_bytes), L.scz = place(sphere_bytes), L.sr = place(sphere_bytes), L.sm = place(sphere_bytes);
		L.pnx = place(plane_bytes), L.pny = place(plane_bytes), L.pnz = place(plane_bytes), L.pd = place(plane_bytes), L.pm = place(plane_bytes);
		L.shading = place(static_cast<size_t>(s.n_materials) * sizeof(float4));
		L.type = place(static_cast<size_t>(s.n_materials) * 4);
		L.geometry = place(n_primitives * sizeof(float4));
		L.prim_shading = place(n_primitives * sizeof(float4));
		L.prim_metal = place(n_primitives * 4);
		L.prim_shading_sm = place(n_primitives * sizeof(float4));
		L.prim_scatter_sm = place(n_primitives * 4);
		L.box_bounds = place(static_cast<size_t>(s.n_boxes) * 2 * sizeof(float4));
		L.albedo = place(static_cast<size_t>(s.n_materials) * sizeof(float4));
		L.total = offset ? offset : column_alignment;
		return L;
	}

	// One render()'s view of the caller's scene: pointers checked, fingerprint taken once — shared by all members of a
	// multi-GPU context — and the staging image of the HBM block, which is built only if some member turns out not to
	// hold these columns yet.
	struct scene_request
	{
		const rt_hip_scene* scene = nullptr;
		scene_layout layout{};
		uint64_t print = 0;
		bool indices_checked = false;
		bool image_built = false;
		std::vector<unsigned char> image; // host image of the block (one H2D copy per member that needs it)
		small_scene small{}, small_sm{};  // the kernel-argument scene of the `small` kernel (mg / sm scatter tables)
	};

	rt_hip_status open_request(scene_request& r, const rt_hip_scene* scene)
	{
		if (const rt_hip_status st = check_scene_pointers(*scene))
			return st;
		r.scene = scene;
		r.layout = layout_of(*scene);
		r.print = fingerprint_of(*scene);
		return ok();
	}

	void build_image(scene_request& r)
	{
		const rt_hip_scene& s = *r.scene;
		const scene_layout& L = r.layout;
		r.image.assign(L.total, 0);
		unsigned char* const host = r.image.data();
		const auto put = [&](size_t at, const void* src, size_t bytes)
		{
			if (bytes)
				std::memcpy(host + at, src, bytes);
		};
		const size_t sphere_bytes = static_cast<size_t>(s.n_spheres) * 4;
		const size_t plane_bytes = static_cast<size_t>(s.n_planes) * 4;
		const size_t n_primitives = static_cast<size_t>(s.n_spheres) + s.n_planes;
		put(L.scx, s.sphere_center_x, sphere_bytes);
		put(L.scy, s.sphere_center_y, sphere_bytes);
		put(L.scz, s.sphere_center_z, sphere_bytes);
		put(L.sr, s.sphere_radius, sphere_bytes);
		put(L.sm, s.sphere_material, sphere_bytes);
		put(L.pnx, s.plane_normal_x, plane_bytes);
		put(L.pny, s.plane_normal_y, plane_bytes);
		put(L.pnz, s.plane_normal_z, plane_bytes);
		put(L.pd, s.plane_d, plane_bytes);
		put(L.pm, s.plane_material, plane_bytes);
		for (uint32_t m = 0; m < s.n_materials; m++)
		{
			// attenuation = vec3{ albedo * reflectivity } (mg_ray_tracer.cpp:115,131; colour * float, colour.hpp:144-149)
			const float refl = s.material_reflectivity[m];
			const float shading[4] = { s.material_albedo[m * 4 + 0] * refl,
									   s.material_albedo[m * 4 + 1] * refl,
									   s.material_albedo[m * 4 + 2] * refl,
									   s.material_roughness[m] };
			put(L.shading + m * sizeof(float4), shading, sizeof(shading));
		}
		put(L.type, s.material_type, static_cast<size_t>(s.n_materials) * 4);
		put(L.albedo, s.material_albedo, static_cast<size_t>(s.n_materials) * sizeof(float4));
		for (uint32_t i = 0; i < s.n_boxes; i++)
		{
			// corners = center -/+ extents (muu::bounding_box), material index riding in the spare lane
			float bounds[8] = { s.box_center_x[i] - s.box_extents_x[i], s.box_center_y[i] - s.box_extents_y[i], s.box_center_z[i] - s.box_extents_z[i], 0.0f,
								s.box_center_x[i] + s.box_extents_x[i], s.box_center_y[i] + s.box_extents_y[i], s.box_center_z[i] + s.box_extents_z[i], 0.0f };
			std::memcpy(&bounds[3], &s.box_material[i], 4);
			put(L.box_bounds + i * 2 * sizeof(float4), bounds, sizeof(bounds));
		}
		// derived per-primitive tables (spheres, then planes)
		r.small = small_scene{};
		r.small_sm = small_scene{};
		for (size_t i = 0; i < n_primitives; i++)
		{
			const bool is_sphere = i < s.n_spheres;
			const size_t k = is_sphere ? i : i - s.n_spheres;
			float geometry[4];
			uint32_t material;
			if (is_sphere)
			{
				const float radius = s.sphere_radius[k];
				geometry[0] = s.sphere_center_x[k], geometry[1] = s.sphere_center_y[k], geometry[2] = s.sphere_center_z[k];
				geometry[3] = radius * radius; // radius^2, as hits_sphere squares it
				material = s.sphere_material[k];
			}
			else
			{
				geometry[0] = s.plane_normal_x[k], geometry[1] = s.plane_normal_y[k], geometry[2] = s.plane_normal_z[k];
				geometry[3] = s.plane_d[k];
				material = s.plane_material[k];
			}
			const uint32_t type = s.material_type[material];
			const uint32_t metal = type == RT_HIP_MATERIAL_METAL ? scatter_metal : scatter_lambert; // mg_ray_tracer.cpp:142-152
			// sm_ray_tracer.cpp:221-236
			const bool refracts = type == RT_HIP_MATERIAL_DIELECTRIC || type == RT_HIP_MATERIAL_AIR || type == RT_HIP_MATERIAL_VACUUM
							   || type == RT_HIP_MATERIAL_WATER || type == RT_HIP_MATERIAL_ICE;
			const uint32_t scatter_sm = refracts ? scatter_dielectric : metal;
			float shading_mg[4], shading_sm[4];
			std::memcpy(shading_mg, host + L.shading + material * sizeof(float4), sizeof(float4));
			std::memcpy(shading_sm, shading_mg, sizeof(float4));
			if (refracts)
				shading_sm[3] = s.material_reflectivity[material]; // index of refraction instead of the (unused) roughness
			put(L.geometry + i * sizeof(float4), geometry, sizeof(geometry));
			put(L.prim_shading + i * sizeof(float4), shading_mg, sizeof(float4));
			put(L.prim_metal + i * 4, &metal, 4);
			put(L.prim_shading_sm + i * sizeof(float4), shading_sm, sizeof(float4));
			put(L.prim_scatter_sm + i * 4, &scatter_sm, 4);
			if (is_sphere && i < scalar_max_spheres)
			{
				std::memcpy(&r.small.geometry[i], geometry, sizeof(geometry));
				std::memcpy(&r.small.shading[i], shading_mg, sizeof(float4));
				r.small.scatter[i] = metal;
				std::memcpy(&r.small_sm.geometry[i], geometry, sizeof(geometry));
				std::memcpy(&r.small_sm.shading[i], shading_sm, sizeof(float4));
				r.small_sm.scatter[i] = scatter_sm;
			}
		}
		r.image_built = true;
	}

	// Make `ctx` hold the request's scene: nothing but the frame's scalars if the columns' fingerprint is the resident
	// one (the reference re-renders an unchanged scene every dirty frame, src/main.cpp:315-321), the full path otherwise.
	// Leaves ctx->device current.
	rt_hip_status make_resident(rt_hip_ctx* ctx, scene_request& r)
	{
		const auto t0 = std::chrono::steady_clock::now();
		const rt_hip_scene& s = *r.scene;
		const scene_layout& L = r.layout;
		RT_HIP_TRY(hipSetDevice(ctx->device));
		const bool resident = ctx->have_scene && ctx->scene_fingerprint == r.print && ctx->scene_bytes == L.total && ctx->scene_columns.bytes >= L.total;
		if (!resident)
		{
			if (!r.indices_checked)
			{
				if (const rt_hip_status st = check_scene_indices(s))
					return st;
				r.indices_checked = true;
			}
			if (!r.image_built)
				build_image(r);
			RT_HIP_TRY(hipDeviceSynchronize()); // a previous frame may still be reading the old scene
			ctx->have_scene = false;
			RT_HIP_TRY(ctx->scene_columns.reserve(L.total));
			RT_HIP_TRY(hipMemcpy(ctx->scene_columns.ptr, r.image.data(), L.total, hipMemcpyHostToDevice));
			ctx->scene_fingerprint = r.print;
			ctx->scene_bytes = L.total;
			ctx->small = r.small;
			ctx->small_sm = r.small_sm;

			unsigned char* base = ctx->scene_columns.as<unsigned char>();
			device_scene& d = ctx->scene;
			d.n_spheres = s.n_spheres;
			d.n_planes = s.n_planes;
			d.n_materials = s.n_materials;
			d.sphere_cx = reinterpret_cast<const float*>(base + L.scx);
			d.sphere_cy = reinterpret_cast<const float*>(base + L.scy);
			d.sphere_cz = reinterpret_cast<const float*>(base + L.scz);
			d.sphere_r = reinterpret_cast<const float*>(base + L.sr);
			d.sphere_material = reinterpret_cast<const uint32_t*>(base + L.sm);
			d.plane_nx = reinterpret_cast<const float*>(base + L.pnx);
			d.plane_ny = reinterpret_cast<const float*>(base + L.pny);
			d.plane_nz = reinterpret_cast<const float*>(base + L.pnz);
			d.plane_d = reinterpret_cast<const float*>(base + L.pd);
			d.plane_material = reinterpret_cast<const uint32_t*>(base + L.pm);
			d.material_shading = reinterpret_cast<const float4*>(base + L.shading);
			d.material_type = reinterpret_cast<const uint32_t*>(base + L.type);
			d.primitive_geometry = reinterpret_cast<const float4*>(base + L.geometry);
			d.primitive_shading = reinterpret_cast<const float4*>(base + L.prim_shading);
			d.primitive_scatter = reinterpret_cast<const uint32_t*>(base + L.prim_metal);
			d.primitive_shading_sm = reinterpret_cast<const float4*>(base + L.prim_shading_sm);
			d.primitive_scatter_sm = reinterpret_cast<const uint32_t*>(base + L.prim_scatter_sm);
			d.n_boxes = s.n_boxes;
			d.box_bounds = reinterpret_cast<const float4*>(base + L.box_bounds);
			d.material_albedo = reinterpret_cast<const float4*>(base + L.albedo);
		}
		ctx->samples_per_pixel = s.samples_per_pixel;
		ctx->max_bounces = s.max_bounces;
		std::memcpy(ctx->inverse_view_projection, s.inverse_view_projection, sizeof(ctx->inverse_view_projection));
		ctx->have_scene = true;
		ctx->phases.scene_resident = resident ? 1u : 0u;
		ctx->stats.upload_ms = static_cast<float>(seconds_since(t0) * 1e3);
		return ok();
	}
}

extern "C" rt_hip_status rt_hip_scene_check(const rt_hip_scene* scene, uint64_t* out_fingerprint)
{
	if (!scene)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_scene_check: NULL argument");
	if (const rt_hip_status st = check_scene_pointers(*scene))
		return st;
	if (const rt_hip_status st = check_scene_indices(*scene))
		return st;
	if (out_fingerprint)
		*out_fingerprint = fingerprint_of(*scene);
	return ok();
}

extern "C" rt_hip_status rt_hip_scene_upload(rt_hip_ctx* ctx, const rt_hip_scene* scene)
{
	if (!ctx || !scene)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_scene_upload: NULL argument");
	try // nothing may propagate through the C boundary (the staging image allocates)
	{
		const auto t0 = std::chrono::steady_clock::now();
		scene_request request;
		if (const rt_hip_status st = open_request(request, scene))
			return st;
		if (const rt_hip_status st = make_resident(ctx, request))
			return st;
		ctx->stats.upload_ms = static_cast<float>(seconds_since(t0) * 1e3); // including the fingerprint pass
		return ok();
	}
	catch (const std::exception& e)
	{
		return fail(RT_HIP_RUNTIME_ERROR, "rt_hip_scene_upload: %s", e.what());
	}
	catch (...)
	{
		return fail(RT_HIP_RUNTIME_ERROR, "rt_hip_scene_upload: unknown exception");
	}
}

namespace
{
	rt_hip_status render_device(rt_hip_ctx* ctx, uint32_t width, uint32_t height, uint64_t seed, uint32_t flags, const rt_hip_partition* part, uint32_t* d_rgba8, float* d_rgb_f32, void* stream, bool whole_frame_buffers, bool keep_stats, bool host_frame);

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

namespace
{
// whole_frame_buffers: d_rgba8 / d_rgb_f32 are the whole width x height frame and every pixel goes to its image row
// (several GPUs rendering into one host frame); otherwise the rank's compact stripe buffer, as the public call documents.
// keep_stats: bracket the launch with timing events, zero the work counters before it and read them back after it.  Without
// it NOTHING but the kernel is enqueued (the plug-in's call: rt_hip_render with stats == NULL).
// host_frame: d_rgba8 is page-locked host memory (the mapped back buffer): the tiles are cut for PCIe writes (choose_queue).
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
	f.max_bounces = ctx->max_bounces;
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
	// w = fma(mx[3], ndc.x, fma(my[3], ndc.y, k[3])) is exactly k[3] for every finite ndc when mx[3] and my[3] are
	// (+-)0 and k[3] is not: then the per-sample 1/w is this one constant
	f.uniform_w = (f.mx[3] == 0.0f && f.my[3] == 0.0f && f.k_near[3] != 0.0f && f.k_far[3] != 0.0f && std::isfinite(f.k_near[3]) && std::isfinite(f.k_far[3])) ? 1u : 0u;
	f.inv_w_near = f.uniform_w ? 1.0f / f.k_near[3] : 0.0f;
	f.inv_w_far = f.uniform_w ? 1.0f / f.k_far[3] : 0.0f;
	if (f.uniform_w)
	{
		// Contract v3, item 3 (oracle/cpu_ref.cpp make_frame has the same lines): with w constant over the frame,
		//   near(px, py) = (mx X + my Y + k_near) / w_near,   X = (2/W) px - 1,   Y = -(2/H) py + 1,
		// is affine in the pixel position, and so is far - near.  The constants are worked out in binary64, in THIS
		// order of operations, and rounded to binary32 once; the kernels evaluate two fmas per component.
		const double sx = 2.0 / static_cast<double>(width), sy = -(2.0 / static_cast<double>(height));
		const double iwn = 1.0 / static_cast<double>(f.k_near[3]), iwf = 1.0 / static_cast<double>(f.k_far[3]);
		for (int c = 0; c < 3; c++)
		{
			const double mx = f.mx[c], my = f.my[c], kn = f.k_near[c], kf = f.k_far[c];
			const double o1 = mx * sx * iwn, o2 = my * sy * iwn, o0 = (kn - mx + my) * iwn;
			const double e1 = mx * sx * iwf, e2 = my * sy * iwf, e0 = (kf - mx + my) * iwf;
			f.ray_o0[c] = static_cast<float>(o0), f.ray_o1[c] = static_cast<float>(o1), f.ray_o2[c] = static_cast<float>(o2);
			f.ray_d0[c] = static_cast<float>(e0 - o0), f.ray_d1[c] = static_cast<float>(e1 - o1), f.ray_d2[c] = static_cast<float>(e2 - o2);
		}
	}

	bool rolling_items = false; // the persistent big-scene kernels draw items from a sequence whose head must start at 0
	rolling_buffers rolling;
	if (!(flags & RT_HIP_FLAG_PREVIEW))
	{
		const uint32_t variant = choose_kernel(ctx->scene, flags, f.samples_per_pixel, f.uniform_w != 0);
		const bool big_scene = variant == RT_HIP_KERNEL_TILED || variant == RT_HIP_KERNEL_STREAMED;
		rolling_items = big_scene;
		const queue_params queue = choose_queue(f.samples_per_pixel, width, f.local_rows, big_scene, host_frame, half_chunk_choice(flags), ctx->scene.n_spheres + ctx->scene.n_planes, variant == RT_HIP_KERNEL_STREAMED && ctx->scene.n_spheres >= sparse_launch_min_spheres);
		// small scenes: a pixel's chunk sums (one per 16 samples) are parked in LDS until the pixel is complete
		const uint64_t slot_bytes = big_scene ? 0u : 4ull * tile_slot_bytes(queue);
		if (slot_bytes > 48u * 1024u)
			return fail(RT_HIP_UNSUPPORTED, "rt_hip_render_device: %u samples per pixel are more than the kernels hold chunk sums for (4096; the reference clamps to 1000, src/scene.cpp:544)", f.samples_per_pixel);
		// big scenes: they meet in HBM, 16 bytes per chunk of this rank's rows
		size_t sums_bytes = 0, done_bytes = 0;
		rolling_buffer_bytes(queue, f.samples_per_pixel, width, f.local_rows, big_scene, sums_bytes, done_bytes);
		if (sums_bytes > (64ull << 30))
			return fail(RT_HIP_UNSUPPORTED, "rt_hip_render_device: %ux%u at %u samples per pixel needs %zu GiB for the chunk sums of a scene of this size", width, height, f.samples_per_pixel, sums_bytes >> 30);
		if (sums_bytes)
		{
			RT_HIP_TRY(ctx->item_sums.reserve(sums_bytes));
			if (ctx->pixel_done.bytes < done_bytes)
			{
				// the counters are zero between launches (the lane that folds a pixel puts its counter back): new memory is zeroed once
				RT_HIP_TRY(ctx->pixel_done.reserve(done_bytes));
				RT_HIP_TRY(hipMemsetAsync(ctx->pixel_done.ptr, 0, ctx->pixel_done.bytes, s));
			}
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

namespace
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

namespace
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

namespace
{
	// (RT_HIP_FLAG_PERSISTENT_FRAME and RT_HIP_FLAG_STATS are rt_hip_render's own: they do not travel to the launch)
	constexpr uint32_t render_flag_mask = RT_HIP_FLAG_FORCE_TILED | RT_HIP_FLAG_FORCE_RESIDENT | RT_HIP_FLAG_SM_MATERIALS | RT_HIP_FLAG_PREVIEW | RT_HIP_FLAG_FORCE_STREAMED | RT_HIP_FLAG_FAST | RT_HIP_FLAG_FORCE_HALF_CHUNKS | RT_HIP_FLAG_FORCE_WHOLE_CHUNKS;

	// image_view memory is ordinary pageable host memory (reference src/image.cpp:9-13).  With
	// RT_HIP_FLAG_PERSISTENT_FRAME it is page-locked on first sight and stays so while the same buffer keeps arriving;
	// any other buffer (or no flag) first drops the old registration — before anything else touches host memory.
	// A buffer whose page-lock was refused (registered by somebody else, not lockable) is remembered and not tried again
	// while it keeps arriving: neither the mbind nor the failing hipHostRegister is repeated every frame.
	// `stripe_nodes` (with the frame's shape): place the row stripes on their owners' nodes instead of the whole buffer on
	// this context's (place_stripes).
	void track_frame_buffer(rt_hip_ctx* ctx, uint32_t* pixels, size_t bytes, bool pin, bool may_move_pages = true, const std::vector<int>* stripe_nodes = nullptr, uint32_t width = 0, uint32_t height = 0)
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
			}
			else
			{
				(void)hipGetLastError(); // not fatal: the frame then goes through HBM and the driver's bounce buffers
				ctx->refused_frame = pixels;
				ctx->refused_bytes = bytes;
			}
		}
	}

	// After the first launch of a multi-GPU frame nothing may return before every member's stream has drained: a member
	// that is still storing into the caller's back buffer (or into stripe buffers a later call would re-use) must not
	// outlive the call that reported the failure.  Also puts the root's device back as the current one.
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

	float elapsed_or_zero(hipEvent_t from, hipEvent_t to)
	{
		float ms = 0.0f;
		if (hipEventElapsedTime(&ms, from, to) != hipSuccess)
		{
			(void)hipGetLastError();
			ms = 0.0f;
		}
		return ms;
	}

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
		settle_members settle(root);
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

		// the caller's page-locked back buffer as the root's GPU sees it (NULL: not page-locked, or this rank has no frame)
		uint32_t* mapped_frame = nullptr;
		if (have_root && root->pinned_frame == pixels_rgba8888)
		{
			RT_HIP_TRY(hipSetDevice(root->device));
			void* view = nullptr;
			if (hipHostGetDevicePointer(&view, pixels_rgba8888, 0) == hipSuccess && view)
				mapped_frame = static_cast<uint32_t*>(view);
			else
				(void)hipGetLastError();
		}

		// RT_HIP_MULTI_DIRECT_FRAME: no gather at all.  The caller's back buffer is page-locked and mapped into every member's
		// address space; each member's kernel stores its pixels straight into their image rows (system-scope stores over
		// that GPU's own PCIe link), and the call is over when the last member's launch is.  Needs the page-locked buffer
		// (RT_HIP_FLAG_PERSISTENT_FRAME) and all ranks in this process; the float mean still goes the gathered way.
		if (root->direct_frame && mapped_frame && n == static_cast<int>(world) && !rgb_f32)
		{
			bool mapped_everywhere = true;
			std::vector<uint32_t*> views(static_cast<size_t>(n), nullptr);
			views[0] = mapped_frame;
			for (int r = 1; r < n && mapped_everywhere; r++)
			{
				RT_HIP_TRY(hipSetDevice(member_of(root, r)->device));
				void* view = nullptr;
				if (hipHostGetDevicePointer(&view, pixels_rgba8888, 0) == hipSuccess && view)
					views[static_cast<size_t>(r)] = static_cast<uint32_t*>(view);
				else
				{
					(void)hipGetLastError();
					mapped_everywhere = false;
				}
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
				root->stats.readback_ms = keep_stats ? static_cast<float>(seconds_since(t0) * 1e3) : 0.0f;
				settle.armed = false;
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
			if (!mapped_frame)
				RT_HIP_TRY(root->frame_rgba.reserve(pixels * sizeof(uint32_t)));
			if (rgb_f32)
				RT_HIP_TRY(root->frame_rgb.reserve(pixels * 3 * sizeof(float)));
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

		// 4. rank 0: de-interleave the other ranks' stripes into the frame.  With a page-locked back buffer the assemble
		//    kernel stores them straight into the caller's frame (system-scope stores: the pixels cross PCIe while the
		//    kernel runs) — no frame in HBM, no copy; otherwise into HBM, followed by one copy.
		if (keep_stats)
			RT_HIP_TRY(hipEventRecord(root->gathered, root->stream));
		uint32_t* const assembled_rgba = mapped_frame ? mapped_frame : root->frame_rgba.as<uint32_t>();
		if (!(root_direct && world == 1u)) // (a world of one rendered everything in place)
			launch_assemble(width, height, world, RT_HIP_DEFAULT_STRIPE_ROWS, padded_rows, root->gathered_rgba.as<uint32_t>(), assembled_rgba, root_direct ? 1u : 0u, mapped_frame != nullptr, root->stream);
		RT_HIP_TRY(hipGetLastError());
		if (rgb_f32)
		{
			launch_assemble(width * 3u, height, world, RT_HIP_DEFAULT_STRIPE_ROWS, padded_rows, root->gathered_rgb.as<uint32_t>(), root->frame_rgb.as<uint32_t>(), 0u, false, root->stream);
			RT_HIP_TRY(hipGetLastError());
		}
		if (keep_stats)
			RT_HIP_TRY(hipEventRecord(root->assembled, root->stream));
		if (!mapped_frame)
			RT_HIP_TRY(hipMemcpyAsync(pixels_rgba8888, root->frame_rgba.ptr, pixels * sizeof(uint32_t), hipMemcpyDeviceToHost, root->stream));
		if (rgb_f32)
			RT_HIP_TRY(hipMemcpyAsync(rgb_f32, root->frame_rgb.ptr, pixels * 3 * sizeof(float), hipMemcpyDeviceToHost, root->stream));
		if (keep_stats)
			RT_HIP_TRY(hipEventRecord(root->copied, root->stream));
		const auto issued = std::chrono::steady_clock::now();
		if (keep_stats)
			RT_HIP_TRY(hipEventSynchronize(root->render_end)); // (the root's own kernel: where the read-back clock starts)
		const auto t0 = std::chrono::steady_clock::now();
		RT_HIP_TRY(hipStreamSynchronize(root->stream));
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

namespace
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
			group.break_group("rank %u: %s", rank, g_last_error.c_str());
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
		// A page-locked back buffer is mapped into the device's address space: the kernel stores every finished pixel
		// straight into it (4 bytes per pixel over PCIe while the rest of the frame is still being traced), and
		// there is no read-back step at all.  Otherwise the frame is rendered into HBM and copied.
		uint32_t* d_frame = nullptr;
		bool mapped = false;
		if (ctx->pinned_frame)
		{
			void* device_view = nullptr;
			if (hipHostGetDevicePointer(&device_view, pixels_rgba8888, 0) == hipSuccess && device_view)
			{
				d_frame = static_cast<uint32_t*>(device_view);
				mapped = true;
			}
			else
				(void)hipGetLastError();
		}
		if (!mapped)
		{
			RT_HIP_TRY(ctx->frame_rgba.reserve(frame_bytes));
			d_frame = ctx->frame_rgba.as<uint32_t>();
		}
		if (rgb_f32)
			RT_HIP_TRY(ctx->frame_rgb.reserve(pixels * 3 * sizeof(float)));
		if (const rt_hip_status st = render_device(ctx, width, height, seed, flags & render_flag_mask, nullptr, d_frame, rgb_f32 ? ctx->frame_rgb.as<float>() : nullptr, ctx->stream, false, keep_stats, mapped))
			return st;
		// from here on the device may be storing into the caller's buffers: no return before the stream has drained
		hipError_t e = hipSuccess;
		if (!mapped)
			e = hipMemcpyAsync(pixels_rgba8888, d_frame, frame_bytes, hipMemcpyDeviceToHost, ctx->stream);
		if (e == hipSuccess && rgb_f32)
			e = hipMemcpyAsync(rgb_f32, ctx->frame_rgb.ptr, pixels * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
		const auto issued = std::chrono::steady_clock::now();
		if (e == hipSuccess && keep_stats)
			e = hipEventSynchronize(ctx->render_end);
		const auto t0 = std::chrono::steady_clock::now();
		const hipError_t drained = hipStreamSynchronize(ctx->stream);
		RT_HIP_TRY(e);
		RT_HIP_TRY(drained);
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

#ifdef RT_HIP_REGION_COUNTERS
// experiment variant only: out[0..12] = runs, out[13..25] = lanes of the most recent launch on this context
extern "C" rt_hip_status rt_hip_debug_region_counters(rt_hip_ctx* ctx, uint64_t* out)
{
	if (!ctx || !out)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_debug_region_counters: NULL argument");
	RT_HIP_TRY(hipSetDevice(ctx->device));
	RT_HIP_TRY(hipEventSynchronize(ctx->counters_copied));
	for (unsigned i = 0; i < device_counters::regions; i++)
	{
		out[i] = ctx->counters_host->region_runs[i];
		out[device_counters::regions + i] = ctx->counters_host->region_lanes[i];
	}
	return ok();
}
#endif

#ifdef RT_HIP_WAVE_CLOCKS
// experiment variant only: out[3 * w + {0, 1, 2}] = start / queue-dry / end tick (100 MHz) of wave w of the most recent
// persistent launch on this context, for w < *count (in: capacity of `out` in waves; out: waves the build records)
extern "C" rt_hip_status rt_hip_debug_wave_clocks(rt_hip_ctx* ctx, uint64_t* out, uint32_t* count)
{
	if (!ctx || !out || !count)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_debug_wave_clocks: NULL argument");
	RT_HIP_TRY(hipSetDevice(ctx->device));
	RT_HIP_TRY(hipEventSynchronize(ctx->counters_copied));
	const uint32_t n = std::min<uint32_t>(*count, device_counters::clocked_waves);
	for (uint32_t w = 0; w < n; w++)
		for (int k = 0; k < 3; k++)
			out[3 * w + k] = ctx->counters_host->wave_clocks[w][k];
	*count = n;
	return ok();
}
#endif

// ---- known-answer entry points --------------------------------------------------------------------------------------

extern "C" rt_hip_status rt_hip_kat_random(rt_hip_ctx* ctx, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, float* out)
{
	if (!ctx || !out || !n)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_kat_random: invalid argument");
	RT_HIP_TRY(hipSetDevice(ctx->device));
	RT_HIP_TRY(ctx->kat_out.reserve(static_cast<size_t>(n) * sizeof(float)));
	const frame_keys keys = make_frame_keys(seed);
	launch_kat_random(keys.a, keys.b, pixel, sample, n, ctx->kat_out.as<float>(), nullptr);
	RT_HIP_TRY(hipGetLastError());
	RT_HIP_TRY(hipMemcpy(out, ctx->kat_out.ptr, static_cast<size_t>(n) * sizeof(float), hipMemcpyDeviceToHost));
	return ok();
}

extern "C" rt_hip_status rt_hip_kat_closest_hit(rt_hip_ctx* ctx,
												uint32_t n,
												const float* origins,
												const float* directions,
												float* out_distance,
												uint32_t* out_kind,
												uint32_t* out_index,
												float* out_normal)
{
	if (!ctx || !n || !origins || !directions || !out_distance || !out_kind || !out_index || !out_normal)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_kat_closest_hit: invalid argument");
	if (!ctx->have_scene)
		return fail(RT_HIP_NO_SCENE, "rt_hip_kat_closest_hit: no scene uploaded");
	RT_HIP_TRY(hipSetDevice(ctx->device));
	const size_t vec_bytes = static_cast<size_t>(n) * 3 * sizeof(float);
	const size_t scalar_bytes = static_cast<size_t>(n) * sizeof(float);
	RT_HIP_TRY(ctx->kat_in.reserve(2 * vec_bytes));
	RT_HIP_TRY(ctx->kat_out.reserve(vec_bytes + 3 * scalar_bytes));
	unsigned char* in = ctx->kat_in.as<unsigned char>();
	unsigned char* out = ctx->kat_out.as<unsigned char>();
	RT_HIP_TRY(hipMemcpy(in, origins, vec_bytes, hipMemcpyHostToDevice));
	RT_HIP_TRY(hipMemcpy(in + vec_bytes, directions, vec_bytes, hipMemcpyHostToDevice));
	float* d_distance = reinterpret_cast<float*>(out);
	uint32_t* d_kind = reinterpret_cast<uint32_t*>(out + scalar_bytes);
	uint32_t* d_index = reinterpret_cast<uint32_t*>(out + 2 * scalar_bytes);
	float* d_normal = reinterpret_cast<float*>(out + 3 * scalar_bytes);
	launch_kat_closest_hit(ctx->scene, n, reinterpret_cast<const float*>(in), reinterpret_cast<const float*>(in + vec_bytes), d_distance, d_kind, d_index, d_normal, nullptr);
	RT_HIP_TRY(hipGetLastError());
	RT_HIP_TRY(hipMemcpy(out_distance, d_distance, scalar_bytes, hipMemcpyDeviceToHost));
	RT_HIP_TRY(hipMemcpy(out_kind, d_kind, scalar_bytes, hipMemcpyDeviceToHost));
	RT_HIP_TRY(hipMemcpy(out_index, d_index, scalar_bytes, hipMemcpyDeviceToHost));
	RT_HIP_TRY(hipMemcpy(out_normal, d_normal, vec_bytes, hipMemcpyDeviceToHost));
	return ok();
}

extern "C" rt_hip_status rt_hip_kat_sqrt_div(rt_hip_ctx* ctx, uint32_t n, const float* a, const float* b, float* out_sqrt, float* out_div)
{
	if (!ctx || !n || !a || !b || !out_sqrt || !out_div)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_kat_sqrt_div: invalid argument");
	RT_HIP_TRY(hipSetDevice(ctx->device));
	const size_t bytes = static_cast<size_t>(n) * sizeof(float);
	RT_HIP_TRY(ctx->kat_in.reserve(2 * bytes));
	RT_HIP_TRY(ctx->kat_out.reserve(2 * bytes));
	unsigned char* in = ctx->kat_in.as<unsigned char>();
	unsigned char* out = ctx->kat_out.as<unsigned char>();
	RT_HIP_TRY(hipMemcpy(in, a, bytes, hipMemcpyHostToDevice));
	RT_HIP_TRY(hipMemcpy(in + bytes, b, bytes, hipMemcpyHostToDevice));
	launch_kat_sqrt_div(n, reinterpret_cast<const float*>(in), reinterpret_cast<const float*>(in + bytes), reinterpret_cast<float*>(out), reinterpret_cast<float*>(out + bytes), nullptr);
	RT_HIP_TRY(hipGetLastError());
	RT_HIP_TRY(hipMemcpy(out_sqrt, out, bytes, hipMemcpyDeviceToHost));
	RT_HIP_TRY(hipMemcpy(out_div, out + bytes, bytes, hipMemcpyDeviceToHost));
	return ok();
}

extern "C" rt_hip_status rt_hip_kat_exhaustive_math(rt_hip_ctx* ctx, uint64_t out_mismatches[3], uint32_t out_first[3])
{
	if (!ctx || !out_mismatches || !out_first)
		return fail(RT_HIP_INVALID_ARGUMENT, "rt_hip_kat_exhaustive_math: NULL argument");
	RT_HIP_TRY(hipSetDevice(ctx->device));
	RT_HIP_TRY(ctx->kat_out.reserve(6 * sizeof(unsigned long long)));
	unsigned long long host[6] = { 0, ~0ull, 0, ~0ull, 0, ~0ull };
	RT_HIP_TRY(hipMemcpy(ctx->kat_out.ptr, host, sizeof(host), hipMemcpyHostToDevice));
	launch_kat_exhaustive_math(ctx->kat_out.as<unsigned long long>(), nullptr);
	RT_HIP_TRY(hipGetLastError());
	RT_HIP_TRY(hipMemcpy(host, ctx->kat_out.ptr, sizeof(host), hipMemcpyDeviceToHost));
	for (int f = 0; f < 3; f++)
	{
		out_mismatches[f] = host[2 * f];
		out_first[f] = host[2 * f] ? static_cast<uint32_t>(host[2 * f + 1] - 1ull) : 0u;
	}
	return ok();
}
