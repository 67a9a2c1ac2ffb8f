// rt_amd/csrc/context.hip — library entry points and the lifetime of a context: one GPU (rt_hip_create), several GPUs of
// this process behind one render() (rt_hip_create_multi), one rank of a renderer whose ranks are processes
// (rt_hip_join_ranks: RCCL; rt_hip_join_frame_group: one shared back buffer).
//
// Mirrors the lifetime of a renderer object in the reference: created by description::create (src/renderer.hpp:39), destroyed
// through the virtual destructor (src/renderer.hpp:13), owned by a unique_ptr in src/main.cpp:49-53,97-99.
#include "internal.hpp"

#include <algorithm>
#include <exception>
#include <new>

using namespace rt_hip;

namespace rt_hip
{
	namespace
	{
		thread_local std::string g_last_error;
	}

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

	const std::string& last_error()
	{
		return g_last_error;
	}
}

extern "C" uint32_t rt_hip_abi_version(void)
{
	return RT_HIP_ABI_VERSION;
}

extern "C" const char* rt_hip_last_error(void)
{
	return last_error().c_str();
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

namespace rt_hip
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
	unpin_frame(ctx); // a context never outlives its page-lock on the caller's memory
	ctx->delivery.reset(); // (joins the delivery threads, frees the module's own frame)
	ctx->scene_columns.release();
	ctx->scene_staging.release();
	ctx->item_sums.release();
	ctx->pixel_done.release();
	ctx->counters.release();
	ctx->frame_rgb.release();
	ctx->staging_rgb.release();
	ctx->stripes_rgba.release();
	ctx->stripes_rgb.release();
	ctx->gathered_rgba.release();
	ctx->gathered_rgb.release();
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

extern "C" uint32_t rt_hip_live_frame_locks(void)
{
	return live_frame_locks();
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
