/*
 * rt_hip.h — C ABI of the MI355X (gfx950) renderer module for marzer/rt.
 *
 * This is the drop-in boundary for ONE path of the reference: mg_ray_tracer::render
 * (reference src/renderers/mg_ray_tracer.cpp:178-205) behind renderer_interface::render
 * (reference src/renderer.hpp:9-14).  Everything that crosses it is plain C: pointers, sizes,
 * POD structs.  No C++ types, no exceptions, no torch types.
 *
 * The reference-side binding (a ~60 line renderer that gathers the soagen column pointers and
 * the camera matrix and calls rt_hip_render) is shown in INTEGRATION.md and shim/hip_ray_tracer.cpp.
 *
 * Every function that can fail returns an rt_hip_status (0 = success).  On failure a
 * human-readable message is available from rt_hip_last_error() (thread-local).  The reference's
 * render() is `noexcept` and returns void (src/renderer.hpp:11); the shim therefore logs the
 * message in the reference's `error: ...` style (src/main.cpp:43-47) and leaves the caller's
 * pre-cleared frame (src/main.cpp:318) untouched.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 6 (round 5): arithmetic contract v4 — the same entry points, DIFFERENT frames for a given seed — and three more words at the
 * end of rt_hip_phases.  5 (round 4): rt_hip_live_frame_locks added; the known-answer hooks of rounds 1-3 moved to the test-only
 * library (include/rt_hip_kat.h); rt_hip_render without RT_HIP_FLAG_PERSISTENT_FRAME delivers through the module's own frame.
 * A consumer checks rt_hip_abi_version() against the header it was compiled with. */
#define RT_HIP_ABI_VERSION 6u

/* librt_hip.so is built with hidden visibility: these entry points are ALL it exports. */
#if defined(__GNUC__)
#define RT_HIP_API __attribute__((visibility("default")))
#else
#define RT_HIP_API
#endif

typedef enum rt_hip_status
{
	RT_HIP_OK				 = 0,
	RT_HIP_INVALID_ARGUMENT	 = 1, /* null pointer, zero size, out-of-range material index, ... */
	RT_HIP_NO_DEVICE		 = 2, /* no gfx950 device visible / bad device ordinal */
	RT_HIP_RUNTIME_ERROR	 = 3, /* a HIP call failed; message carries hipGetErrorString */
	RT_HIP_NO_SCENE			 = 4, /* render requested before a scene was uploaded */
	RT_HIP_UNSUPPORTED		 = 5, /* unknown flag bits */
	RT_HIP_TIMEOUT			 = 6  /* rt_hip_join_ranks: the other ranks did not arrive within the deadline */
} rt_hip_status;

/* Material kinds, in the order of `enum class material_type` (reference src/common.hpp:105-115).
 * mg_ray_tracer shades metal with metal_scatter and EVERYTHING else with lambert_scatter
 * (reference src/renderers/mg_ray_tracer.cpp:142-152). */
enum
{
	RT_HIP_MATERIAL_LAMBERT	   = 0,
	RT_HIP_MATERIAL_METAL	   = 1,
	RT_HIP_MATERIAL_DIELECTRIC = 2,
	RT_HIP_MATERIAL_AIR		   = 3,
	RT_HIP_MATERIAL_VACUUM	   = 4,
	RT_HIP_MATERIAL_WATER	   = 5,
	RT_HIP_MATERIAL_ICE		   = 6,
	RT_HIP_MATERIAL_DIAMOND	   = 7,
	RT_HIP_MATERIAL_COUNT	   = 8
};

/*
 * The scene as the renderer sees it: the soagen struct-of-arrays columns of rt::spheres / rt::planes /
 * rt::materials (reference src/soa.toml:6-33, accessors src/soa.hpp:177-199) plus the two scalars of
 * rt::scene (reference src/scene.hpp:10-11) and the camera's inverse view-projection for the frame
 * size being rendered (reference src/camera.hpp:18,122-137).
 *
 * All pointers are HOST pointers borrowed for the duration of the call that takes the struct.
 * Columns hold exactly n_* valid rows; soagen's padding rows past size() are never read
 * (reference vendor/soagen.hpp:3777,7075-7082).  A count of 0 allows the matching pointers to be NULL.
 */
typedef struct rt_hip_scene
{
	/* rt::spheres — center_x/center_y/center_z/radius/material columns (src/soa.toml:25-33) */
	uint32_t n_spheres;
	const float* sphere_center_x;
	const float* sphere_center_y;
	const float* sphere_center_z;
	const float* sphere_radius;
	const uint32_t* sphere_material;

	/* rt::planes — normal_x/normal_y/normal_z/d/material columns (src/soa.toml:15-23);
	 * plane equation n·p + d = 0 with |n| = 1 (src/scene.cpp:580-583) */
	uint32_t n_planes;
	const float* plane_normal_x;
	const float* plane_normal_y;
	const float* plane_normal_z;
	const float* plane_d;
	const uint32_t* plane_material;

	/* rt::materials — type/albedo/roughness/reflectivity columns (src/soa.toml:6-13);
	 * albedo is rt::colour = 4 floats r,g,b,a per material (src/colour.hpp:17-57) */
	uint32_t n_materials;
	const uint32_t* material_type;
	const float* material_albedo;
	const float* material_roughness;
	const float* material_reflectivity;

	/* rt::scene::samples_per_pixel / max_bounces (src/scene.hpp:10-11), both >= 1 */
	uint32_t samples_per_pixel;
	uint32_t max_bounces;

	/* viewport::inverse_view_projection for THIS frame size (src/camera.hpp:18,134).
	 * Fixed order, independent of muu's storage order: element [r*4 + c] = m(r, c), so that
	 * transform_position(v) = (M * (v.x, v.y, v.z, 1)).xyz / .w  with row r = sum_c m(r,c) * v_c. */
	float inverse_view_projection[16];

	/* rt::boxes — center_x..center_z/extents_x..extents_z/material columns (src/soa.toml:35-45); extents are half
	 * sizes (muu::bounding_box).  mg_ray_tracer never hits boxes (mg_ray_tracer.cpp:89-93): only the preview
	 * (RT_HIP_FLAG_PREVIEW, reference src/renderers/rasterizer.cpp:57) draws them. */
	uint32_t n_boxes;
	const float* box_center_x;
	const float* box_center_y;
	const float* box_center_z;
	const float* box_extents_x;
	const float* box_extents_y;
	const float* box_extents_z;
	const uint32_t* box_material;
} rt_hip_scene;

/*
 * How the image is split across the GPUs of one node: rows are grouped into stripes of `stripe_rows`
 * rows; stripe b belongs to rank (b % world).  A rank renders only its own stripes into a compact
 * buffer of rt_hip_local_rows() rows x W pixels (stripe b of the image = stripe b / world of the local
 * buffer).  Random streams are keyed by the GLOBAL pixel index, so the assembled image does not depend
 * on `world`.  {0, 1, any} = the whole image.  (New: the reference is single-process,
 * src/renderers/mg_ray_tracer.cpp:203.)
 */
typedef struct rt_hip_partition
{
	uint32_t rank;
	uint32_t world;
	uint32_t stripe_rows;
} rt_hip_partition;

#define RT_HIP_DEFAULT_STRIPE_ROWS 8u

/* Work counters of the most recent render on a context (all ranks count their own share). */
typedef struct rt_hip_stats
{
	uint64_t primary_samples; /* pixels rendered by this rank x samples_per_pixel */
	uint64_t segments;		  /* calls of trace() that did not return at the bounce limit check, i.e. closest-hit queries */
	uint64_t sphere_tests;	  /* segments x n_spheres */
	uint64_t plane_tests;	  /* segments x n_planes */
	float render_ms;		  /* device time of the render kernel(s), HIP events on the launch stream */
	float upload_ms;		  /* host wall time of the last scene upload */
	float readback_ms;		  /* rt_hip_render only: host wall time from "kernels done" to "frame in the caller's buffer" */
	uint32_t kernel_variant;  /* which kernel ran: RT_HIP_KERNEL_* */
} rt_hip_stats;

enum
{
	RT_HIP_KERNEL_NONE		= 0,
	RT_HIP_KERNEL_RESIDENT	= 1, /* a pixel tile per wave; planes (and fewer than 40 spheres) resident in LDS for the lifetime of the workgroup (<= 1024 of them) */
	RT_HIP_KERNEL_TILED		= 2, /* primitives streamed from the SoA columns through LDS in tiles (large scenes) */
	RT_HIP_KERNEL_SMALL		= 3, /* <= 8 primitives (>= 1 sphere, <= 3 planes): scene in scalar registers, scan fully unrolled */
	RT_HIP_KERNEL_PREVIEW	= 4, /* RT_HIP_FLAG_PREVIEW: one primary ray per pixel, N.L shading */
	RT_HIP_KERNEL_STREAMED	= 5	 /* primitives read from the table in HBM/L2 with wave-uniform scalar loads: no staging, no barriers */
};

/* Render flags.  0 = the parity contract: arithmetic bit-identical to oracle/ (see DESIGN.md §3). */
enum
{
	RT_HIP_FLAG_NONE = 0u,
	/* force the LDS-tiled kernel even for scenes that fit a smaller one (testing) */
	RT_HIP_FLAG_FORCE_TILED = 1u << 0,
	/* force the LDS-resident kernel for scenes that would take the scalar-register one (testing) */
	RT_HIP_FLAG_FORCE_RESIDENT = 1u << 1,
	/* rt_hip_render only, OPT-IN: zero-copy delivery.  By default (flag absent) the kernels store finished pixels into a
	 * page-locked frame the MODULE owns and host threads carry them on into `pixels_rgba8888` while the frame is still being
	 * traced: the caller's buffer is plain memory to the module, touched only by CPU stores inside the call, and may be
	 * freed, re-created or recycled at will between two calls (what rt does on every resize, reference src/window.cpp:198-203).
	 * WITH the flag the caller promises that `pixels_rgba8888` stays allocated, at this address and size, until the next
	 * rt_hip_render call on this context with another buffer, rt_hip_forget_frame or rt_hip_destroy.  The module then
	 * page-locks the CALLER's buffer once and maps it into the GPU's address space: the kernels store every finished pixel
	 * straight into it and no second copy of the frame exists (a few tens of microseconds less per 1080p frame).  A caller
	 * that frees the buffer and may get the same address back from its allocator before the next frame MUST call
	 * rt_hip_forget_frame() in between: a mapping re-created under a live page-lock is a GPU memory fault
	 * (INTEGRATION.md §3). */
	RT_HIP_FLAG_PERSISTENT_FRAME = 1u << 2,
	/* shade with sm_ray_tracer's scatter table (reference src/renderers/sm_ray_tracer.cpp:221-236) instead of
	 * mg_ray_tracer's: dielectric, air, vacuum, water and ice refract/reflect through dielectric_scatter (:181-219),
	 * with the material's reflectivity as index of refraction.  Everything else is unchanged.  Opt-in: the parity
	 * contract of this module is mg_ray_tracer, under which those materials are lambert. */
	RT_HIP_FLAG_SM_MATERIALS = 1u << 3,
	/* draw the fast preview of reference src/renderers/rasterizer.cpp:24-85 instead of tracing paths (rt shows it at
	 * low resolution while the camera moves, src/main.cpp:106,319): ONE ray through each pixel centre, closest hit
	 * over planes, then boxes, then spheres, shaded 0.25 + 0.75 * albedo * (N . direction to the eye), the sky where
	 * nothing is hit.  Deterministic: seed, samples_per_pixel and max_bounces are not read; d_rgb_f32 / rgb_f32
	 * receive the colour before packing.  Partition, gather and assemble work as for the traced frame. */
	RT_HIP_FLAG_PREVIEW = 1u << 4,
	/* force the scalar-streamed kernel (it is what scenes above about 1300 primitives get) */
	RT_HIP_FLAG_FORCE_STREAMED = 1u << 5,
	/* Contract "v2-fast": the kernels built with the hardware's reciprocal / square-root / reciprocal-square-root
	 * approximations (about 1 ulp each, no correction steps, no range guards) and with multiply-adds contracted — the
	 * latitude the reference's own build takes (-ffast-math -ffp-contract=fast, meson.build:153-160).  Same random
	 * streams, same algorithm, same operation order otherwise.  The frame is NOT bit-identical to the oracle's: the
	 * per-pixel float mean agrees to a few 1e-6 relative wherever no sample's hit/miss decision flipped at a silhouette
	 * (tests/test_gpu_fast.py states and checks the bounds).  Not available with RT_HIP_FLAG_SM_MATERIALS or
	 * RT_HIP_FLAG_PREVIEW.  Opt-in; flags == 0 stays the parity contract. */
	RT_HIP_FLAG_FAST = 1u << 6,
	/* rt_hip_render only: keep the work counters and the kernel's device time of this frame for a later
	 * rt_hip_stats_fetch / rt_hip_phases_fetch even though `stats` is NULL.  A call with stats == NULL and without this flag
	 * is the plug-in's call (shim/hip_ray_tracer.cpp): nothing but the launch and the wait is enqueued — no timing events, no
	 * zeroing and read-back of the counters — which is what the launch-bound low-resolution preview frames of
	 * reference src/main.cpp:315-321 want; rt_hip_stats_fetch after such a frame reports render_ms = 0 and segments = 0. */
	RT_HIP_FLAG_STATS = 1u << 7,
	/* Work items of the small-scene kernels are whole 16-sample chunks, or — for launches that hold only a few chunks per
	 * lane of the device — half chunks (HISTORY.md §5 "Half-chunk items"): the launch code decides by the size of the launch, and the frame is
	 * the same bit for bit either way.  These two take the decision away from it (tests; never both). */
	RT_HIP_FLAG_FORCE_HALF_CHUNKS = 1u << 8,
	RT_HIP_FLAG_FORCE_WHOLE_CHUNKS = 1u << 9
};

typedef struct rt_hip_ctx rt_hip_ctx;

/* ---- library ---------------------------------------------------------------------------------------------------- */

RT_HIP_API uint32_t rt_hip_abi_version(void);

/* Message for the most recent failure on the calling thread ("" if none).  Never NULL. */
RT_HIP_API const char* rt_hip_last_error(void);

/* Number of visible HIP devices.  Replaces nothing in the reference (CPU-only). */
RT_HIP_API rt_hip_status rt_hip_device_count(int* count);

/* ---- context ---------------------------------------------------------------------------------------------------- */

/* One context per GPU per renderer instance; owns the device copy of the scene, the stats block and
 * staging buffers.  Mirrors the lifetime of a renderer object in the reference: created by
 * description::create (src/renderer.hpp:39), destroyed through the virtual destructor (src/renderer.hpp:13). */
RT_HIP_API rt_hip_status rt_hip_create(rt_hip_ctx** out_ctx, int device);
RT_HIP_API void rt_hip_destroy(rt_hip_ctx* ctx);

/*
 * One renderer over SEVERAL GPUs of this process — what lets the reference's single blocking
 * render(scene, back_buffer) call (src/renderers/mg_ray_tracer.cpp:178-205; the caller waits in it, src/window.cpp:213-217)
 * use a whole node.  The context owns one member per entry of `devices`; rt_hip_render() on it then
 *   - keeps the scene replicated on every member (each re-uploads only when the host columns changed),
 *   - launches member r's share of the frame — rt_hip_partition{r, n_devices, RT_HIP_DEFAULT_STRIPE_ROWS} — on that
 *     member's own stream, all members concurrently,
 *   - collects the compact stripe buffers on devices[0] with ONE gather over xGMI (RCCL: a communicator per member from
 *     ncclCommInitAll, one ncclGather inside a group call),
 *   - de-interleaves them there and copies the frame to the caller's host buffer once.
 * The frame is bit-identical to the single-GPU one (random streams are keyed by the global pixel index).
 *   devices    device ordinals, rank order; devices[0] is the root.  NULL = 0 .. n_devices-1.
 *   multi_flags RT_HIP_MULTI_*.
 * Every other entry point (rt_hip_scene_upload, rt_hip_render_device, the known-answer calls, ...) used on such a context
 * acts on the root member alone.  n_devices == 1 is allowed and still goes through the communicator.
 */
enum
{
	RT_HIP_MULTI_NONE = 0u,
	/* move the stripes with hipMemcpyPeerAsync instead of RCCL.  RCCL refuses a communicator that names a device twice;
	 * with this flag `devices` may (a one-GPU box can then exercise the whole multi-member path: tests). */
	RT_HIP_MULTI_PEER_COPY = 1u << 0,
	/* no gather: when the caller's back buffer is page-locked (RT_HIP_FLAG_PERSISTENT_FRAME) every member's kernel stores
	 * its pixels straight into their image rows of that buffer, each over its own PCIe link, and rt_hip_render returns when
	 * the last member's launch is done — no stripe buffers, no collective, no de-interleave, no copy.  What a frame of a
	 * few milliseconds wants on 8 GPUs, where gather + assemble + copy cost as much as a member's share of the tracing.
	 * Opt-in: the documented default remains the single RCCL gather.  (Calls without the flag, or that ask for the float
	 * mean, take the gathered way.) */
	RT_HIP_MULTI_DIRECT_FRAME = 1u << 1
};
RT_HIP_API rt_hip_status rt_hip_create_multi(rt_hip_ctx** out_ctx, const int* devices, int n_devices, uint32_t multi_flags);

/*
 * The same renderer with ONE PROCESS PER GPU (how `torchrun` launches a job): every process creates one rank of it.
 *   rt_hip_unique_id   on one process: a fresh RCCL id (rccl.h: ncclGetUniqueId); hand the bytes to every rank by any means
 *                      (the harness broadcasts them with torch.distributed).
 *   rt_hip_create_rank on every process, collectively (rccl.h: ncclCommInitRank): rank `rank` of `world` on `device`.
 * rt_hip_render() on such a context is collective too: every rank passes the same scene, size, seed and flags and renders
 * rt_hip_partition{rank, world, 8}; the stripes are gathered on rank 0 with the same single ncclGather; rank 0 assembles
 * and fills ITS caller's `pixels_rgba8888` (the other ranks may pass NULL and return once their stripes are sent).
 * rgb_f32 must be NULL or non-NULL on all ranks alike.  Nothing in the reference corresponds (it is one process).
 */
#define RT_HIP_UNIQUE_ID_BYTES 128
RT_HIP_API rt_hip_status rt_hip_unique_id(char out_id[RT_HIP_UNIQUE_ID_BYTES]);
RT_HIP_API rt_hip_status rt_hip_create_rank(rt_hip_ctx** out_ctx, int device, int rank, int world, const char id[RT_HIP_UNIQUE_ID_BYTES]);
/*
 * rt_hip_create_rank in two halves, so that the part that can fail on ONE rank alone — no such device, not a gfx950, out
 * of memory — is over before anything collective starts (a rank that fails inside rt_hip_create_rank leaves the others
 * waiting in ncclCommInitRank):
 *   1. every process: rt_hip_create(&ctx, device)                                — local, may fail alone
 *   2. the launcher lets the ranks agree that all of them hold a context (bench.py: one all_reduce), and only then
 *   3. every process: rt_hip_join_ranks(ctx, rank, world, id, timeout_ms)        — collective (rccl.h: ncclCommInitRank)
 * rt_hip_join_ranks waits at most `timeout_ms` milliseconds for the communicator (0 = RT_HIP_JOIN_TIMEOUT_MS from the
 * environment, else 120 000) and then gives up with RT_HIP_TIMEOUT; the context stays a valid single-GPU context, which the
 * caller may use or destroy.  On success the context is what rt_hip_create_rank returns.
 */
RT_HIP_API rt_hip_status rt_hip_join_ranks(rt_hip_ctx* ctx, int rank, int world, const char id[RT_HIP_UNIQUE_ID_BYTES], uint32_t timeout_ms);

/*
 * ONE PROCESS PER GPU WITHOUT AN EXCHANGE STEP — what RT_HIP_MULTI_DIRECT_FRAME is for one process, for `torchrun`-style
 * jobs.  The caller's back buffer is a SHARED mapping (shm_open / memfd + mmap(MAP_SHARED)) that every rank's process
 * maps; each rank page-locks its own mapping and its kernel stores the rank's stripes straight into their image rows,
 * over that GPU's own PCIe link, while the frame is still being traced.  No RCCL, no stripe buffers, no assemble, no
 * copy: the gathered form puts 7/8 of an 8-GPU frame through rank 0's single PCIe link AFTER the tracing (0.14 ms for a
 * 1080p frame whose tracing takes 0.36 ms per rank).  The ranks meet in a control block in POSIX shared memory
 * (rt_amd/csrc/frame_group.hpp) twice per frame — "everybody is in the call", "everybody's stripes are in place".
 *   rt_hip_join_frame_group(ctx, rank, world, name, timeout_ms) on a context from rt_hip_create, on every process,
 *       collectively.  `name` is a fresh shm_open name ("/rt_hip_<something unique>") all ranks were handed by any means;
 *       rank 0 creates the object, the others wait for it; when all have mapped it the name is removed again.  Waits at
 *       most timeout_ms (0 = RT_HIP_JOIN_TIMEOUT_MS, else 120 000) -> RT_HIP_TIMEOUT; the context then stays a plain one.
 *   rt_hip_render(ctx, scene, pixels, ...) on EVERY rank, with the same scene, size, seed and flags and with `pixels` =
 *       that process's mapping of the one shared buffer (the library checks all of this: a rank called with other
 *       arguments, or whose `pixels` is not the memory rank 0 renders into, fails the frame on every rank).  Returns on
 *       every rank when the WHOLE frame is in the buffer.  RT_HIP_FLAG_PERSISTENT_FRAME is implied; rgb_f32 must be NULL.
 *       `stats` / rt_hip_stats_fetch give the whole frame (counts summed over the ranks, the slowest rank's kernel time;
 *       a rank that was called without `stats` and without RT_HIP_FLAG_STATS contributes zeros),
 *       rt_hip_member_stats(ctx, r, ..) / rt_hip_member_device(ctx, r, ..) rank r's share and device, for any r < world.
 * A rank that fails, leaves (rt_hip_destroy) or stays away longer than RT_HIP_GROUP_DEADLINE_MS (default 120 000) breaks
 * the group: every rank's current and later rt_hip_render returns an error naming the rank and the reason; the renderer
 * is then destroyed and made anew.  Nothing in the reference corresponds (it is one process, one thread pool).
 */
RT_HIP_API rt_hip_status rt_hip_join_frame_group(rt_hip_ctx* ctx, int rank, int world, const char* name, uint32_t timeout_ms);

/* How the stripes of a multi-GPU frame reach the root: what a context was created with (and what a bench line should say). */
enum
{
	RT_HIP_TRANSPORT_NONE		  = 0, /* one GPU: nothing to exchange */
	RT_HIP_TRANSPORT_RCCL_GATHER  = 1, /* one ncclGather to rank 0 */
	RT_HIP_TRANSPORT_PEER_COPY	  = 2, /* RT_HIP_MULTI_PEER_COPY: hipMemcpyPeerAsync into the root */
	RT_HIP_TRANSPORT_DIRECT_FRAME = 3, /* RT_HIP_MULTI_DIRECT_FRAME took effect in the most recent frame: no exchange at all */
	RT_HIP_TRANSPORT_SHARED_FRAME = 4  /* rt_hip_join_frame_group: every rank's process stores into ONE shared back buffer */
};
/* What member `member` of the context talks through, as RCCL itself reports it (rccl.h: ncclCommCount, ncclCommUserRank,
 * ncclCommCuDevice): the communicator's size, this member's rank in it, and the device the communicator lives on.  Contexts
 * without a communicator (one GPU, peer copies) report the context's own world / rank / device.  Any out pointer may be NULL. */
RT_HIP_API rt_hip_status rt_hip_comm_info(const rt_hip_ctx* ctx, int member, int* out_ranks, int* out_rank, int* out_device, uint32_t* out_transport);

/* number of members of a context (1 for rt_hip_create) and the device of member `rank` */
RT_HIP_API rt_hip_status rt_hip_member_count(const rt_hip_ctx* ctx, int* out_count);
RT_HIP_API rt_hip_status rt_hip_member_device(const rt_hip_ctx* ctx, int rank, int* out_device);
/* counters of member `rank`'s share of the most recent rt_hip_render (rt_hip_stats_fetch on a multi context returns the
 * whole frame: counts summed, render_ms = the slowest member) */
RT_HIP_API rt_hip_status rt_hip_member_stats(rt_hip_ctx* ctx, int rank, rt_hip_stats* out_stats);

/* ---- partition helpers (pure host arithmetic; usable without a GPU) ------------------------------------------- */

/* Rows of an H-row image owned by part->rank. */
RT_HIP_API rt_hip_status rt_hip_local_rows(uint32_t height, const rt_hip_partition* part, uint32_t* out_rows);
/* max over ranks of rt_hip_local_rows: the per-rank buffer height used for the equal-sized gather. */
RT_HIP_API rt_hip_status rt_hip_padded_local_rows(uint32_t height, const rt_hip_partition* part, uint32_t* out_rows);

/*
 * What rt_hip_render does with the caller's columns before anything touches a GPU, on its own (pure host code; usable
 * without a device): the pointer / count check, the material-index check (the reference's loader refuses out-of-range
 * indices, src/scene.cpp:568-574) and the fingerprint that decides whether the columns resident in HBM are still the
 * caller's (rt has no scene version counter, src/main.cpp:233-311).  Reads exactly n_* rows of every column — never the
 * padding rows soagen keeps behind size() (vendor/soagen.hpp:3777,7075-7082).  out_fingerprint may be NULL.
 */
RT_HIP_API rt_hip_status rt_hip_scene_check(const rt_hip_scene* scene, uint64_t* out_fingerprint);

/* ---- the hot path ----------------------------------------------------------------------------------------------- */

/* Copy the scene columns to HBM (once per scene/camera change).  The reference has no scene version
 * counter (src/main.cpp:233-311), so rt_hip_render() calls this every frame; a caller that knows the
 * scene is unchanged keeps it resident and calls rt_hip_render_device() only. */
RT_HIP_API rt_hip_status rt_hip_scene_upload(rt_hip_ctx* ctx, const rt_hip_scene* scene);

/*
 * Render this rank's stripes of a width x height frame from the resident scene.
 *   d_rgba8   DEVICE buffer, padded_local_rows x width uint32, receives RGBA8888 packed exactly as
 *             rt::colour::operator uint32_t (src/colour.hpp:101-106); row-major, no pitch (src/image.hpp:143-147).
 *   d_rgb_f32 optional DEVICE buffer, padded_local_rows x width x 3 floats: the per-pixel mean radiance
 *             before the sqrt "gamma" (mg_ray_tracer.cpp:195), for float-level parity checks.  May be NULL.
 *   seed      key of the counter-based random streams (the reference seeds from std::random_device,
 *             src/random.cpp:12-13, and is not reproducible).
 *   stream    hipStream_t to launch on (NULL = the default stream).  Asynchronous: returns after enqueue.
 */
RT_HIP_API rt_hip_status rt_hip_render_device(rt_hip_ctx* ctx,
								   uint32_t width,
								   uint32_t height,
								   uint64_t seed,
								   uint32_t flags,
								   const rt_hip_partition* part, /* NULL = whole image */
								   uint32_t* d_rgba8,
								   float* d_rgb_f32,
								   void* stream);

/* Rank 0, after the gather: de-interleave `world` compact per-rank buffers (each padded_local_rows x width,
 * concatenated in rank order) into the width x height frame.  Device to device, asynchronous on `stream`. */
RT_HIP_API rt_hip_status rt_hip_assemble_device(rt_hip_ctx* ctx,
									 uint32_t width,
									 uint32_t height,
									 uint32_t world,
									 uint32_t stripe_rows,
									 const uint32_t* d_gathered,
									 uint32_t* d_frame,
									 void* stream);

/* Synchronise with the last render on this context and read its counters. */
RT_HIP_API rt_hip_status rt_hip_stats_fetch(rt_hip_ctx* ctx, rt_hip_stats* out_stats);

/*
 * Where the time of the most recent rt_hip_render went (frames rendered with `stats` or RT_HIP_FLAG_STATS; otherwise the
 * device times are 0).  Device times come from HIP events on the root's stream; on one GPU only render_ms is non-zero.
 * Nothing in the reference corresponds: it prints no timing at all (src/main.cpp:30-47).
 */
typedef struct rt_hip_phases
{
	float render_ms;	 /* the root member's kernel (its share of the frame) */
	float gather_ms;	 /* end of the root's kernel -> every rank's stripes are on the root (waits for the slowest rank) */
	float assemble_ms;	 /* de-interleave; with a page-locked back buffer this IS the transfer to the host (stores over PCIe) */
	float copy_ms;		 /* device-to-host copy of the assembled frame (0 when it was assembled straight into the back buffer) */
	float host_issue_ms; /* host wall time from entry into rt_hip_render until everything was enqueued */
	float host_wait_ms;	 /* host wall time blocked until the frame was complete */
	uint32_t transport;	 /* RT_HIP_TRANSPORT_* that this frame took */
	uint32_t scene_resident; /* 1: the columns' fingerprint matched, nothing was uploaded */
	/* ABI 6.  The default frame mode (module-owned page-locked frame + host carrier threads): how the frame reached the caller's
	 * buffer.  carrier_bands = 64 KB bands the frame was carried in (0: the kernels stored straight into the caller's page-locked
	 * buffer, or a multi-GPU / preview path); carrier_bands_early = of those, carried over BEFORE the stream had drained — by
	 * the helper threads, while the frame was being traced; carrier_helpers = helper threads the context runs (0: the caller's own
	 * thread carries everything after the kernel).  A frame whose carrier_bands_early is 0 although carrier_helpers is not was
	 * carried by the caller's thread alone: the helpers never got to run (frame delivery). */
	uint32_t carrier_bands;
	uint32_t carrier_bands_early;
	uint32_t carrier_helpers;
} rt_hip_phases;
RT_HIP_API rt_hip_status rt_hip_phases_fetch(rt_hip_ctx* ctx, rt_hip_phases* out_phases);

/*
 * The drop-in for renderer_interface::render(const scene&, image_view&, muu::thread_pool&)
 * (src/renderer.hpp:11; mg_ray_tracer.cpp:178): upload `scene`, render the whole frame on the context's
 * GPU, and copy it into the caller's HOST pixel buffer (image_view::data(), width*height uint32) before
 * returning — the caller presents it immediately (src/window.cpp:215-216).
 *   rgb_f32  optional HOST buffer of 3*width*height floats (pre-gamma mean), may be NULL.
 *   stats    optional, may be NULL.
 */
RT_HIP_API rt_hip_status rt_hip_render(rt_hip_ctx* ctx,
							const rt_hip_scene* scene,
							uint32_t* pixels_rgba8888,
							uint32_t width,
							uint32_t height,
							uint64_t seed,
							uint32_t flags,
							float* rgb_f32,
							rt_hip_stats* stats);

/* Drop the page-lock taken under RT_HIP_FLAG_PERSISTENT_FRAME (see there).  Waits for the context's stream first. */
RT_HIP_API void rt_hip_forget_frame(rt_hip_ctx* ctx);

/* Page-locks on CALLERS' memory that contexts of this process hold right now: 0 unless somebody rendered with
 * RT_HIP_FLAG_PERSISTENT_FRAME (or as a rank of a frame group) and has neither moved on to another buffer nor called
 * rt_hip_forget_frame / rt_hip_destroy since.  A diagnostic for integrators and for this repository's tests (which assert
 * 0 after every GPU test): memory the module no longer knows about can never be written by it. */
RT_HIP_API uint32_t rt_hip_live_frame_locks(void);

#ifdef __cplusplus
}
#endif

#endif /* RT_HIP_H */
