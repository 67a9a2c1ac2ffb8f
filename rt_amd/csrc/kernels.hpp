// rt_amd/csrc/kernels.hpp — host-visible launch interface of the gfx950 kernels (internal to librt_hip.so).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rt_hip.h" // (the render flags: half_chunk_choice)

namespace rt_hip
{
	// The scene resident in HBM.
	//   * the soagen columns exactly as the reference fills them at load (src/scene.cpp:583,595): the LDS-tiled
	//     kernel streams these, one coalesced dword per lane per column;
	//   * per-primitive tables derived from them at upload: (center, radius^2) / (normal, d) as float4, and the
	//     shading inputs of the primitive's material — (albedo.rgb * reflectivity, roughness) and a metal flag
	//     (mg_ray_tracer.cpp:115,131,142-152) — so that a hit needs one indexed lookup instead of two dependent ones;
	//   * the per-material shading table for kernels that only know the winning primitive's material index.
	struct device_scene
	{
		uint32_t n_spheres, n_planes, n_materials;
		const float* sphere_cx;
		const float* sphere_cy;
		const float* sphere_cz;
		const float* sphere_r;
		const uint32_t* sphere_material;
		const float* plane_nx;
		const float* plane_ny;
		const float* plane_nz;
		const float* plane_d;
		const uint32_t* plane_material;
		const float4* material_shading; // per material: (attenuation.rgb, roughness)
		const uint32_t* material_type;
		// derived per-primitive tables; spheres first, then planes (index n_spheres + i)
		const float4* primitive_geometry; // sphere: (cx, cy, cz, r*r); plane: (nx, ny, nz, d)
		const float4* primitive_shading;  // (attenuation.rgb, roughness) of the primitive's material
		const uint32_t* primitive_scatter; // scatter function of the primitive's material: scatter_lambert / scatter_metal
		// the same two tables under sm_ray_tracer's scatter table (RT_HIP_FLAG_SM_MATERIALS): dielectric, air, vacuum, water
		// and ice become scatter_dielectric, and their shading.w carries the reflectivity (= index of refraction)
		const float4* primitive_shading_sm;
		const uint32_t* primitive_scatter_sm;
		// what only the preview reads (RT_HIP_FLAG_PREVIEW): the boxes as two float4 each — (min corner, material index
		// as bits) and (max corner, 0), corners = center -/+ extents — and the materials' plain albedo
		uint32_t n_boxes;
		const float4* box_bounds;
		const float4* material_albedo;
		// every plane normal is finite with components of at most 2^40 in magnitude (scene.hip; any scene rt's loader makes: it
		// normalises them).  A ray's direction is normalised, so n . d is then NaN, infinite (a degenerate direction) or below 2^60:
		// what lets the scalar-register kernels take a lone plane's 1 / (n . d) without a guard (scan.hpp, test_one_plane)
		uint32_t planes_tame;
	};

	enum : uint32_t
	{
		scatter_lambert = 0,	// mg_ray_tracer.cpp:110-123
		scatter_metal = 1,		// mg_ray_tracer.cpp:126-140
		scatter_dielectric = 2	// sm_ray_tracer.cpp:181-219 (opt-in)
	};

	// Per-frame uniforms (kernel arguments -> SGPRs).
	struct frame_params
	{
		uint32_t width, height;			   // full frame
		uint32_t local_rows;			   // rows this rank renders (compact buffer height in use)
		uint32_t rank, world, stripe_rows; // rt_hip_partition
		uint32_t stripe_shift;			   // log2(stripe_rows) if that is a power of two, else 0xFFFFFFFF (general division)
		uint32_t frame_rows;			   // != 0: the output buffers are the WHOLE frame, a pixel goes to its image row (several
										   // GPUs storing straight into one host frame); 0: this rank's compact stripe buffer
		uint32_t samples_per_pixel, max_bounces;
		uint32_t frame_key_a, frame_key_b;  // the two halves of the mixed 64-bit seed (contract.hpp, random streams)
		float sx, neg_sy;				   // 2/W and -(2/H): ndc = (fma(px, sx, -1), fma(py, neg_sy, 1))
		// inverse view-projection, pre-split for depth 0 / depth 1 (camera.hpp:42-48):
		// row_r(depth) = fma(mx[r], ndc.x, fma(my[r], ndc.y, k[r])),  k_near[r] = fma(M[r][2], 0, M[r][3]),
		// k_far[r] = fma(M[r][2], 1, M[r][3])
		float mx[4], my[4], k_near[4], k_far[4];
		// Contract v4, primary rays.  For a camera built as in camera.hpp:122-137 the last row of the inverse view-projection
		// does not depend on x and y, so w is a per-frame constant, the un-projected near and far points are AFFINE in the
		// pixel position, and every near-to-far line passes through the eye: a PINHOLE.  pinhole != 0 says the matrix is one
		// (decided on the host from binary64 constants, render.hip; the oracle has the same lines) and the kernels then use
		//   base_c   = fma(ray_d1[c], x, fma(ray_d2[c], y, ray_d0[c]))          once per pixel (x, y: its column and row)
		//   toward_c = fma(ray_j1[c], ka, fma(ray_j2[c], kb, base_c))           per sample; (ka, kb) = the jitter's numerators,
		//                                                                        ray_j = ray_d * 2^-24
		//   origin_c = ray_eye[c] + toward_c                                    the near point (ray_d holds kappa * (far - near)
		//                                                                        = near - eye, kappa = near / (far - near))
		// The scalar-register kernels are built for ONE form; mx .. k_far above serve the preview and any other matrix (an
		// orthographic or sheared frustum, a w that varies over the frame: the homogeneous form with one division per sample).
		uint32_t pinhole;
		float ray_d0[3], ray_d1[3], ray_d2[3];
		float ray_j1[3], ray_j2[3];
		float ray_eye[3];
		// eye_form != 0: the matrix is a PERSPECTIVE one (its depth column Z has a finite point E = Z.xyz / Z.w: the eye every
		// near-to-far line passes through) — what the general-camera kernels use for a camera that is not axis-aligned.  With
		// N = the homogeneous near point, N' = N.xyz - E N.w and s = sign(-Z.w), both s N' and s N.w are affine in the pixel
		// position (binary64 constants, like the pinhole's):
		//   base_c = fma(eye_q1[c], x, fma(eye_q2[c], y, eye_q0[c])),   base_w = fma(eye_w1, x, fma(eye_w2, y, eye_w0))    per pixel
		//   t_c    = fma(eye_jq1[c], ka, fma(eye_jq2[c], kb, base_c)),  ws     = fma(eye_jw1, ka, fma(eye_jw2, kb, base_w)) per sample
		//   origin_c = fma(t_c, 1 / ws, eye_e[c]);   toward = t, negated if ws * (ws + eye_zws) < 0   (= N.w F.w < 0)
		// One division per sample and no far point.  A matrix without a finite eye (an orthographic frustum) takes the
		// homogeneous form from mx .. k_far above.
		uint32_t eye_form; // 1 as above; 2: additionally s N.w and s F.w keep one sign and stay deep inside the reciprocal's band over the whole frame (render.hip): no guard, no flip
		float eye_q0[3], eye_q1[3], eye_q2[3];
		float eye_jq1[3], eye_jq2[3];
		float eye_w0, eye_w1, eye_w2, eye_jw1, eye_jw2;
		float eye_e[3];
		float eye_zws;
	};

	// the whole scene of the `small` kernel, passed by value as a kernel argument (-> SGPRs); host copy kept by the context
	constexpr uint32_t scalar_max_spheres = 8; // scenes of up to this many primitives (spheres + planes) run with the scene in SGPRs
	constexpr uint32_t scalar_max_planes = 3;  // ... of which at most this many planes (and at least one sphere)
	struct small_scene
	{
		float4 geometry[scalar_max_spheres]; // spheres (center, radius^2), then planes (normal, d)
		float4 shading[scalar_max_spheres];	 // (attenuation.rgb, roughness or index of refraction) of the primitive's material
		uint32_t scatter[scalar_max_spheres]; // scatter_*
	};

	// pixel sums are taken in chunks of this many consecutive samples (arithmetic contract; see oracle/cpu_ref.cpp)
#ifdef RT_HIP_SAMPLE_CHUNK // (timing experiments only: frames of such a build are not the contract's)
	constexpr uint32_t sample_chunk = RT_HIP_SAMPLE_CHUNK;
#else
	constexpr uint32_t sample_chunk = 16;
#endif

	// Work distribution.  The unit of work is one ITEM = one chunk of 16 consecutive samples of one pixel (K = chunks per
	// pixel).  Small scenes: the frame is cut into pixel tiles of P = 2^pixels_log2 pixels, one tile (P x K items) per wave,
	// launched as a grid of tiles.  Big scenes (tiled / streamed kernels): a persistent launch whose waves draw single
	// items, in blocks of `block_items`, from one launch-wide sequence (device_counters::next_item), pixel-major, bottom
	// row first; a pixel's chunk sums meet in HBM (rolling_buffers).
	struct queue_params
	{
		uint32_t chunks;		   // K = ceil(spp / sample_chunk)
		uint32_t pixels_log2;	   // P (small scenes)
		uint32_t tile_w_log2;	   // a tile is 2^tile_w_log2 columns wide
		uint32_t tiles_x, tiles_y; // tiles across / down this rank's rows
		uint32_t block_items;	   // big scenes: items a wave draws from the launch-wide sequence at a time
		uint32_t lane_cap;		   // big scenes: rays a wave holds at most (64 = all lanes; less in sparse launches of the streamed kernel)
		uint32_t sparse_rays;	   // streamed kernel: a wave holding at most this many rays scans cooperatively
		uint32_t halves;		   // short launches: 1 = the work items are smaller than a chunk (render_queue<.., HALF>): small scenes
								   // half chunks; big scenes item_samples consecutive samples, every sample's value parked
		uint32_t item_samples;	   // big scenes with halves: samples per work item (8, 4, 2 or 1)
	};
	// LDS floats per chunk of a tile: its sum — or, with half-chunks, the first half's partial sum and the second half's 8 x 3 sample values
	constexpr uint32_t half_chunk_slot_floats = 3u + 3u * (sample_chunk / 2u);
	inline size_t tile_slot_bytes(const queue_params& q) // of ONE wave's tile
	{
		return static_cast<size_t>(q.chunks << q.pixels_log2) * (q.halves ? half_chunk_slot_floats : 3u) * sizeof(float);
	}
	// `host_frame`: the packed pixels go to page-locked HOST memory (every row fragment of a tile is a PCIe write)
	// `half_chunks`: 0 = whole chunks (the sm table's kernels have no half-chunk build; RT_HIP_FLAG_FORCE_WHOLE_CHUNKS),
	// 1 = by the size of the launch, 2 = half chunks wherever the samples allow (RT_HIP_FLAG_FORCE_HALF_CHUNKS)
	queue_params choose_queue(uint32_t samples_per_pixel, uint32_t width, uint32_t local_rows, bool big_scene, bool host_frame, int half_chunks, uint32_t primitives, bool sparse_launch);
	constexpr uint32_t sparse_launch_min_spheres = 1024; // the streamed kernel's cooperative scan (a wave with a handful of rays) exists from here
	inline int half_chunk_choice(uint32_t flags)
	{
		if (flags & (RT_HIP_FLAG_SM_MATERIALS | RT_HIP_FLAG_FORCE_WHOLE_CHUNKS))
			return 0;
		return (flags & RT_HIP_FLAG_FORCE_HALF_CHUNKS) ? 2 : 1;
	}
	// internal launch flag, or-ed into the render flags by the callers of launch_render / launch_render_fast: see choose_queue
	constexpr uint32_t launch_flag_host_frame = 1u << 31;

	// What the big-scene kernels exchange chunk sums through (owned by the context, grown on demand):
	//   item_sums   16 bytes per item of this rank's rows (a chunk sum on its way to the lane that folds the pixel);
	//               not needed when a pixel is one chunk.  With sub-chunk items: 16 bytes per SAMPLE
	//   pixel_done  one arrival counter per pixel; zeroed on the launch's own stream in front of every launch (render.hip)
	struct rolling_buffers
	{
		unsigned long long* item_sums = nullptr;
		uint32_t* pixel_done = nullptr;
	};
	// bytes of the two buffers for a launch (0, 0 for the small-scene kernels)
	void rolling_buffer_bytes(const queue_params& queue, uint32_t samples_per_pixel, uint32_t width, uint32_t local_rows, bool big_scene, size_t& item_sums_bytes, size_t& pixel_done_bytes);

	struct device_counters
	{
		// path segments traced, as `segment_counters` partial sums (the host adds them up): every wave adds its count
		// once, and 130 000 atomics on ONE address serialise at about 12 ns each — more than a small launch takes
		static constexpr unsigned segment_counters = 64;
		unsigned long long segments[segment_counters];
		unsigned long long next_item; // head of the item sequence of the big-scene kernels; zeroed before every launch
#ifdef RT_HIP_REGION_COUNTERS
		static constexpr unsigned regions = 13; // experiment variant only (kernels.hip, RT_HIP_REGION)
		unsigned long long region_runs[regions], region_lanes[regions];
#endif
#ifdef RT_HIP_WAVE_CLOCKS
		// experiment variant only (tools/gpu_wave_tail.py): when every wave of a persistent launch started, found the
		// tile queue dry and retired, in ticks of the constant 100 MHz clock (s_memrealtime)
		static constexpr unsigned clocked_waves = 16384;
		unsigned long long wave_clocks[clocked_waves][3];
#endif
	};

#ifndef RT_HIP_RESIDENT_SCALAR_FROM
#define RT_HIP_RESIDENT_SCALAR_FROM 40
#endif
	constexpr uint32_t resident_scalar_scan_from = RT_HIP_RESIDENT_SCALAR_FROM; // spheres from which the resident kernel scans through the scalar cache (kernels.hip)
	constexpr uint32_t resident_max_primitives = 1024; // what the resident kernel keeps in LDS at most: the planes, and the spheres of a scene below resident_scalar_scan_from
	// The launch code prefers the resident kernel (a pixel tile per wave) up to this many primitives, the streamed kernel's rolling
	// items beyond.  From resident_scalar_scan_from spheres on the resident kernel reads the sphere table in memory, as the streamed
	// kernel does, and its LDS holds planes only — so its capacity is no limit to the spheres; what ends its lead is that a tile's
	// lanes run dry one by one while a trip costs the wave a whole scan.  Against the streamed kernel's build for dense frames
	// (1080p x 64 spp, kernel ms): 400 spheres 16.8 against 18.4, 700: 29.7 against 31.0, 1 000: 42.9 against 44.9, 1 100: 47.8 against
	// 48.4, 1 500: 66.5 against 65.8, 2 000: 90.3 against 87.9, 3 000: 141.8 against 130.7 (profiles/r05/resident_vs_dense_streamed.txt;
	// against the streamed kernel as it was before that build the lead lasted to 4 000 spheres: resident_beyond_1024_ab.txt).
	constexpr uint32_t streamed_from_primitives = 1300;
	constexpr uint32_t tile_primitives = 1024;		   // primitives per LDS tile in the tiled kernel

	// what a context remembers between launches: workgroups per CU that stay resident, for the persistent (big-scene) kernels
	struct launch_cache
	{
		struct entry
		{
			size_t lds_bytes = 0;
			int per_cu = 0;
		};
		entry persistent[18]; // { tiled, streamed, streamed for dense frames } x { mg, sm scatter table, fast arithmetic } x { whole chunks, sub-chunk items }
		// one definition for both builds of kernels.hip (the parity contract and RT_HIP_FAST_BUILD), which are linked into
		// one library: the build says which arithmetic it is through the argument (the fast build has no sm scatter table)
		static constexpr unsigned slot(unsigned kernel /* 0 tiled, 1 streamed, 2 streamed for dense frames */, bool sm, bool fast_arithmetic, bool sub_chunk_items)
		{
			return (sub_chunk_items ? 9u : 0u) + 3u * kernel + (fast_arithmetic ? 2u : (sm ? 1u : 0u));
		}
	};

	// which kernel launch_render would pick (RT_HIP_KERNEL_*)
	uint32_t choose_kernel(const device_scene& scene, uint32_t flags, uint32_t samples_per_pixel, bool perspective /* the frame's camera is a pinhole or a plain eye-form one: frame_params::pinhole or eye_form == 2 */,
						   uint64_t pixels /* of this rank's rows */);

	// returns the kernel variant launched (RT_HIP_KERNEL_*)
	uint32_t launch_render(const frame_params& frame,
						   const device_scene& scene,
						   const small_scene& small, // valid when choose_kernel() says RT_HIP_KERNEL_SMALL; tables already chosen by flag
						   uint32_t flags,
						   uint32_t* d_rgba8,
						   float* d_rgb_f32,
						   device_counters* d_counters,
						   const rolling_buffers& rolling, // big scenes: sized by rolling_buffer_bytes(), pixel_done zeroed in front of the launch
						   uint32_t compute_units, // of the device: the big-scene kernels are launched persistent
						   launch_cache& cache,
						   hipStream_t stream);

	// the same with RT_HIP_FLAG_FAST's arithmetic (kernels.hip built with RT_HIP_FAST_BUILD); mg scatter table only
	uint32_t launch_render_fast(const frame_params& frame,
								const device_scene& scene,
								const small_scene& small,
								uint32_t flags,
								uint32_t* d_rgba8,
								float* d_rgb_f32,
								device_counters* d_counters,
								const rolling_buffers& rolling,
								uint32_t compute_units,
								launch_cache& cache,
								hipStream_t stream);

	// RT_HIP_FLAG_PREVIEW: one ray per pixel, reference src/renderers/rasterizer.cpp:24-85
	void launch_preview(const frame_params& frame, const device_scene& scene, uint32_t* d_rgba8, float* d_rgb_f32, device_counters* d_counters, hipStream_t stream);

	// `width` in 32-bit words per row (pixels, or 3 x pixels for the float mean)
	void launch_assemble(uint32_t width,
						 uint32_t height,
						 uint32_t world,
						 uint32_t stripe_rows,
						 uint32_t padded_local_rows,
						 const uint32_t* d_gathered,
						 uint32_t* d_frame,
						 uint32_t first_rank,		// rows of ranks below this one are left alone (already in the frame)
						 bool frame_is_host_memory, // the caller's page-locked back buffer: system-scope stores
						 hipStream_t stream);
}
