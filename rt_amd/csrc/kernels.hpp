// rt_amd/csrc/kernels.hpp — host-visible launch interface of the gfx950 kernels (internal to librt_hip.so).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt_hip
{
	// Scene columns resident in HBM, exactly the soagen columns the reference fills at load
	// (src/scene.cpp:583,595) plus the per-material shading table.
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
		// per material: (albedo.rgb * reflectivity, roughness) — mg_ray_tracer.cpp:115,131 — and the type
		const float4* material_shading;
		const uint32_t* material_type;
	};

	// Per-frame uniforms (kernel arguments -> SGPRs).
	struct frame_params
	{
		uint32_t width, height;			  // full frame
		uint32_t local_rows;			  // rows this rank renders (compact buffer height in use)
		uint32_t rank, world, stripe_rows; // rt_hip_partition
		uint32_t samples_per_pixel, max_bounces;
		uint32_t frame_key;				  // hash of the 64-bit seed
		float sx, neg_sy;				  // 2/W and -(2/H): ndc = (fma(px, sx, -1), fma(py, neg_sy, 1))
		// inverse view-projection, pre-split for depth 0 / depth 1 (camera.hpp:42-48):
		// row_r(depth) = fma(mx[r], ndc.x, fma(my[r], ndc.y, k[r])),  k_near[r] = fma(M[r][2], 0, M[r][3]),
		// k_far[r] = fma(M[r][2], 1, M[r][3])
		float mx[4], my[4], k_near[4], k_far[4];
	};

	struct device_counters
	{
		unsigned long long segments;
	};

	constexpr uint32_t resident_max_primitives = 1024; // spheres + planes kept whole in LDS by the resident kernel
	constexpr uint32_t tile_primitives = 1024;		   // primitives per LDS tile in the tiled kernel

	// returns the kernel variant launched (RT_HIP_KERNEL_*)
	uint32_t launch_render(const frame_params& frame,
						   const device_scene& scene,
						   bool force_tiled,
						   uint32_t* d_rgba8,
						   float* d_rgb_f32,
						   device_counters* d_counters,
						   hipStream_t stream);

	void launch_assemble(uint32_t width,
						 uint32_t height,
						 uint32_t world,
						 uint32_t stripe_rows,
						 uint32_t padded_local_rows,
						 const uint32_t* d_gathered,
						 uint32_t* d_frame,
						 hipStream_t stream);

	void launch_kat_random(uint32_t frame_key, uint32_t pixel, uint32_t sample, uint32_t n, float* d_out, hipStream_t stream);
	void launch_kat_closest_hit(const device_scene& scene,
								uint32_t n,
								const float* d_origins,
								const float* d_directions,
								float* d_distance,
								uint32_t* d_kind,
								uint32_t* d_index,
								float* d_normal,
								hipStream_t stream);
	void launch_kat_sqrt_div(uint32_t n, const float* d_a, const float* d_b, float* d_sqrt, float* d_div, hipStream_t stream);
}
