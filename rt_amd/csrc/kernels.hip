// rt_amd/csrc/kernels.hip — the gfx950 kernels of the mg_ray_tracer path.
//
// What runs here is the whole of reference src/renderers/mg_ray_tracer.cpp:178-205 for one frame: the per-pixel
// worker (:182-201), trace (:155-174), the closest-hit scans (:36-102) and the scatter functions (:110-140).
//
// Execution model (MI355X / CDNA4, 64-wide waves):
//   * one lane owns one pixel and walks ALL of its samples; the per-pixel sum is therefore in sample order,
//     exactly like the reference's `colour += trace(...)` loop (:187-194);
//   * a wave covers an 8x8 pixel tile (lane = 8*row + column) so that its 64 rays stay coherent; a 256-thread
//     workgroup is four such tiles side by side (32x8 pixels) — 8 rows = one stripe of the multi-GPU partition;
//   * the (sample, bounce) double loop is FLATTENED: each loop trip advances every lane by one path segment; a
//     lane whose path ended adds its sample and starts the next one in the same trip.  A wave therefore runs for
//     max-over-lanes of the TOTAL segment count of a pixel (which concentrates around spp x mean path length),
//     not for sum-over-samples of the max-over-lanes path length;
//   * `resident` kernel: all primitives are staged ONCE per workgroup from the SoA columns into LDS as
//     (cx, cy, cz, r^2) / (nx, ny, nz, d) float4s; the scan reads them with wave-uniform addresses, i.e. one
//     broadcast ds_read_b128 per primitive per wave and no bank conflicts;
//   * `tiled` kernel (scenes that do not fit): the same scan, but primitives stream through LDS in tiles of
//     tile_primitives; the workgroup moves in lock step (one segment per trip, barriers around each tile).
// No MFMA: this is intersection arithmetic (subtract/dot/compare/sqrt), not a contraction.
#include "kernels.hpp"
#include "contract.hpp"

#include "../../include/rt_hip.h"

namespace rt_hip
{
	namespace
	{
		constexpr uint32_t block_threads = 256;
		constexpr uint32_t block_pixels_x = 32;
		constexpr uint32_t block_pixels_y = 8;

		// hit_result, mg_ray_tracer.cpp:22-33 (distance < 0 = miss)
		struct hit_result
		{
			float distance;
			vec3 normal;
			uint32_t material;
			uint32_t kind; // 0 none, 1 sphere, 2 plane
			uint32_t index;
		};

		// best candidate of one linear scan (test_planes / test_spheres, mg_ray_tracer.cpp:36-87)
		struct scan_result
		{
			bool have;
			float distance;
			uint32_t index;
		};

		// one step of the scan loop: `if (!hit || *hit < min_hit_dist || (hit_index && hit_dist <= *hit)) continue;`
		__device__ __forceinline__ void scan_accept(scan_result& best, bool hit, float t, uint32_t index)
		{
			if (!hit || t < min_hit_dist || (best.have && best.distance <= t))
				return;
			best.have = true;
			best.index = index;
			best.distance = t;
		}

		__device__ __forceinline__ void scan_spheres(scan_result& best, vec3 o, vec3 d, const float4* lds_spheres, uint32_t count, uint32_t first_index)
		{
			for (uint32_t i = 0; i < count; i++)
			{
				const float4 s = lds_spheres[i]; // wave-uniform address: broadcast read
				float t = 0.0f;
				const bool hit = hits_sphere(o, d, { s.x, s.y, s.z }, s.w, t);
				scan_accept(best, hit, t, first_index + i);
			}
		}

		__device__ __forceinline__ void scan_planes(scan_result& best, vec3 o, vec3 d, const float4* lds_planes, uint32_t count, uint32_t first_index)
		{
			for (uint32_t i = 0; i < count; i++)
			{
				const float4 pl = lds_planes[i];
				float t = 0.0f;
				const bool hit = hits_plane(o, d, { pl.x, pl.y, pl.z }, pl.w, t);
				scan_accept(best, hit, t, first_index + i);
			}
		}

		// cooperative copy of `count` primitives starting at `first` from the SoA columns into float4 LDS slots;
		// lane i of the workgroup reads element first+i of each column (coalesced), radius is squared on the way in
		__device__ __forceinline__ void stage_spheres(float4* lds, const device_scene& s, uint32_t first, uint32_t count)
		{
			for (uint32_t i = threadIdx.x; i < count; i += block_threads)
			{
				const float r = s.sphere_r[first + i];
				lds[i] = make_float4(s.sphere_cx[first + i], s.sphere_cy[first + i], s.sphere_cz[first + i], r * r);
			}
		}

		__device__ __forceinline__ void stage_planes(float4* lds, const device_scene& s, uint32_t first, uint32_t count)
		{
			for (uint32_t i = threadIdx.x; i < count; i += block_threads)
				lds[i] = make_float4(s.plane_nx[first + i], s.plane_ny[first + i], s.plane_nz[first + i], s.plane_d[first + i]);
		}

		// select(test_spheres, test_planes) then select(test_boxes, ...) — mg_ray_tracer.cpp:96-102,160-162.
		// `operator bool` of hit_result is `distance >= 0` (:29-32), which also rejects a NaN distance.
		__device__ __forceinline__ hit_result resolve_hit(const device_scene& s, vec3 o, vec3 d, const scan_result& spheres, const scan_result& planes)
		{
			const float sphere_distance = spheres.have ? spheres.distance : -1.0f;
			const float plane_distance = planes.have ? planes.distance : -1.0f;
			const bool a = sphere_distance >= 0.0f;
			const bool b = plane_distance >= 0.0f;
			hit_result h;
			if (a && (!b || sphere_distance <= plane_distance))
			{
				const uint32_t i = spheres.index;
				const vec3 center = { s.sphere_cx[i], s.sphere_cy[i], s.sphere_cz[i] };
				h.distance = sphere_distance;
				h.normal = normalize(ray_at(o, d, sphere_distance) - center); // vec3::direction(center, r.at(t)), :85
				h.material = s.sphere_material[i];
				h.kind = 1;
				h.index = i;
			}
			else if (b)
			{
				const uint32_t i = planes.index;
				h.distance = plane_distance;
				h.normal = { s.plane_nx[i], s.plane_ny[i], s.plane_nz[i] };
				h.material = s.plane_material[i];
				h.kind = 2;
				h.index = i;
			}
			else
			{
				// note: when a is false and b is false the reference returns b's (negative) distance: a miss
				h.distance = -1.0f;
				h.normal = { 0, 0, 0 };
				h.material = 0;
				h.kind = 0;
				h.index = 0;
			}
			return h;
		}

		// everything a lane carries between loop trips
		struct lane_state
		{
			vec3 origin, dir;	// current ray
			vec3 throughput;	// product of attenuations so far (trace unrolled front to back)
			vec3 colour;		// running sum over samples (:186,193)
			float fx, fy;		// pixel coordinates as floats
			uint32_t pixel_key; // random stream key of the pixel
			uint32_t counter;	// random stream position
			uint32_t sample;	// index of the sample in flight
			uint32_t bounces_left;
			uint32_t segments;
		};

		// worker lambda :189-193 — jittered position, un-project to near and far, build the primary ray
		__device__ __forceinline__ void start_sample(lane_state& st, const frame_params& p)
		{
			st.counter = sample_counter(st.pixel_key, st.sample);
			float jx = 0.5f, jy = 0.5f; // sample 0: pixel centre
			if (st.sample)
			{
				jx = next_random(st.counter);
				jy = next_random(st.counter);
			}
			const float px = st.fx + jx;
			const float py = st.fy + jy;
			const float ndc_x = fma(px, p.sx, -1.0f);
			const float ndc_y = fma(py, p.neg_sy, 1.0f);
			float near_row[4], far_row[4];
#pragma unroll
			for (int r = 0; r < 4; r++)
			{
				near_row[r] = fma(p.mx[r], ndc_x, fma(p.my[r], ndc_y, p.k_near[r]));
				far_row[r] = fma(p.mx[r], ndc_x, fma(p.my[r], ndc_y, p.k_far[r]));
			}
			const float inv_wn = 1.0f / near_row[3];
			const float inv_wf = 1.0f / far_row[3];
			const vec3 near_pos = { near_row[0] * inv_wn, near_row[1] * inv_wn, near_row[2] * inv_wn };
			const vec3 far_pos = { far_row[0] * inv_wf, far_row[1] * inv_wf, far_row[2] * inv_wf };
			st.origin = near_pos;
			st.dir = normalize(far_pos - near_pos);
			st.throughput = { 1.0f, 1.0f, 1.0f };
			st.bounces_left = p.max_bounces;
		}

		// the part of trace() after the closest-hit query (:163-173) for one segment.
		// returns true when the path ended; `contribution` is then the sample's value.
		__device__ __forceinline__ bool shade_segment(lane_state& st, const device_scene& s, const hit_result& hit, vec3& contribution)
		{
			if (!(hit.distance >= 0.0f))
			{
				contribution = st.throughput * sky(st.dir.y);
				return true;
			}
			const float4 shading = s.material_shading[hit.material];
			const bool metal = s.material_type[hit.material] == RT_HIP_MATERIAL_METAL; // everything else is lambert (:142-152)
			const vec3 hit_pos = ray_at(st.origin, st.dir, hit.distance);

			vec3 base = hit.normal;
			float spread = 1.0f;
			if (metal)
			{
				// reflect(normalize(r.direction), n) (:133, common.hpp:100-103)
				const vec3 v = normalize(st.dir);
				const float k = 2.0f * dot(v, hit.normal);
				base = { fma(-k, hit.normal.x, v.x), fma(-k, hit.normal.y, v.y), fma(-k, hit.normal.z, v.z) };
				spread = shading.w;
			}
			const vec3 u = random_unit_vector(st.counter);
			// lambert: n + u (:117) == fma(1, u, n) exactly; metal: reflected + roughness * u (:133-134)
			vec3 scatter = { fma(spread, u.x, base.x), fma(spread, u.y, base.y), fma(spread, u.z, base.z) };
			if (metal)
			{
				if (dot(scatter, hit.normal) <= 0.0f) // absorbed (:135-136)
				{
					contribution = { 0.0f, 0.0f, 0.0f };
					return true;
				}
			}
			else if (__builtin_fabsf(scatter.x) <= approx_zero_epsilon && __builtin_fabsf(scatter.y) <= approx_zero_epsilon
					 && __builtin_fabsf(scatter.z) <= approx_zero_epsilon)
				scatter = hit.normal; // (:118-119)

			st.throughput = st.throughput * vec3{ shading.x, shading.y, shading.z };
			st.origin = hit_pos;
			st.dir = normalize(scatter);
			if (st.bounces_left == 0) // the next trace() call would return {} at :157-158
			{
				contribution = { 0.0f, 0.0f, 0.0f };
				return true;
			}
			return false;
		}

		__device__ __forceinline__ uint32_t global_row(uint32_t local_row, const frame_params& p)
		{
			return ((local_row / p.stripe_rows) * p.world + p.rank) * p.stripe_rows + (local_row % p.stripe_rows);
		}

		// pixel owned by this lane; false if outside this rank's part of the frame
		__device__ __forceinline__ bool lane_pixel(const frame_params& p, uint32_t& lx, uint32_t& ly)
		{
			const uint32_t lane = threadIdx.x & 63u;
			const uint32_t wave = threadIdx.x >> 6;
			lx = blockIdx.x * block_pixels_x + wave * 8u + (lane & 7u);
			ly = blockIdx.y * block_pixels_y + (lane >> 3);
			return lx < p.width && ly < p.local_rows;
		}

		__device__ __forceinline__ void init_lane(lane_state& st, const frame_params& p, uint32_t lx, uint32_t ly)
		{
			const uint32_t gy = global_row(ly, p);
			st.fx = static_cast<float>(lx);
			st.fy = static_cast<float>(gy);
			st.pixel_key = pixel_key(p.frame_key, gy * p.width + lx); // image_view::position_of, image.hpp:155-159
			st.colour = { 0.0f, 0.0f, 0.0f };
			st.sample = 0;
			st.segments = 0;
		}

		// :195-200 — mean, sqrt "gamma", pack, store
		__device__ __forceinline__ void finish_pixel(const lane_state& st, const frame_params& p, uint32_t lx, uint32_t ly, uint32_t* out_rgba, float* out_rgb)
		{
			const float n = static_cast<float>(p.samples_per_pixel);
			const vec3 mean = { st.colour.x / n, st.colour.y / n, st.colour.z / n };
			const size_t o = static_cast<size_t>(ly) * p.width + lx;
			if (out_rgb)
			{
				out_rgb[o * 3 + 0] = mean.x;
				out_rgb[o * 3 + 1] = mean.y;
				out_rgb[o * 3 + 2] = mean.z;
			}
			out_rgba[o] = pack_rgba8888({ __builtin_sqrtf(mean.x), __builtin_sqrtf(mean.y), __builtin_sqrtf(mean.z) });
		}

		__device__ __forceinline__ void add_segments(device_counters* counters, uint32_t segments)
		{
			unsigned long long total = segments;
#pragma unroll
			for (int offset = 32; offset > 0; offset >>= 1)
				total += __shfl_down(total, offset, 64);
			if ((threadIdx.x & 63u) == 0 && total)
				atomicAdd(&counters->segments, total);
		}

		// ---- resident kernel ------------------------------------------------------------------------------------
		__global__ __launch_bounds__(block_threads) void render_resident(const frame_params p,
																		 const device_scene s,
																		 uint32_t* __restrict__ out_rgba,
																		 float* __restrict__ out_rgb,
																		 device_counters* __restrict__ counters)
		{
			extern __shared__ float4 lds[];
			float4* const lds_spheres = lds;
			float4* const lds_planes = lds + s.n_spheres;
			stage_spheres(lds_spheres, s, 0, s.n_spheres);
			stage_planes(lds_planes, s, 0, s.n_planes);
			__syncthreads();

			uint32_t lx, ly;
			lane_state st;
			st.segments = 0;
			if (lane_pixel(p, lx, ly))
			{
				init_lane(st, p, lx, ly);
				start_sample(st, p);
				while (true)
				{
					st.bounces_left--;
					st.segments++;
					scan_result planes = { false, 0.0f, 0 };
					scan_result spheres = { false, 0.0f, 0 };
					scan_planes(planes, st.origin, st.dir, lds_planes, s.n_planes, 0);
					scan_spheres(spheres, st.origin, st.dir, lds_spheres, s.n_spheres, 0);
					const hit_result hit = resolve_hit(s, st.origin, st.dir, spheres, planes);
					vec3 contribution;
					if (shade_segment(st, s, hit, contribution))
					{
						st.colour = st.colour + contribution;
						if (++st.sample >= p.samples_per_pixel)
							break;
						start_sample(st, p);
					}
				}
				finish_pixel(st, p, lx, ly, out_rgba, out_rgb);
			}
			add_segments(counters, st.segments);
		}

		// ---- tiled kernel -----------------------------------------------------------------------------------------
		__global__ __launch_bounds__(block_threads) void render_tiled(const frame_params p,
																	  const device_scene s,
																	  uint32_t* __restrict__ out_rgba,
																	  float* __restrict__ out_rgb,
																	  device_counters* __restrict__ counters)
		{
			__shared__ float4 tile[tile_primitives];

			uint32_t lx, ly;
			lane_state st;
			st.segments = 0;
			bool alive = lane_pixel(p, lx, ly);
			if (alive)
			{
				init_lane(st, p, lx, ly);
				start_sample(st, p);
			}

			while (__syncthreads_or(alive))
			{
				if (alive)
				{
					st.bounces_left--;
					st.segments++;
				}
				scan_result planes = { false, 0.0f, 0 };
				scan_result spheres = { false, 0.0f, 0 };
				for (uint32_t first = 0; first < s.n_planes; first += tile_primitives)
				{
					const uint32_t count = min(tile_primitives, s.n_planes - first);
					__syncthreads(); // previous tile fully consumed
					stage_planes(tile, s, first, count);
					__syncthreads();
					if (alive)
						scan_planes(planes, st.origin, st.dir, tile, count, first);
				}
				for (uint32_t first = 0; first < s.n_spheres; first += tile_primitives)
				{
					const uint32_t count = min(tile_primitives, s.n_spheres - first);
					__syncthreads();
					stage_spheres(tile, s, first, count);
					__syncthreads();
					if (alive)
						scan_spheres(spheres, st.origin, st.dir, tile, count, first);
				}
				if (alive)
				{
					const hit_result hit = resolve_hit(s, st.origin, st.dir, spheres, planes);
					vec3 contribution;
					if (shade_segment(st, s, hit, contribution))
					{
						st.colour = st.colour + contribution;
						if (++st.sample >= p.samples_per_pixel)
						{
							alive = false;
							finish_pixel(st, p, lx, ly, out_rgba, out_rgb);
						}
						else
							start_sample(st, p);
					}
				}
			}
			add_segments(counters, st.segments);
		}

		// ---- multi-GPU assemble: rank-major compact stripes -> frame ------------------------------------------------
		__global__ __launch_bounds__(block_threads) void assemble_stripes(uint32_t width,
																		  uint32_t height,
																		  uint32_t world,
																		  uint32_t stripe_rows,
																		  uint32_t padded_local_rows,
																		  const uint32_t* __restrict__ gathered,
																		  uint32_t* __restrict__ frame)
		{
			const uint32_t x = blockIdx.x * block_threads + threadIdx.x;
			const uint32_t y = blockIdx.y;
			if (x >= width || y >= height)
				return;
			const uint32_t stripe = y / stripe_rows;
			const uint32_t rank = stripe % world;
			const uint32_t local_row = (stripe / world) * stripe_rows + (y % stripe_rows);
			frame[static_cast<size_t>(y) * width + x] = gathered[(static_cast<size_t>(rank) * padded_local_rows + local_row) * width + x];
		}

		// ---- known-answer kernels -----------------------------------------------------------------------------------
		__global__ void kat_random(uint32_t frame, uint32_t pixel, uint32_t sample, uint32_t n, float* out)
		{
			if (blockIdx.x || threadIdx.x)
				return;
			uint32_t counter = sample_counter(pixel_key(frame, pixel), sample);
			for (uint32_t i = 0; i < n; i++)
				out[i] = next_random(counter);
		}

		__global__ __launch_bounds__(block_threads) void kat_closest_hit(const device_scene s,
																		 uint32_t n,
																		 const float* __restrict__ origins,
																		 const float* __restrict__ directions,
																		 float* __restrict__ out_distance,
																		 uint32_t* __restrict__ out_kind,
																		 uint32_t* __restrict__ out_index,
																		 float* __restrict__ out_normal)
		{
			__shared__ float4 tile[tile_primitives];
			const uint32_t i = blockIdx.x * block_threads + threadIdx.x;
			const bool alive = i < n;
			vec3 o = { 0, 0, 0 }, d = { 0, 0, 1 };
			if (alive)
			{
				o = { origins[i * 3], origins[i * 3 + 1], origins[i * 3 + 2] };
				d = { directions[i * 3], directions[i * 3 + 1], directions[i * 3 + 2] };
			}
			scan_result planes = { false, 0.0f, 0 };
			scan_result spheres = { false, 0.0f, 0 };
			for (uint32_t first = 0; first < s.n_planes; first += tile_primitives)
			{
				const uint32_t count = min(tile_primitives, s.n_planes - first);
				__syncthreads();
				stage_planes(tile, s, first, count);
				__syncthreads();
				if (alive)
					scan_planes(planes, o, d, tile, count, first);
			}
			for (uint32_t first = 0; first < s.n_spheres; first += tile_primitives)
			{
				const uint32_t count = min(tile_primitives, s.n_spheres - first);
				__syncthreads();
				stage_spheres(tile, s, first, count);
				__syncthreads();
				if (alive)
					scan_spheres(spheres, o, d, tile, count, first);
			}
			if (alive)
			{
				const hit_result h = resolve_hit(s, o, d, spheres, planes);
				out_distance[i] = h.distance;
				out_kind[i] = h.kind;
				out_index[i] = h.index;
				out_normal[i * 3 + 0] = h.normal.x;
				out_normal[i * 3 + 1] = h.normal.y;
				out_normal[i * 3 + 2] = h.normal.z;
			}
		}

		__global__ void kat_sqrt_div(uint32_t n, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out_sqrt, float* __restrict__ out_div)
		{
			const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
			if (i < n)
			{
				out_sqrt[i] = __builtin_sqrtf(a[i]);
				out_div[i] = a[i] / b[i];
			}
		}
	}

	uint32_t launch_render(const frame_params& frame,
						   const device_scene& scene,
						   bool force_tiled,
						   uint32_t* d_rgba8,
						   float* d_rgb_f32,
						   device_counters* d_counters,
						   hipStream_t stream)
	{
		const dim3 grid((frame.width + block_pixels_x - 1) / block_pixels_x, (frame.local_rows + block_pixels_y - 1) / block_pixels_y);
		const dim3 block(block_threads);
		if (!grid.x || !grid.y)
			return RT_HIP_KERNEL_NONE;
		const uint32_t primitives = scene.n_spheres + scene.n_planes;
		if (!force_tiled && primitives <= resident_max_primitives)
		{
			const size_t lds_bytes = static_cast<size_t>(primitives ? primitives : 1u) * sizeof(float4);
			hipLaunchKernelGGL(render_resident, grid, block, lds_bytes, stream, frame, scene, d_rgba8, d_rgb_f32, d_counters);
			return RT_HIP_KERNEL_RESIDENT;
		}
		hipLaunchKernelGGL(render_tiled, grid, block, 0, stream, frame, scene, d_rgba8, d_rgb_f32, d_counters);
		return RT_HIP_KERNEL_TILED;
	}

	void launch_assemble(uint32_t width,
						 uint32_t height,
						 uint32_t world,
						 uint32_t stripe_rows,
						 uint32_t padded_local_rows,
						 const uint32_t* d_gathered,
						 uint32_t* d_frame,
						 hipStream_t stream)
	{
		const dim3 grid((width + block_threads - 1) / block_threads, height);
		hipLaunchKernelGGL(assemble_stripes, grid, dim3(block_threads), 0, stream, width, height, world, stripe_rows, padded_local_rows, d_gathered, d_frame);
	}

	void launch_kat_random(uint32_t frame, uint32_t pixel, uint32_t sample, uint32_t n, float* d_out, hipStream_t stream)
	{
		hipLaunchKernelGGL(kat_random, dim3(1), dim3(64), 0, stream, frame, pixel, sample, n, d_out);
	}

	void launch_kat_closest_hit(const device_scene& scene,
								uint32_t n,
								const float* d_origins,
								const float* d_directions,
								float* d_distance,
								uint32_t* d_kind,
								uint32_t* d_index,
								float* d_normal,
								hipStream_t stream)
	{
		const dim3 grid((n + block_threads - 1) / block_threads);
		hipLaunchKernelGGL(kat_closest_hit, grid, dim3(block_threads), 0, stream, scene, n, d_origins, d_directions, d_distance, d_kind, d_index, d_normal);
	}

	void launch_kat_sqrt_div(uint32_t n, const float* d_a, const float* d_b, float* d_sqrt, float* d_div, hipStream_t stream)
	{
		const dim3 grid((n + block_threads - 1) / block_threads);
		hipLaunchKernelGGL(kat_sqrt_div, grid, dim3(block_threads), 0, stream, n, d_a, d_b, d_sqrt, d_div);
	}
}
