// tools/scan_bench.hip — micro-benchmark of the closest-hit scan over an LDS tile of spheres (the inner loop of the tiled
// kernel): cycles per wave-test for several loop shapes.  Same arithmetic as rt_amd/csrc/kernels.hip's probe_sphere.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/scan_bench tools/scan_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <cmath>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct vec3 { float x, y, z; };
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float dot(vec3 a, vec3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }

struct probe { float a, e2, disc; bool pos; };
__device__ __forceinline__ probe probe_sphere(vec3 o, vec3 d, float4 s)
{
	const vec3 e = { s.x - o.x, s.y - o.y, s.z - o.z };
	probe p;
	p.a = dot(e, d);
	p.e2 = dot(e, e);
	p.disc = s.w - fma_(-p.a, p.a, p.e2);
	p.pos = !(p.disc < 0.0f);
	return p;
}
struct cand { float t; uint32_t index; bool have; };
__device__ __forceinline__ void finish(cand& best, const probe& p, float r2, uint32_t index, unsigned long long lanes)
{
	if (lanes != 0)
	{
		const float f = __builtin_sqrtf(p.pos ? p.disc : 1.0f);
		const float t = (p.e2 < r2) ? p.a + f : p.a - f;
		const bool accept = p.pos && !(t < 0.001f) && !(best.have && best.t <= t);
		best.t = accept ? t : best.t;
		best.index = accept ? index : best.index;
		best.have = best.have || accept;
	}
}

constexpr uint32_t tile = 1024;

// V0: the current shape — groups of four, one branch per group
__device__ __forceinline__ void scan_v0(cand& best, vec3 o, vec3 d, const float4* lds, uint32_t base)
{
	for (uint32_t i = 0; i < tile; i += 4)
	{
		const float4 s0 = lds[i], s1 = lds[i + 1], s2 = lds[i + 2], s3 = lds[i + 3];
		const probe p0 = probe_sphere(o, d, s0), p1 = probe_sphere(o, d, s1), p2 = probe_sphere(o, d, s2), p3 = probe_sphere(o, d, s3);
		const unsigned long long m0 = __builtin_amdgcn_ballot_w64(p0.pos), m1 = __builtin_amdgcn_ballot_w64(p1.pos);
		const unsigned long long m2 = __builtin_amdgcn_ballot_w64(p2.pos), m3 = __builtin_amdgcn_ballot_w64(p3.pos);
		if ((m0 | m1 | m2 | m3) != 0)
		{
			finish(best, p0, s0.w, base + i, m0);
			finish(best, p1, s1.w, base + i + 1, m1);
			finish(best, p2, s2.w, base + i + 2, m2);
			finish(best, p3, s3.w, base + i + 3, m3);
		}
	}
}

// V1: a first pass over a BLOCK of spheres that only collects "some lane may hit" (no branch inside), then the V0 code
// over the block if anything showed up
template <uint32_t BLOCK>
__device__ __forceinline__ void scan_v1(cand& best, vec3 o, vec3 d, const float4* lds, uint32_t base)
{
	for (uint32_t b = 0; b < tile; b += BLOCK)
	{
		bool any = false;
#pragma unroll
		for (uint32_t i = 0; i < BLOCK; i++)
			any = any || probe_sphere(o, d, lds[b + i]).pos;
		if (__builtin_amdgcn_ballot_w64(any) != 0)
		{
			for (uint32_t i = b; i < b + BLOCK; i += 4)
			{
				const float4 s0 = lds[i], s1 = lds[i + 1], s2 = lds[i + 2], s3 = lds[i + 3];
				const probe p0 = probe_sphere(o, d, s0), p1 = probe_sphere(o, d, s1), p2 = probe_sphere(o, d, s2), p3 = probe_sphere(o, d, s3);
				const unsigned long long m0 = __builtin_amdgcn_ballot_w64(p0.pos), m1 = __builtin_amdgcn_ballot_w64(p1.pos);
				const unsigned long long m2 = __builtin_amdgcn_ballot_w64(p2.pos), m3 = __builtin_amdgcn_ballot_w64(p3.pos);
				if ((m0 | m1 | m2 | m3) != 0)
				{
					finish(best, p0, s0.w, base + i, m0);
					finish(best, p1, s1.w, base + i + 1, m1);
					finish(best, p2, s2.w, base + i + 2, m2);
					finish(best, p3, s3.w, base + i + 3, m3);
				}
			}
		}
	}
}

// V2: like V1, but the first pass keeps the largest discriminant instead of a flag (max is one instruction; the
// comparison happens once per block)
template <uint32_t BLOCK>
__device__ __forceinline__ void scan_v2(cand& best, vec3 o, vec3 d, const float4* lds, uint32_t base)
{
	for (uint32_t b = 0; b < tile; b += BLOCK)
	{
		float top = -1.0f;
#pragma unroll
		for (uint32_t i = 0; i < BLOCK; i++)
			top = __builtin_fmaxf(top, probe_sphere(o, d, lds[b + i]).disc);
		if (__builtin_amdgcn_ballot_w64(!(top < 0.0f)) != 0)
		{
			for (uint32_t i = b; i < b + BLOCK; i += 4)
			{
				const float4 s0 = lds[i], s1 = lds[i + 1], s2 = lds[i + 2], s3 = lds[i + 3];
				const probe p0 = probe_sphere(o, d, s0), p1 = probe_sphere(o, d, s1), p2 = probe_sphere(o, d, s2), p3 = probe_sphere(o, d, s3);
				const unsigned long long m0 = __builtin_amdgcn_ballot_w64(p0.pos), m1 = __builtin_amdgcn_ballot_w64(p1.pos);
				const unsigned long long m2 = __builtin_amdgcn_ballot_w64(p2.pos), m3 = __builtin_amdgcn_ballot_w64(p3.pos);
				if ((m0 | m1 | m2 | m3) != 0)
				{
					finish(best, p0, s0.w, base + i, m0);
					finish(best, p1, s1.w, base + i + 1, m1);
					finish(best, p2, s2.w, base + i + 2, m2);
					finish(best, p3, s3.w, base + i + 3, m3);
				}
			}
		}
	}
}

template <int V>
__global__ __launch_bounds__(256, 5) void bench(const float4* __restrict__ spheres, uint32_t n_tiles, uint32_t repeats, const float* __restrict__ rays, float* __restrict__ out_t, uint32_t* __restrict__ out_index)
{
	__shared__ float4 lds[tile];
	const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
	const vec3 o = { rays[tid * 6 + 0], rays[tid * 6 + 1], rays[tid * 6 + 2] };
	const vec3 d = { rays[tid * 6 + 3], rays[tid * 6 + 4], rays[tid * 6 + 5] };
	cand best = { 0.0f, 0u, false };
	for (uint32_t r = 0; r < repeats; r++)
		for (uint32_t t = 0; t < n_tiles; t++)
		{
			__syncthreads();
			for (uint32_t i = threadIdx.x; i < tile; i += 256)
				lds[i] = spheres[t * tile + i];
			__syncthreads();
			if (V == 0) scan_v0(best, o, d, lds, t * tile);
			if (V == 1) scan_v1<16>(best, o, d, lds, t * tile);
			if (V == 2) scan_v1<32>(best, o, d, lds, t * tile);
			if (V == 3) scan_v2<16>(best, o, d, lds, t * tile);
			if (V == 4) scan_v2<32>(best, o, d, lds, t * tile);
			if (V == 5) scan_v2<64>(best, o, d, lds, t * tile);
		}
	out_t[tid] = best.have ? best.t : -1.0f;
	out_index[tid] = best.have ? best.index : 0xFFFFFFFFu;
}

static uint64_t state = 20250310;
static double u01() { state += 0x9E3779B97F4A7C15ull; uint64_t z = state; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31; return (z >> 40) * (1.0 / 16777216.0); }

int main()
{
	const uint32_t n_tiles = 96, n_spheres = n_tiles * tile; // ~100 k small spheres on the ground, like synthetic-100k
	std::vector<float4> spheres(n_spheres);
	for (auto& s : spheres)
	{
		const float r = 0.05f + 0.20f * static_cast<float>(u01());
		s = make_float4(-40.0f + 80.0f * static_cast<float>(u01()), r, -2.0f - 78.0f * static_cast<float>(u01()), r * r);
	}
	const uint32_t n_rays = 256 * 1280 * 2; // two workgroups per resident slot
	std::vector<float> rays(n_rays * 6);
	for (uint32_t i = 0; i < n_rays; i++)
	{
		// half primary-like rays from the camera, half bounce rays from the ground into the positive octant
		const bool primary = (i / 64) % 2 == 0;
		float o[3], d[3];
		if (primary)
		{
			o[0] = 0, o[1] = 6, o[2] = 3;
			d[0] = static_cast<float>(u01() - 0.5), d[1] = static_cast<float>(-0.35 - 0.3 * u01()), d[2] = -1;
		}
		else
		{
			o[0] = static_cast<float>(-30 + 60 * u01()), o[1] = 0.001f, o[2] = static_cast<float>(-5 - 60 * u01());
			d[0] = static_cast<float>(u01()), d[1] = static_cast<float>(u01()), d[2] = static_cast<float>(u01());
		}
		const float len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
		for (int k = 0; k < 3; k++)
			rays[i * 6 + k] = o[k], rays[i * 6 + 3 + k] = d[k] / len;
	}
	float4* d_spheres; float* d_rays; float* d_t; uint32_t* d_index;
	CHECK(hipMalloc(&d_spheres, spheres.size() * sizeof(float4)));
	CHECK(hipMalloc(&d_rays, rays.size() * sizeof(float)));
	CHECK(hipMalloc(&d_t, n_rays * sizeof(float)));
	CHECK(hipMalloc(&d_index, n_rays * sizeof(uint32_t)));
	CHECK(hipMemcpy(d_spheres, spheres.data(), spheres.size() * sizeof(float4), hipMemcpyHostToDevice));
	CHECK(hipMemcpy(d_rays, rays.data(), rays.size() * sizeof(float), hipMemcpyHostToDevice));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	std::vector<uint32_t> reference;
	const char* names[6] = { "V0 groups of 4, branch per group", "V1 flag pass over 16, then V0 on the block", "V1 flag pass over 32", "V2 max-disc pass over 16", "V2 max-disc pass over 32", "V2 max-disc pass over 64" };
	const uint32_t repeats = 4;
	for (int v = 0; v < 6; v++)
	{
		float best_ms = 1e30f;
		for (int rep = 0; rep < 3; rep++)
		{
			CHECK(hipEventRecord(e0));
			const dim3 grid(n_rays / 256), block(256);
			switch (v)
			{
				case 0: hipLaunchKernelGGL(bench<0>, grid, block, 0, 0, d_spheres, n_tiles, repeats, d_rays, d_t, d_index); break;
				case 1: hipLaunchKernelGGL(bench<1>, grid, block, 0, 0, d_spheres, n_tiles, repeats, d_rays, d_t, d_index); break;
				case 2: hipLaunchKernelGGL(bench<2>, grid, block, 0, 0, d_spheres, n_tiles, repeats, d_rays, d_t, d_index); break;
				case 3: hipLaunchKernelGGL(bench<3>, grid, block, 0, 0, d_spheres, n_tiles, repeats, d_rays, d_t, d_index); break;
				case 4: hipLaunchKernelGGL(bench<4>, grid, block, 0, 0, d_spheres, n_tiles, repeats, d_rays, d_t, d_index); break;
				default: hipLaunchKernelGGL(bench<5>, grid, block, 0, 0, d_spheres, n_tiles, repeats, d_rays, d_t, d_index); break;
			}
			CHECK(hipEventRecord(e1));
			CHECK(hipEventSynchronize(e1));
			float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
			best_ms = ms < best_ms ? ms : best_ms;
		}
		std::vector<uint32_t> index(n_rays);
		CHECK(hipMemcpy(index.data(), d_index, n_rays * sizeof(uint32_t), hipMemcpyDeviceToHost));
		if (v == 0) reference = index;
		size_t differ = 0, hits = 0;
		for (uint32_t i = 0; i < n_rays; i++) differ += index[i] != reference[i], hits += index[i] != 0xFFFFFFFFu;
		const double wave_tests = static_cast<double>(n_rays / 64) * n_spheres * repeats;
		// 1024 SIMDs at 2.1 GHz nominal
		std::printf("%-44s %8.2f ms  %6.1f SIMD-cycles per wave-test  hits %zu  differs-from-V0 %zu\n", names[v], best_ms, best_ms * 1e-3 * 1024 * 2.1e9 / wave_tests, hits, differ);
	}
	return 0;
}
