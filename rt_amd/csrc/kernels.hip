// rt_amd/csrc/kernels.hip — the gfx950 kernels of the mg_ray_tracer path.
//
// What runs here is the whole of reference src/renderers/mg_ray_tracer.cpp:178-205 for one frame: the per-pixel
// worker (:182-201), trace (:155-174), the closest-hit scans (:36-102) and the scatter functions (:110-140).
//
// Execution model (MI355X / CDNA4, 64-wide waves, 256-thread workgroups = 4 waves):
//   * a lane traces one ray at a time; the unit of work is one CHUNK of 16 consecutive samples of one pixel (the
//     arithmetic contract sums a pixel chunk-wise: each chunk in sample order, chunk sums in chunk order);
//   * a wave owns a small pixel tile (2x2 ... 8x8) and works through a queue of its (pixel, chunk) items: a lane pulls
//     the next item when it finishes one (ballot + mbcnt, no atomics), parks the chunk sum in a wave-private LDS slot,
//     and at the end the pixels' slots are folded in chunk order and stored.  Short items keep all lanes busy and
//     keep waves short, so that a launch has tens of thousands of waves to balance over the 1024 SIMDs;
//   * the (sample, bounce) double loop is FLATTENED: each loop trip advances every lane by one path segment.  A trip is
//     closest-hit query -> a miss ends the sample (sky) -> free lanes pull items -> ONE fused tail for "scatter a hit"
//     and "start the next sample" (both consume random draws and end in the normalisation of a direction: issued
//     once for the wave; only what really differs runs under its own lane mask);
//   * the closest-hit scan reads every primitive with a WAVE-UNIFORM operand and skips a primitive's square-root half
//     when no lane of the wave can hit it (ballots combined on the scalar unit);
//   * one kernel template, its modes chosen by scene size:
//       small    (<= 8 spheres, no planes: basic.toml, dielectric.toml): the spheres arrive as kernel arguments and
//                live in SGPRs; the scan is fully unrolled straight-line code with scalar operands; only the
//                per-lane lookups of the winning sphere go through LDS;
//       resident (<= 1024 primitives): all primitives are staged ONCE per workgroup into LDS as float4s; the scan
//                reads them as broadcast ds_read_b128 (no bank conflicts), four spheres per group;
//       tiled    (anything larger): primitives stream from the SoA columns in HBM/L2 through one LDS tile of 1024
//                shared by the workgroup's waves (coalesced dword loads per column, radius squared on the way in);
//                the workgroup moves in lock step, one path segment per trip, with barriers around each tile;
//       streamed (larger scenes from 32 samples per pixel): the resident loop reading the primitive table with
//                wave-uniform scalar loads from HBM/L2, the next group's load issued before this group's probes; a
//                wave left with a handful of rays walks the table once per ray with all 64 lanes instead.
//     The two big-scene modes are launched persistent and draw single (pixel, chunk) items from ONE launch-wide
//     sequence; a pixel's chunk sums meet in HBM ("rolling items", see render_queue): there a trip costs one scan
//     over all primitives whatever the number of lanes holding a ray, so lane occupancy is everything — and so is
//     that the whole launch runs dry at one moment.
// No MFMA: this is intersection arithmetic (subtract / dot / compare / sqrt), not a contraction.
//
// This file is compiled TWICE into librt_hip.so: as it stands (the parity contract: -ffp-contract=off, exactly rounded
// square roots and quotients), and with -DRT_HIP_FAST_BUILD -ffp-contract=fast (RT_HIP_FLAG_FAST: contract.hpp's
// hardware approximations), where it provides launch_render_fast() and nothing else.
#include "kernels.hpp"
#include "contract.hpp"
#include "scan.hpp"

#ifdef RT_HIP_FAST_BUILD
#define render_queue render_queue_fast
#define launch_render launch_render_fast
#endif

#include "../../include/rt_hip.h"

#include <algorithm>
#include <cstdlib>

// waves per SIMD the kernel variants are compiled for (register budget = 512 / waves): measured, see render_queue
#ifndef RT_HIP_WAVES_FEW
#define RT_HIP_WAVES_FEW 7 // scalar-register kernels in general
#endif
// ... and those that hold six or more spheres and seven or more primitives (7 + 0, 8 + 0, 6 + 1, 7 + 1, 6 + 2): 80 registers instead of
// 72 spare them most of their scratch, which costs more than the seventh wave brings — every (spheres, planes) build at 7 and at 6 waves:
// profiles/r05/small_sweep.txt (round 4 had settled the same question build by build: waves_full_ab.txt)
#ifndef RT_HIP_WAVES_MANY
#define RT_HIP_WAVES_MANY 6
#endif
#ifndef RT_HIP_WAVES_RESIDENT
#define RT_HIP_WAVES_RESIDENT 7
#endif
// the streamed kernel of frames that fill the device (NS == -3: no cooperative scan of sparse waves, whose four 16-byte loads in flight
// per lane set the register count of the other build): 80 registers hold its loop, half-chunk items included — config 5 5.47 s at 5
// waves WITH the cooperative scan, 5.09 without, 4.99 at 6 waves, 5.05 at 7 (profiles/r05/streamed_occupancy_ab.txt)
#ifndef RT_HIP_WAVES_DENSE
#define RT_HIP_WAVES_DENSE 6
#endif
#ifndef RT_HIP_PERSISTENT_WAVES_CAP
#define RT_HIP_PERSISTENT_WAVES_CAP 5 // workgroups per CU of the persistent (big-scene) launches: see launch_queue_sm
#endif

// Region counters (tools/region_profile.py; built only as an experiment variant, never in the product): how often each
// part of a loop trip runs and with how many lanes — the dynamic side of profiles/r02/isa_trip_breakdown.txt.
#ifdef RT_HIP_REGION_COUNTERS
#define RT_HIP_REGION(i)                                                                                               \
	do                                                                                                                 \
	{                                                                                                                  \
		region_runs[i] += 1u;                                                                                          \
		region_lanes[i] += static_cast<uint32_t>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(true)));             \
	}                                                                                                                  \
	while (false)
#else
#define RT_HIP_REGION(i) ((void)0)
#endif

namespace rt_hip
{
	namespace
	{
		// The streamed kernel's sphere scan: the (center, radius^2) table is read from HBM/L2 with wave-uniform scalar loads,
		// four spheres = 64 bytes = one s_load_dwordx16.  Every wave of a CU walks the 1.6 MB table at its own position, so
		// the 16 KB scalar cache holds next to nothing of it (SQC_DCACHE: 29 % hits at 100 000 spheres,
		// profiles/r03/config5_streamed/) and most loads come from L2.  The load of group g+1 is therefore issued BEFORE
		// group g is probed (two register sets, the loop unrolled by two so that no set is ever copied): the wave has 48
		// vector instructions to issue while its next 64 bytes are on their way.
		__device__ __forceinline__ void probe_group(candidate& best, vec3 o, vec3 d, float4 s0, float4 s1, float4 s2, float4 s3, uint32_t first_index)
		{
			const sphere_probe p0 = probe_sphere(o, d, s0);
			const sphere_probe p1 = probe_sphere(o, d, s1);
			const sphere_probe p2 = probe_sphere(o, d, s2);
			const sphere_probe p3 = probe_sphere(o, d, s3);
			const unsigned long long m0 = __builtin_amdgcn_ballot_w64(p0.pos), m1 = __builtin_amdgcn_ballot_w64(p1.pos);
			const unsigned long long m2 = __builtin_amdgcn_ballot_w64(p2.pos), m3 = __builtin_amdgcn_ballot_w64(p3.pos);
			if ((m0 | m1 | m2 | m3) != 0)
			{
				finish_sphere(best, p0, s0.w, first_index, m0);
				finish_sphere(best, p1, s1.w, first_index + 1, m1);
				finish_sphere(best, p2, s2.w, first_index + 2, m2);
				finish_sphere(best, p3, s3.w, first_index + 3, m3);
			}
		}

		__device__ __forceinline__ void scan_streamed_spheres(candidate& best, vec3 o, vec3 d, const float4* __restrict__ table, uint32_t count)
		{
			const uint32_t groups = count >> 2;
			if (groups)
			{
				float4 a0 = table[0], a1 = table[1], a2 = table[2], a3 = table[3];
				uint32_t g = 0;
				for (; g + 2 <= groups; g += 2)
				{
					const float4* const b = table + (g + 1) * 4;
					const float4 b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
					probe_group(best, o, d, a0, a1, a2, a3, g * 4);
					const float4* const a = table + (g + 2 < groups ? g + 2 : groups - 1) * 4; // (past the end: the last group again, unused)
					a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
					probe_group(best, o, d, b0, b1, b2, b3, (g + 1) * 4);
				}
				if (g < groups)
					probe_group(best, o, d, a0, a1, a2, a3, g * 4);
			}
			for (uint32_t i = groups * 4; i < count; i++)
				test_sphere(best, o, d, table[i], i);
		}

		// SPARSE WAVES (streamed kernel).  A scan costs a wave the same whether one lane holds a ray or all 64 do.  At the
		// end of a launch — and in small frames of big scenes — waves hold a handful of rays: such a wave turns the scan
		// around and walks the table ONCE PER RAY with all 64 lanes, lane l testing spheres l, l + 64, ... (coalesced 16-byte
		// loads), followed by a wave-wide minimum.  test_spheres keeps the first sphere at the smallest accepted distance
		// (`hit_dist <= *hit` -> continue, mg_ray_tracer.cpp:74), i.e. the lexicographic minimum of (t, index) over the
		// accepted candidates; every lane takes that minimum over its own spheres in index order, and the 64 partial results
		// are reduced with the same order — accepted distances are >= 0.001, so their bit patterns order like the numbers.
		// The one thing a minimum cannot reproduce is a NaN distance (a degenerate ray), which the sequential rule lets
		// in and then never displaces consistently: if any lane meets one, the ray is reported as not scanned and goes
		// through the sequential scan.  The per-sphere arithmetic is finish_sphere's, bit for bit.
		[[maybe_unused]] constexpr uint32_t sparse_wave_rays = 8;	 // a wave holding at most this many rays scans together
		constexpr uint32_t sparse_min_spheres = sparse_launch_min_spheres; // (below that a sequential scan is a few microseconds anyway)

		__device__ __forceinline__ void test_sphere_alone(candidate& best, bool& met_nan, vec3 o, vec3 d, float4 s, uint32_t index)
		{
			const sphere_probe p = probe_sphere(o, d, s);
			if (p.pos) // a per-lane branch: every lane looks at a different sphere, and few spheres lie on a ray's line
			{
				const float f = sqrt_rn(p.disc);
				const float t = (p.e2 < s.w) ? p.a + f : p.a - f;
				met_nan = met_nan || t != t;
				const bool accept = !(t < min_hit_dist) && !(best.have && best.t <= t);
				best.t = accept ? t : best.t;
				best.index = accept ? index : best.index;
				best.have = best.have || accept;
			}
		}

		// call converged, with a wave-uniform ray; returns false if the ray has to be scanned sequentially instead
		__device__ __forceinline__ bool scan_spheres_together(candidate& found, vec3 o, vec3 d, const float4* __restrict__ table, uint32_t count, uint32_t lane)
		{
			candidate mine = { 0.0f, 0u, false };
			bool met_nan = false;
			uint32_t i = lane;
			// four independent loads in flight per lane: the walk is bound by L2 latency, not by arithmetic.  (The registers
			// they take cost the kernel one wave per SIMD of occupancy, 5 instead of 6, which the sequential scan — bound by
			// vector issue — does not miss: config 5 5.61 against 5.69 s with two loads in flight, 10 000 spheres 277 against
			// 290 ms; profiles/r03/config5_streamed/sparse_waves_ab.txt.)
			for (; i + 192u < count; i += 256u)
			{
				const float4 s0 = table[i], s1 = table[i + 64u], s2 = table[i + 128u], s3 = table[i + 192u];
				test_sphere_alone(mine, met_nan, o, d, s0, i);
				test_sphere_alone(mine, met_nan, o, d, s1, i + 64u);
				test_sphere_alone(mine, met_nan, o, d, s2, i + 128u);
				test_sphere_alone(mine, met_nan, o, d, s3, i + 192u);
			}
			for (; i < count; i += 64u)
				test_sphere_alone(mine, met_nan, o, d, table[i], i);
			if (__builtin_amdgcn_ballot_w64(met_nan) != 0)
				return false;
			uint32_t key_t = mine.have ? __float_as_uint(mine.t) : 0xFFFFFFFFu; // (no accepted distance has this pattern: it is a NaN's)
			uint32_t key_i = mine.have ? mine.index : 0xFFFFFFFFu;
#pragma unroll
			for (int offset = 32; offset > 0; offset >>= 1)
			{
				const uint32_t other_t = __shfl_xor(key_t, offset, 64);
				const uint32_t other_i = __shfl_xor(key_i, offset, 64);
				const bool take = other_t < key_t || (other_t == key_t && other_i < key_i);
				key_t = take ? other_t : key_t;
				key_i = take ? other_i : key_i;
			}
			found.have = key_t != 0xFFFFFFFFu;
			found.t = found.have ? __uint_as_float(key_t) : 0.0f;
			found.index = found.have ? key_i : 0u;
			return true;
		}

		constexpr uint32_t small_table_float4s = 2u * scalar_max_spheres; // LDS tables of the scalar-register kernels: geometry, shading

		// everything a lane carries between loop trips
		struct lane_state
		{
			vec3 origin, dir;	// current ray
			vec3 throughput;	// product of attenuations so far (trace unrolled front to back)
			vec3 chunk_sum;		// running sum of the chunk of samples in flight (:186,193)
			// what a pixel's samples start from (frame_params): pinhole camera — the near-to-far vector through the pixel's
			// corner; eye form — s N' and s N.w of the corner; homogeneous form — (x, y, -, -) of the pixel
			float base_x, base_y, base_z, base_w; // (named words, not an array: hipcc otherwise keeps half of it in scratch)
			stream_keys keys;	// random streams of the pixel (contract.hpp): the key of its hash function and its counter stride
			uint32_t counter;	// random stream position
			// The samples of an item, counted in STREAM POSITIONS: a sample's window starts at stride * (index << 12)
			// (contract.hpp), so "the next sample" is one shift-and-add on the window start and nothing has to be multiplied
			// per restart.  `window` = where the sample in flight started, `window_end` = where the first sample behind the
			// item would start (the stride is odd and an index is below 2^20: distinct samples, distinct windows; index 0 is
			// the one window that starts at position 0).
			uint32_t window, window_end;
			uint32_t sample;	 // [INDEXED kernels only] index of the sample in flight
			uint32_t sample_end; // [INDEXED kernels only] one past the last sample of the item in flight
		};

		// image row of row `local_row` of this rank's compact buffer (rt_hip_partition).  All branches are wave-uniform.
		__device__ __forceinline__ uint32_t global_row(uint32_t local_row, const frame_params& p)
		{
			if (p.world == 1u)
				return local_row; // the whole frame: no arithmetic at all
			if (p.stripe_shift != 0xFFFFFFFFu) // stripes of 2^k rows (the default, 8): shifts instead of a division
				return ((((local_row >> p.stripe_shift) * p.world) + p.rank) << p.stripe_shift) | (local_row & ((1u << p.stripe_shift) - 1u));
			return ((local_row / p.stripe_rows) * p.world + p.rank) * p.stripe_rows + (local_row % p.stripe_rows);
		}

		// row of the output buffers that local row `ly` of this rank is written to (wave-uniform branch)
		__device__ __forceinline__ uint32_t output_row(uint32_t ly, const frame_params& p) { return p.frame_rows ? global_row(ly, p) : ly; }

		// :195-200 — mean, sqrt "gamma", pack, store
		__device__ __forceinline__ void finish_pixel(vec3 colour, const frame_params& p, uint32_t lx, uint32_t ly, uint32_t* out_rgba, float* out_rgb)
		{
			const float n = static_cast<float>(p.samples_per_pixel);
			const vec3 mean = { colour.x / n, colour.y / n, colour.z / n };
			const size_t o = static_cast<size_t>(output_row(ly, p)) * p.width + lx;
			if (out_rgb)
			{
				out_rgb[o * 3 + 0] = mean.x;
				out_rgb[o * 3 + 1] = mean.y;
				out_rgb[o * 3 + 2] = mean.z;
			}
			// A system-scope store: written through to memory at once.  When the frame is the caller's page-locked back buffer
			// (rt_hip_render) the pixels then cross PCIe while the rest of the frame is still being traced; an ordinary store
			// would sit in L2 until the end-of-kernel write-back and put the whole 8 MB transfer behind the kernel
			// (measured: + 0.10-0.14 ms per frame, profiles/r02/frame_store_ab.txt).  For a frame in HBM it costs nothing.
			__hip_atomic_store(&out_rgba[o], pack_rgba8888({ __builtin_sqrtf(mean.x), __builtin_sqrtf(mean.y), __builtin_sqrtf(mean.z) }), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}

		// `segments` = the wave's count (wave-uniform: kept in a scalar register, one popcount of the tracing lanes per trip)
		__device__ __forceinline__ void add_segments(device_counters* counters, unsigned long long segments)
		{
			if ((threadIdx.x & 63u) == 0 && segments)
				atomicAdd(&counters->segments[(blockIdx.x + blockIdx.y * 7u + (threadIdx.x >> 6)) % device_counters::segment_counters], segments);
		}

		// ---- small / resident kernel: a wave works through a queue of (pixel, sample chunk) items -----------------------
		//
		// The pixel sum is defined chunk-wise (arithmetic contract: chunks of 16 consecutive samples, each summed in
		// sample order, chunk sums added in chunk order).  That makes ONE CHUNK the unit of work: a wave owns a small
		// tile of P pixels = P x K items (K = chunks per pixel); its 64 lanes pull items from the wave's queue as they
		// finish the previous one, park every chunk sum in a wave-private LDS slot, and at the end P lanes fold the
		// slots in chunk order and write the pixels.  Short items keep every lane busy until the queue is empty and
		// keep waves short, so that a launch has tens of thousands of them to balance over the chip; which lane
		// computes which chunk cannot change a result.
		//
		// NS > 0: `small` kernel, NS spheres in SGPRs (kernel argument).  NS == 0: `resident` kernel, all primitives in LDS.
		// NS == -1: `tiled` kernel — the primitives stream from the SoA columns in HBM/L2 through one LDS tile shared by the
		// workgroup's four waves (coalesced dword per lane per column, radius squared on the way in); every wave still
		// runs its own queue, but the workgroup advances in lock step, one path segment per trip, with barriers around
		// each tile, until all four waves are done.  NS == -2: `streamed` kernel — the resident loop reading the primitive
		// table from HBM/L2 with wave-uniform scalar loads.  NS == -3: the same without the cooperative scan of sparse waves, for frames
		// that fill the device (6 waves per SIMD; RT_HIP_WAVES_DENSE above).
		// SM: scatter table of sm_ray_tracer (RT_HIP_FLAG_SM_MATERIALS) instead of mg_ray_tracer's.
		// launch bounds: compiled for 7 waves per SIMD.  That raises the compiler's budgets from 64 to 72 vector and from 72
		// to 88 scalar registers.  With 5..8 spheres in SGPRs and in the resident kernel 64 VGPRs mean scratch
		// (dielectric.toml, 7 spheres: 3.85 -> 3.54 ms with 72; resident, 1000 spheres: -2 %).  The 1..4 sphere kernels
		// still fit 64 VGPRs — they keep running 8 waves — and use the extra scalar registers: the lane-mask bookkeeping of
		// the divergent loop no longer spills to VGPR lanes (3.12 -> 3.05 ms).  6 waves are slower everywhere.  The
		// big-scene modes get the registers of 5 waves per SIMD.
		//
		// ROLLING ITEMS (NS < 0).  In a big scene a trip is one closest-hit scan over all primitives — the same cost for a
		// wave with one lane holding a ray as for a wave with 64 — so what matters is that every lane holds a ray in every
		// trip, and that all waves of the launch run dry at the same moment.  The big-scene kernels are therefore launched
		// persistent, and their unit of hand-out is the single ITEM: all (pixel, chunk) items of this rank's rows form ONE
		// launch-wide sequence (pixel-major, bottom row first); a wave draws small blocks of consecutive items from one
		// counter — one block ahead, so that the atomic's latency is off the path — and gives them to its lanes as they
		// fall free, in the same trip.  A pixel's chunks may end up in different waves, on different CUs and XCDs: every
		// finished chunk sum goes to a scratch slot in HBM (write-through stores, drained), then the pixel's arrival
		// counter is incremented; whoever brings the LAST chunk of a pixel reads all of them back (loads that bypass the
		// non-coherent caches), folds them in chunk order and writes the pixel.  Which lane of which wave that is cannot
		// change a bit of the result.
		// Round 2 handed out pixel TILES (128 items, two open per wave, sums in LDS).  Measured with per-wave clocks on the
		// 100 000-sphere frame (profiles/r03/config5_streamed/wave_tail_*.txt): the queue ran dry after 74 % of the launch
		// and a wave then still owned up to 256 items — four rounds of 34 trips for its lanes — so 20 % of all wave-time
		// was spent waiting for the slowest waves.  With items the stragglers' excess is one item, not four tiles' worth.
		// (For small scenes the per-trip cost is the shading code, and lanes that never start together lose the phase
		// coherence that keeps it short — profiles/r01/queue_shape_sweep.txt.  They keep one tile per wave.)
		// half-chunks: how many samples of chunk `chunk` lie in its second half (indices chunk * 16 + 8 ..)
		__device__ __forceinline__ uint32_t parked_samples(uint32_t samples_per_pixel, uint32_t chunk)
		{
			const uint32_t first = chunk * sample_chunk + sample_chunk / 2u;
			return samples_per_pixel > first ? min(samples_per_pixel - first, sample_chunk / 2u) : 0u;
		}

		__device__ __forceinline__ unsigned long long fetch_items(device_counters* counters, uint32_t count) // call converged
		{
			unsigned long long base = 0;
			if ((threadIdx.x & 63u) == 0)
				base = atomicAdd(&counters->next_item, static_cast<unsigned long long>(count));
			const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(base));
			const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(base >> 32));
			return (static_cast<unsigned long long>(hi) << 32) | lo;
		}

		// A value (a chunk's sum, or one sample's) on its way between waves: 16 bytes per slot, ONE store and ONE load, both
		// past this CU's L1 and this XCD's L2 (... sc1: the XCDs' L2s are not coherent with one another).
		//   Round 3 wrote a slot as an 8-byte and a 4-byte agent-scope atomic store and read it back the same way: two memory
		//   requests each way per sample, tallied by the counters at 32 bytes apiece (config 5: WRITE_SIZE 8.9 GB for 2.1 GB of
		//   payload).  hipcc has no 16-byte atomic, so the single dwordx4 access is written out
		//   (the A/B against the old form: profiles/r04/ab_publish_16_bytes.txt).
		// THE HAND-OVER PROTOCOL (publish -> count the arrival -> the last arrival reads), as the hardware executes it:
		//   1. the producer's stores are write-through (sc1): when `s_waitcnt vmcnt(0)` lets the wave go on, they have been
		//      acknowledged by the memory side (gfx950 counts stores in vmcnt and acknowledges a write-through store when it
		//      has left the L2 for the fabric);
		//   2. the arrival is an agent-scope atomic add, executed at the memory side, issued after (1) in program order;
		//   3. the consumer's loads are issued after the atomic's RETURN VALUE has come back (they depend on the branch on
		//      it) and bypass the non-coherent caches (sc1): they observe memory as of a moment after every producer's (1).
		// That is an argument from the gfx950 ISA, not from the HIP memory model: in the model's terms the
		// arrival would be a release (buffer_wbl2 sc1 + s_waitcnt in front of the atomic) and the last arrival an acquire
		// (buffer_inv sc1 behind it).  RT_HIP_MODEL_ORDERING=1 builds exactly that; measured, it costs the streamed kernel
		// its scene — buffer_inv sc1 drops the XCD's L2 copy of the sphere table once per finished pixel
		// (profiles/r04/arrival_ordering_ab.txt) — so the ISA-level form stays, and this file refuses to be compiled for
		// anything but gfx950, where the argument was made and the suite's parity tests exercise it.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "kernels.hip: the cross-wave hand-over of the rolling kernels relies on gfx950's store acknowledgement and sc1 semantics"
#endif
		typedef float v4f __attribute__((ext_vector_type(4)));

		__device__ __forceinline__ void publish_sum(unsigned long long* slot, vec3 sum)
		{
			const v4f value = { sum.x, sum.y, sum.z, 0.0f };
			// The two wait states behind the store are part of it: a VMEM store of more than 64 bits reads its data registers
			// after issue, and a vector instruction that overwrites them within two wait states corrupts the stored value
			// (gfx940+ "VMEM store data hazard").  The compiler pads its own stores; it cannot see into this one — found the
			// hard way: the sm kernels' next instruction recycled the value's upper half for an address, and z arrived as a
			// pointer's low word (profiles/r04/case0_bisect.txt).
			asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(slot), "v"(value) : "memory");
		}
		__device__ __forceinline__ vec3 read_sum(unsigned long long* slot)
		{
			v4f value;
			asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(value) : "v"(slot) : "memory");
			return { value.x, value.y, value.z };
		}
		// eight consecutive slots, the loads in flight together
		__device__ __forceinline__ void read_sums8(unsigned long long* first, vec3 (&out)[8])
		{
			v4f v0, v1, v2, v3, v4, v5, v6, v7;
			asm volatile("global_load_dwordx4 %0, %8, off sc1\n\t"
						 "global_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
						 "global_load_dwordx4 %2, %8, off offset:32 sc1\n\t"
						 "global_load_dwordx4 %3, %8, off offset:48 sc1\n\t"
						 "global_load_dwordx4 %4, %8, off offset:64 sc1\n\t"
						 "global_load_dwordx4 %5, %8, off offset:80 sc1\n\t"
						 "global_load_dwordx4 %6, %8, off offset:96 sc1\n\t"
						 "global_load_dwordx4 %7, %8, off offset:112 sc1\n\t"
						 "s_waitcnt vmcnt(0)"
						 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
						 : "v"(first)
						 : "memory");
			out[0] = { v0.x, v0.y, v0.z }, out[1] = { v1.x, v1.y, v1.z }, out[2] = { v2.x, v2.y, v2.z }, out[3] = { v3.x, v3.y, v3.z };
			out[4] = { v4.x, v4.y, v4.z }, out[5] = { v5.x, v5.y, v5.z }, out[6] = { v6.x, v6.y, v6.z }, out[7] = { v7.x, v7.y, v7.z };
		}

		// the arrival of an item at its pixel's counter, and what the lane that brings the LAST item does before it reads
		__device__ __forceinline__ uint32_t count_arrival(uint32_t* counter)
		{
#ifdef RT_HIP_MODEL_ORDERING
			return __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
			asm volatile("s_waitcnt vmcnt(0) ; what was published has left before the arrival is counted" ::: "memory");
			return __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
		}
		__device__ __forceinline__ void last_arrival()
		{
#ifdef RT_HIP_MODEL_ORDERING
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // (no instruction: keeps the loads below the counter's answer)
#endif
		}

		// HALF (small and resident kernels, short launches; see choose_queue): the unit of work is HALF a chunk, 8 samples.
		// The chunk sum stays what the contract says — sixteen samples added in sample order — so the lane that traces a
		// chunk's second half cannot add anything up itself: it parks its (up to) eight sample values in the tile's LDS
		// slot, next to the first half's partial sum, and the fold at the end of the wave continues that partial sum with
		// them, in order.  Twice as many, half as long items: what a launch with two or three chunks per lane needs to end
		// evenly (profiles/r03/chunk_probe.txt).
		// NP (small kernel only; round 4): the NS scalar-register spheres are FOLLOWED BY NP PLANES (slots NS .. NS + NP - 1 of
		// the kernel argument), so that the reference's own scenes with their ground plane back in (scenes/basic.toml:11-13,
		// scenes/dielectric.toml:17-19) stay on the scalar-operand path.  Which slot is what is known at compile time: no
		// kind check at run time, and the no-plane variants are what they were.  NS + NP <= 8, NP <= 3 (more planes, or planes
		// without a sphere, take the LDS-resident kernel).
		// GC (small kernel only; round 4): the GENERAL camera — an inverse view-projection whose w varies over the frame takes
		// the per-sample division (camera.hpp:42-48) instead of contract v3's affine primary rays.  rt's own matrix is the
		// float inverse of projection x view (camera.hpp:122-137): for a camera that is not axis-aligned its last row comes
		// out with rounding noise in the x and y terms (-4.5e-7 against a w of 0.001 at the far plane: 4.5e-4 relative — not
		// something to round away), i.e. every frame while the user looks around.  Those frames used to fall to the
		// LDS-resident kernel (+35 % on basic.toml); now they keep the scalar-register kernel, in a build of it that carries
		// the 18 scalars of the general form INSTEAD of the 18 of the affine one (both would not fit its scalar registers).
		template <int NS, bool SM, bool HALF = false, int NP = 0, bool GC = false>
		__global__ __launch_bounds__(block_threads, NS == -3 ? RT_HIP_WAVES_DENSE : (NS < 0 ? 5 : (NS == 0 ? RT_HIP_WAVES_RESIDENT : (NS >= 6 && NS + NP >= 7 ? RT_HIP_WAVES_MANY : RT_HIP_WAVES_FEW)))) void render_queue(const frame_params p,
																	  const queue_params q,
																	  const small_scene small,
																	  const device_scene s,
																	  const float4* __restrict__ geometry, // = s.primitive_geometry, as a plain read-only argument
																	  uint32_t* __restrict__ out_rgba,
																	  float* __restrict__ out_rgb,
																	  device_counters* __restrict__ counters,
																	  unsigned long long* item_sums, // [NS < 0] chunk sums in transit: 16 bytes per item
																	  uint32_t* pixel_done)			 // [NS < 0] items arrived per pixel (zeroed in front of every launch)
		{
			extern __shared__ float4 lds[];
			// [NS > 0] 8 geometry (with the scatter function) + 8 shading float4s | [NS == 0] all primitives; then the chunk slots
			float4* const lds_geometry = lds;
			float4* const lds_shading = lds + scalar_max_spheres;
			constexpr bool RESIDENT = NS == 0; // all primitives in LDS
			// [NS == 0] from resident_scalar_scan_from spheres on the sphere scan reads the table in memory (scalar loads): only the planes are staged
			// (a build of its own, NP == 1: the scalar-load scan keeps two groups of four spheres in 32 scalar registers, which the
			// LDS-scan build — scenes below the threshold — has for the frame's constants instead)
			constexpr bool RESIDENT_SCALAR_SCAN = NS == 0 && NP == 1;
			const bool spheres_in_lds = RESIDENT && !RESIDENT_SCALAR_SCAN;
			const uint32_t lds_spheres = spheres_in_lds ? s.n_spheres : 0u;
			const uint32_t table_float4s = NS > 0 ? small_table_float4s : (RESIDENT ? lds_spheres + s.n_planes : (NS == -1 ? tile_primitives : 0u));
			if (NS > 0)
			{
				if (threadIdx.x == 0)
				{
#pragma unroll
					for (int i = 0; i < NS + NP; i++) // constant indices: the argument block is never indexed dynamically
					{
						// (what a hit looks up: the centre / the normal, and in the fourth word — the scan has radius^2 / d from the
						// scalar registers — the scatter function: ONE 16-byte read for both)
						lds_geometry[i] = make_float4(small.geometry[i].x, small.geometry[i].y, small.geometry[i].z, __uint_as_float(small.scatter[i]));
						lds_shading[i] = small.shading[i];
					}
				}
			}
			else if (RESIDENT)
			{
				for (uint32_t i = threadIdx.x; i < lds_spheres + s.n_planes; i += block_threads)
					lds[i] = s.primitive_geometry[(s.n_spheres - lds_spheres) + i];
			}
			__syncthreads();

#ifdef RT_HIP_REGION_COUNTERS
			uint32_t region_runs[device_counters::regions] = {}, region_lanes[device_counters::regions] = {};
#endif
			constexpr bool ROLLING = NS < 0;
			// the camera form a scalar-register kernel is built for: the pinhole form, or (GC) the eye form.  A matrix without a
			// finite eye (an orthographic frustum: nothing rt's camera can produce) takes the LDS-resident kernel, which carries all
			// three forms — together with seven spheres their scalars do not fit the scalar registers, and hipcc then reloads the
			// SPHERES from the argument block inside the loop (round 5: dielectric.toml through a tilted camera 3.85 ms).
			// The LDS-resident kernel is built per scan, and its LDS-scan build per camera form too: as ONE kernel it carried the three camera
			// forms' constants AND the scalar-load scan's two groups of four spheres (32 scalar registers), and its loop reloaded the frame's
			// constants from the argument block (30 s_load and 35 v_readlane of spilled scalars per loop body).  NP == 0: the LDS scan (scenes
			// below resident_scalar_scan_from spheres), GC = "the frame is not a pinhole's" (eye or homogeneous form, chosen at run time);
			// NP == 1: the scalar-load scan, all forms at run time as before (split by form it gained nothing and lost 4 % through a tilted
			// camera).  12 spheres 1.23 -> 1.15 ms, 32: 2.06 -> 2.01, basic.toml forced here 2.89 -> 2.72 (profiles/r05/resident_forms_ab.txt).
			constexpr bool RESIDENT_LDS_SCAN = NS == 0 && NP == 0;
			constexpr bool PINHOLE_ONLY = (NS > 0 || RESIDENT_LDS_SCAN) && !GC, EYE_ONLY = NS > 0 && GC, NEVER_PINHOLE = RESIDENT_LDS_SCAN && GC;
			const uint32_t lane = threadIdx.x & 63u;
			const uint32_t wave = threadIdx.x >> 6;
			const uint32_t chunk_items = q.chunks << q.pixels_log2;	   // chunks of one pixel tile: P x K
			const uint32_t items = HALF ? 2u * chunk_items : chunk_items; // work items of the tile
			// chunk-sum slots of this wave's tile (one tile per wave; the rolling kernels keep no sums in LDS): 3 floats per
			// chunk; HALF: 27 — the first half's partial sum, then the second half's eight sample values
			constexpr uint32_t slot_floats = HALF ? half_chunk_slot_floats : 3u;
			float* const slots = reinterpret_cast<float*>(lds + table_float4s) + static_cast<size_t>(wave) * chunk_items * slot_floats;
			const uint32_t tile_w = 1u << q.tile_w_log2;
			const uint32_t tile_h = (1u << q.pixels_log2) >> q.tile_w_log2;

			// one tile per wave: workgroups are handed out bottom row first — the lower part of a frame is usually the
			// expensive one (ground under sky), and starting with it keeps the cheap sky tiles for the tail
			const uint32_t tile_x0 = (blockIdx.x * 4u + wave) * tile_w;
			const uint32_t tile_y0 = (gridDim.y - 1u - blockIdx.y) * tile_h;
			uint32_t next_item = 0; // wave-uniform queue head

			// rolling items: the launch-wide sequence, the block this wave is handing out, the block drawn ahead
			// rolling + HALF: a pixel is cut into items of q.item_samples consecutive samples, whatever the chunks are
			const uint32_t items_per_pixel = (ROLLING && HALF) ? (p.samples_per_pixel + q.item_samples - 1u) / q.item_samples : q.chunks;
			const unsigned long long total_items = static_cast<unsigned long long>(p.width) * p.local_rows * items_per_pixel;
			unsigned long long block_next = 0, block_end = 0; // [block_next, block_end): not yet given to a lane
			unsigned long long prefetched = 0;				  // first item of the block after that
			uint32_t prefetched_count = ROLLING ? min(64u, q.lane_cap) : 64u; // (the first block is a whole wave's worth — or the wave's cap: every lane that may hold a ray starts at once)
			bool dry = false;								  // the launch-wide sequence has run out
			if (ROLLING)
				prefetched = fetch_items(counters, prefetched_count);
#ifdef RT_HIP_WAVE_CLOCKS
			const unsigned long long clock_start = wall_clock64();
			unsigned long long clock_dry = 0;
#endif

			lane_state st;
			// path segments traced: counted on the scalar unit (one popcount of the tracing lanes per trip) where scalar
			// registers are to spare — the 1..4-sphere kernels; 2 more live SGPRs make the others spill — else per lane
			constexpr bool SCALAR_SEGMENTS = NS >= 1 && NS + NP <= 4;
			unsigned long long wave_segments = 0;
			uint32_t lane_segments = 0;
			uint32_t slot = 0; // chunk-sum slot of the item in flight on this lane; rolling: its pixel's place in the hand-out order
			// what the lane does in the current trip:
			//   free    - between items                       restart - has a sample to start (needs a primary ray)
			//   trace   - has a ray: closest-hit query next    retired - the queue ran dry
			// ONE register says both: 0 free, 1 restart, 3 retired, and from lane_trace (4) upwards "tracing, and the path may still
			// make (mode - 4) bounces after the segment in flight" — the bounce count of `if (!(max_bounces--)) return {}` (:157).  A
			// restart sets both with one move, a bounce takes one off, and the lane that would go below 4 has run out of bounces.
			// (Round 4 tried the mode as three lane MASKS kept in scalar registers instead of a number in a vector register: same
			// vector instruction count, 42 % more scalar ones, 2.64 -> 2.85 ms — profiles/r04/ab_mode_masks.txt, HISTORY.md.)
			enum : uint32_t { lane_free = 0, lane_restart = 1, lane_retired = 3, lane_trace = 4 };
			uint32_t mode = lane_free;
#define RT_HIP_IS(m) ((m) == lane_trace ? mode >= lane_trace : mode == (m))
#define RT_HIP_BECOME(m) (mode = (m))

			// all items of a tile are in: fold the chunk sums of each pixel in chunk order and write it (:195-200)
			const auto fold_tile = [&](const float* sums, uint32_t x0, uint32_t y0)
			{
				if (!ROLLING && (3u << q.pixels_log2) <= 64u) // wave-uniform
				{
					// Small tiles (<= 16 pixels: from 113 samples per pixel upwards) are folded one CHANNEL per lane — lane
					// = channel * P + pixel — so that the chunk loop, the division by the sample count and the square
					// root are issued once for all three channels instead of three times by a third of the lanes.  Per
					// channel the arithmetic is what finish_pixel does; the packed bytes then meet in the pixel's lane.
					const uint32_t channel = lane >> q.pixels_log2;
					const uint32_t pixel = lane & ((1u << q.pixels_log2) - 1u);
					const uint32_t lx = x0 + (pixel & (tile_w - 1u));
					const uint32_t ly = y0 + (pixel >> q.tile_w_log2);
					const bool live = channel < 3u && lx < p.width && ly < p.local_rows;
					float sum = 0.0f;
					if (live)
					{
						if (HALF)
						{
							for (uint32_t c = 0; c < q.chunks; c++)
							{
								const float* const slot_of_chunk = sums + ((c << q.pixels_log2) + pixel) * slot_floats;
								float chunk_sum = slot_of_chunk[channel]; // samples 0..7 of the chunk, then 8.. in order
								const uint32_t parked = parked_samples(p.samples_per_pixel, c);
								for (uint32_t j = 0; j < parked; j++)
									chunk_sum = chunk_sum + slot_of_chunk[3u + j * 3u + channel];
								sum = c ? sum + chunk_sum : chunk_sum;
							}
						}
						else
						{
							sum = sums[pixel * 3u + channel];
							for (uint32_t c = 1; c < q.chunks; c++)
								sum = sum + sums[((c << q.pixels_log2) + pixel) * 3u + channel];
						}
					}
					const float mean = sum / static_cast<float>(p.samples_per_pixel);
					const size_t o = static_cast<size_t>(output_row(live ? ly : 0u, p)) * p.width + lx;
					if (live && out_rgb)
						out_rgb[o * 3u + channel] = mean;
					const uint32_t byte = static_cast<uint32_t>(clamp01(__builtin_sqrtf(mean)) * 255.99999f);
					const uint32_t green = __shfl(byte, lane + (1u << q.pixels_log2), 64);
					const uint32_t blue = __shfl(byte, lane + (2u << q.pixels_log2), 64);
					if (live && channel == 0u)
						__hip_atomic_store(&out_rgba[o], (byte << 24u) | (green << 16u) | (blue << 8u) | 255u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					return;
				}
				for (uint32_t pixel = lane; pixel < (1u << q.pixels_log2); pixel += 64u)
				{
					const uint32_t lx = x0 + (pixel & (tile_w - 1u));
					const uint32_t ly = y0 + (pixel >> q.tile_w_log2);
					if (lx < p.width && ly < p.local_rows)
					{
						vec3 colour = { 0.0f, 0.0f, 0.0f };
						if (HALF)
						{
							for (uint32_t c = 0; c < q.chunks; c++)
							{
								const float* const slot_of_chunk = sums + ((c << q.pixels_log2) + pixel) * slot_floats;
								vec3 chunk_sum = { slot_of_chunk[0], slot_of_chunk[1], slot_of_chunk[2] };
								const uint32_t parked = parked_samples(p.samples_per_pixel, c);
								for (uint32_t j = 0; j < parked; j++)
									chunk_sum = chunk_sum + vec3{ slot_of_chunk[3u + j * 3u], slot_of_chunk[4u + j * 3u], slot_of_chunk[5u + j * 3u] };
								colour = c ? colour + chunk_sum : chunk_sum;
							}
						}
						else
						{
							colour = { sums[pixel * 3u], sums[pixel * 3u + 1u], sums[pixel * 3u + 2u] };
							for (uint32_t c = 1; c < q.chunks; c++)
							{
								const uint32_t at = ((c << q.pixels_log2) + pixel) * 3u;
								colour = colour + vec3{ sums[at], sums[at + 1u], sums[at + 2u] };
							}
						}
						finish_pixel(colour, p, lx, ly, out_rgba, out_rgb);
					}
				}
			};
			// rolling: the lane's item is finished.  One chunk per pixel (<= 16 samples): the sum IS the pixel.  Otherwise the
			// sum is published, the pixel's arrival counter incremented, and the lane that brings the last chunk folds them.
			const auto complete_item = [&]()
			{
				const uint32_t row = slot / p.width; // hand-out order: bottom row first
				const uint32_t lx = slot - row * p.width;
				const uint32_t ly = p.local_rows - 1u - row;
				if (HALF)
				{
					// sub-chunk items: every sample's value went to its own place (end_sample); the lane that brings a pixel's
					// last item adds them up as the contract says — sixteen in sample order per chunk, chunks in chunk order
					unsigned long long* const samples = item_sums + 2u * static_cast<size_t>(slot) * p.samples_per_pixel;
					const uint32_t before = count_arrival(&pixel_done[slot]);
					if (before + 1u == items_per_pixel)
					{
						last_arrival();
						vec3 colour = { 0.0f, 0.0f, 0.0f };
						for (uint32_t first = 0; first < p.samples_per_pixel; first += sample_chunk)
						{
							const uint32_t end = min(first + sample_chunk, p.samples_per_pixel);
							vec3 chunk_sum = { 0.0f, 0.0f, 0.0f };
							uint32_t i = first;
							for (; i + 8u <= end; i += 8u) // eight loads in flight, then eight additions in sample order
							{
								vec3 value[8];
								read_sums8(samples + 2u * i, value);
#pragma unroll
								for (uint32_t k = 0; k < 8u; k++)
									chunk_sum = chunk_sum + value[k];
							}
							for (; i < end; i++)
								chunk_sum = chunk_sum + read_sum(samples + 2u * i);
							colour = first ? colour + chunk_sum : chunk_sum;
						}
						finish_pixel(colour, p, lx, ly, out_rgba, out_rgb); // (the counter stays: every launch zeroes its counters first)
					}
					return;
				}
				if (q.chunks == 1u) // (wave-uniform)
				{
					finish_pixel(st.chunk_sum, p, lx, ly, out_rgba, out_rgb);
					return;
				}
				const uint32_t chunk = (st.sample_end - 1u) / sample_chunk;
				unsigned long long* const sums = item_sums + 2u * static_cast<size_t>(slot) * q.chunks;
				publish_sum(sums + 2u * chunk, st.chunk_sum);
				const uint32_t before = count_arrival(&pixel_done[slot]);
				if (before + 1u == q.chunks)
				{
					last_arrival();
					vec3 colour = read_sum(sums);
					for (uint32_t c = 1; c < q.chunks; c++)
						colour = colour + read_sum(sums + 2u * c);
					finish_pixel(colour, p, lx, ly, out_rgba, out_rgb); // (the counter stays: every launch zeroes its counters first)
				}
			};

			// on to the next sample of the item in flight; false = that was the last one.  The lane's stream moves to the next
			// sample's window (a restarting lane has no other use for its position).  The kernels that look at a sample's INDEX
			// (half chunks park by it, rolling items publish by it) count indices; the others count windows only.
			constexpr bool INDEXED = HALF || ROLLING;
			const auto next_sample = [&]() -> bool
			{
#ifdef RT_HIP_FAST_BUILD
				st.window += st.keys.stride << draws_per_sample_log2;
#else
				// (written out with its result tied to its operand's register: hipcc otherwise forms the sum in the counter's register
				// and carries the window through two more moves to the end of the query)
				asm("v_lshl_add_u32 %0, %1, %2, %0" : "+v"(st.window) : "v"(st.keys.stride), "n"(draws_per_sample_log2));
#endif
				if (INDEXED)
					return ++st.sample < st.sample_end;
				return st.window != st.window_end;
			};

			// `colour += trace(...)` (:193) for the sample in flight, then the next sample of the chunk or the end of the item
			const auto end_sample = [&](vec3 contribution)
			{
				if (HALF && ROLLING)
				{
					publish_sum(item_sums + 2u * (static_cast<size_t>(slot) * p.samples_per_pixel + st.sample), contribution);
					if (next_sample())
						RT_HIP_BECOME(lane_restart);
					else
					{
						complete_item();
						RT_HIP_BECOME(lane_free);
					}
					return;
				}
				if (HALF)
				{
					// slot = (half-chunk index << pixels_log2) + pixel; its chunk's LDS slot; sample bit 3 = second half
					const uint32_t half_chunk = slot >> q.pixels_log2;
					float* const slot_of_chunk = slots + ((((half_chunk >> 1u) << q.pixels_log2) + (slot & ((1u << q.pixels_log2) - 1u))) * slot_floats);
					if (st.sample & 8u) // second half: the sample's value is parked as it is
					{
						float* const place = slot_of_chunk + 3u + (st.sample & 7u) * 3u;
						place[0] = contribution.x;
						place[1] = contribution.y;
						place[2] = contribution.z;
					}
					else
						st.chunk_sum = st.chunk_sum + contribution;
					if (next_sample())
						RT_HIP_BECOME(lane_restart);
					else
					{
						if (!((st.sample - 1u) & 8u)) // first half: its partial sum
						{
							slot_of_chunk[0] = st.chunk_sum.x;
							slot_of_chunk[1] = st.chunk_sum.y;
							slot_of_chunk[2] = st.chunk_sum.z;
						}
						RT_HIP_BECOME(lane_free);
					}
					return;
				}
				st.chunk_sum = st.chunk_sum + contribution;
				if (next_sample())
					RT_HIP_BECOME(lane_restart);
				else
				{
					if (ROLLING)
						complete_item();
					else
					{
						slots[slot * 3u + 0u] = st.chunk_sum.x;
						slots[slot * 3u + 1u] = st.chunk_sum.y;
						slots[slot * 3u + 2u] = st.chunk_sum.z;
					}
					RT_HIP_BECOME(lane_free);
				}
			};

			while (true)
			{
				// ---- closest-hit query for every lane that holds a ray (trace(), :160-162) -----------------------------------
				RT_HIP_REGION(0); // a trip
				const bool tracing = RT_HIP_IS(lane_trace);
				if (SCALAR_SEGMENTS)
					wave_segments += static_cast<unsigned>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(tracing)));
				candidate tiled_planes = { 0.0f, 0u, false };
				candidate tiled_spheres = { 0.0f, 0u, false };
				if (NS == -1 && __syncthreads_or(tracing)) // nothing to stream on the very first trip: no lane holds a ray yet
				{
					// all 256 threads stage, lanes without a ray just do not scan
					for (uint32_t first = 0; first < s.n_planes; first += tile_primitives)
					{
						const uint32_t count = min(tile_primitives, s.n_planes - first);
						__syncthreads(); // previous tile fully consumed
						stage_planes(lds, s, first, count);
						__syncthreads();
						if (tracing)
							scan_lds<false>(tiled_planes, st.origin, st.dir, lds, count, first);
					}
					for (uint32_t first = 0; first < s.n_spheres; first += tile_primitives)
					{
						const uint32_t count = min(tile_primitives, s.n_spheres - first);
						__syncthreads();
						stage_spheres(lds, s, first, count);
						__syncthreads();
						if (tracing)
							scan_lds<true>(tiled_spheres, st.origin, st.dir, lds, count, first);
					}
				}

				// After the query a lane either SHADES a hit (scatter: three draws, a new direction) or, having missed,
				// RESTARTS with the next sample's primary ray (two draws, a new direction).  Both end in the same
				// operations — random draws and the normalisation of a direction — which are therefore issued once for
				// the whole wave; only the pieces that really differ run under their own lane masks.
				// streamed kernel, a wave with a handful of rays: one cooperative scan per ray (see scan_spheres_together)
				candidate together = { 0.0f, 0u, false };
				bool scanned_together = false;
#ifdef RT_HIP_NO_SPARSE_WAVES // (experiment builds: the kernel without the cooperative scan)
				constexpr bool sparse_waves = false;
#else
				constexpr bool sparse_waves = true;
#endif
				if (sparse_waves && NS == -2 && s.n_spheres >= sparse_min_spheres)
				{
					unsigned long long holders = __builtin_amdgcn_ballot_w64(tracing);
					if (holders != 0 && static_cast<uint32_t>(__builtin_popcountll(holders)) <= q.sparse_rays)
					{
						while (holders != 0) // (wave-uniform)
						{
							const int src = __builtin_ctzll(holders);
							holders &= holders - 1u;
							const vec3 o = { __int_as_float(__builtin_amdgcn_readlane(__float_as_int(st.origin.x), src)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(st.origin.y), src)),
											 __int_as_float(__builtin_amdgcn_readlane(__float_as_int(st.origin.z), src)) };
							const vec3 d = { __int_as_float(__builtin_amdgcn_readlane(__float_as_int(st.dir.x), src)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(st.dir.y), src)),
											 __int_as_float(__builtin_amdgcn_readlane(__float_as_int(st.dir.z), src)) };
							candidate found;
							const bool clean = scan_spheres_together(found, o, d, geometry, s.n_spheres, lane);
							if (clean && lane == static_cast<uint32_t>(src))
							{
								together = found;
								scanned_together = true;
							}
						}
					}
				}

				bool shade = false;
				vec3 normal; // meaningful only where `shade` holds: deliberately not initialised
				float4 shading;
				uint32_t scatter_kind = scatter_lambert;
				if (tracing)
				{
					RT_HIP_REGION(1); // query: probes
					if (!SCALAR_SEGMENTS)
						lane_segments++;
					uint32_t kind;
					float distance;
					uint32_t small_index = 0;
					if (NS == -1)
					{
						uint32_t index;
						kind = select_hit(tiled_spheres, tiled_planes, distance, index);
						fetch_hit<SM>(s, st.origin, st.dir, kind, distance, index, normal, shading, scatter_kind);
					}
					else if (NS > 0)
					{
						// The discriminants of a GROUP of spheres first (independent straight-line code with scalar operands), one
						// branch for "no lane can hit any of them", then their square-root halves in index order.  Up to four spheres
						// are one group.  From five spheres on they go in groups of three: every discriminant of a group is three live
						// vector registers until its square-root half has run, and seven or eight at once put the kernel over its
						// budget (72 at 7 waves per SIMD) — round 4 lived with 12-16 bytes of scratch there; in round 5 the same source
						// took 84-124 bytes with the contract-v4 tail, and dielectric.toml went from 2.7 to 9.4 ms
						// (profiles/r05/codegen_ab.txt; tools/kernel_registers.py lists every build's registers and scratch).
						candidate best = { 0.0f, 0u, false };
						constexpr int spheres_ns = NS > 0 ? NS : 1; // (this branch is compiled, though never taken, for NS <= 0)
#ifdef RT_HIP_PROBE_GROUP
						constexpr int group = RT_HIP_PROBE_GROUP; // (A/B builds)
#else
						constexpr int group = spheres_ns <= 4 ? spheres_ns : 3;
#endif
#pragma unroll
						for (int first = 0; first < NS; first += group)
						{
							sphere_probe probes[group];
							unsigned long long lanes[group];
							unsigned long long any_lane = 0;
#pragma unroll
							for (int i = 0; i < group; i++)
								if (first + i < NS)
								{
									probes[i] = probe_sphere(st.origin, st.dir, small.geometry[first + i]); // SGPR operands
									lanes[i] = __builtin_amdgcn_ballot_w64(probes[i].pos);
									any_lane |= lanes[i];
								}
							if (any_lane != 0)
							{
#pragma unroll
								for (int i = 0; i < group; i++)
									if (first + i < NS)
									{
										if (lanes[i] != 0)
											RT_HIP_REGION(2); // square-root half of one sphere
										finish_sphere(best, probes[i], small.geometry[first + i].w, static_cast<uint32_t>(first + i), lanes[i]);
									}
							}
						}
						if (NP > 0)
						{
							// test_planes (:36-60) over the plane slots, with its own best candidate; then select(spheres, planes) (:96-102,160) —
							// select_hit() (scan.hpp) on the comparisons' masks: written over sentinel distances it costs three selects and a
							// divergent block more per query.  `>= 0` is hit_result's `operator bool` (:29-32; false for a NaN distance).
							float plane_distance;
							uint32_t plane_slot = static_cast<uint32_t>(NS);
							bool plane;
							if (NP == 1)
								plane = test_one_plane(st.origin, st.dir, small.geometry[NS > 0 ? NS : 0], plane_distance);
							else
							{
								candidate best_plane = { 0.0f, 0u, false };
#pragma unroll
								for (int j = 0; j < NP; j++)
									test_plane(best_plane, st.origin, st.dir, small.geometry[(NS > 0 ? NS : 0) + j], static_cast<uint32_t>(j));
								plane = best_plane.have && best_plane.t >= 0.0f;
								plane_distance = best_plane.t;
								plane_slot += best_plane.index;
							}
							const bool nearer = best.t <= plane_distance;
							const bool sphere = best.have && best.t >= 0.0f;
							const bool use_sphere = sphere && (!plane || nearer);
							const bool use_plane = plane && !use_sphere;
							distance = use_sphere ? best.t : plane_distance; // (a miss never reads it)
							kind = use_sphere ? 1u : (use_plane ? 2u : 0u);
							small_index = use_sphere ? best.index : plane_slot; // the winner's slot (one select; with one plane its index is a constant)
						}
						else
						{
							const bool hit = best.have && best.t >= 0.0f;
							kind = hit ? 1u : 0u;
							distance = best.t;
							small_index = best.index; // the lookups of the winning sphere follow below, in the one `hit` region
						}
					}
					else
					{
						candidate planes = { 0.0f, 0u, false };
						candidate spheres = { 0.0f, 0u, false };
						// resident: the LDS copy; streamed: the table in HBM/L2 itself, read with wave-uniform (scalar) loads
						const float4* const primitives = RESIDENT ? lds : geometry;
						scan_lds<false>(planes, st.origin, st.dir, RESIDENT ? lds + lds_spheres : geometry + s.n_spheres, s.n_planes, 0);
						if (NS == -2 || NS == -3)
						{
							if (scanned_together)
								spheres = together;
							else
								scan_streamed_spheres(spheres, st.origin, st.dir, geometry, s.n_spheres);
						}
						// From a few dozen spheres on the resident kernel reads the sphere table as the streamed kernel does — wave-uniform
						// scalar loads of (c, r^2) groups from the table in memory, the next group's load issued before this group's
						// probes — instead of broadcast reads from its LDS copy: a table of this size sits in the scalar cache, the probes
						// take their operands from scalar registers, and neither the LDS pipe nor four vector registers per sphere in
						// flight are spent on it (64 spheres x 256 spp 12.9 -> 12.5 ms, 200 spheres 9.6 -> 8.7, 700 33.3 -> 29.9; below
						// about 40 the LDS copy is ahead: 12 spheres 1.17 against 1.24 ms; profiles/r05/resident_scalar_ab.txt)
						else if (RESIDENT_SCALAR_SCAN)
							scan_streamed_spheres(spheres, st.origin, st.dir, geometry, s.n_spheres);
						else
							scan_lds<true>(spheres, st.origin, st.dir, primitives, s.n_spheres, 0);
						uint32_t index;
						kind = select_hit(spheres, planes, distance, index);
						fetch_hit<SM>(s, st.origin, st.dir, kind, distance, index, normal, shading, scatter_kind);
					}
					if (!kind)
					{
						RT_HIP_REGION(4); // miss: sky, end of sample
						end_sample(st.throughput * sky(st.dir.y)); // miss (:163-164)
					}
					else
					{
						shade = true;
						// r.at(t): where the scattered ray starts (:122,139).  The lane is done with the old origin — it goes straight
						// into the ray's registers (as a value of its own it was copied there, three moves per scatter, after the hand-out).
						st.origin = ray_at(st.origin, st.dir, distance);
						if (NS > 0)
						{
							RT_HIP_REGION(3); // hit: lookups + normal
							const float4 g = lds_geometry[small_index];
							shading = lds_shading[small_index];
							scatter_kind = __float_as_uint(g.w);
							if (NP > 0 && kind == 2u)
								normal = { g.x, g.y, g.z }; // the plane's normal as it is, not flipped toward the ray (:58)
							else
							{
								// (a branch, not a select: a wave whose hits are all on the ground plane — most of a frame's — skips the
								// normalisation altogether)
								if (NP > 0)
									asm volatile("; hit: some lane hit a sphere" ::: "memory");
								normal = normalize(st.origin - vec3{ g.x, g.y, g.z }); // direction(center, r.at(t)) (:85)
							}
						}
					}
				}


				// ---- hand out items to free lanes (converged): a lane whose chunk just ended with a miss restarts right below ----
				const unsigned long long asking = __builtin_amdgcn_ballot_w64(RT_HIP_IS(lane_free));
				// a free lane becomes the owner of item `item` of the tile at (x0, y0)
				// a free lane starts on chunk `chunk` of the pixel at (lx, ly) of this rank's rows
				// (HALF: `chunk` counts half-chunks of 8 samples; one that lies wholly behind the last sample is empty)
				const auto start_item = [&](uint32_t lx, uint32_t ly, uint32_t chunk)
				{
					const uint32_t item_samples = (HALF && ROLLING) ? q.item_samples : (HALF ? sample_chunk / 2u : sample_chunk);
					if (HALF && !ROLLING && chunk * item_samples >= p.samples_per_pixel)
						return; // (the lane stays free and asks again)
					const uint32_t gy = global_row(ly, p);
					st.chunk_sum = { 0.0f, 0.0f, 0.0f };
					{
						const float fx = static_cast<float>(lx), fy = static_cast<float>(gy);
						if (PINHOLE_ONLY || (!EYE_ONLY && !NEVER_PINHOLE && p.pinhole)) // (wave-uniform: a kernel argument)
						{
							st.base_x = fma(p.ray_d1[0], fx, fma(p.ray_d2[0], fy, p.ray_d0[0]));
							st.base_y = fma(p.ray_d1[1], fx, fma(p.ray_d2[1], fy, p.ray_d0[1]));
							st.base_z = fma(p.ray_d1[2], fx, fma(p.ray_d2[2], fy, p.ray_d0[2]));
							st.base_w = 0.0f;
						}
						else if (EYE_ONLY || p.eye_form) // (wave-uniform)
						{
							st.base_x = fma(p.eye_q1[0], fx, fma(p.eye_q2[0], fy, p.eye_q0[0]));
							st.base_y = fma(p.eye_q1[1], fx, fma(p.eye_q2[1], fy, p.eye_q0[1]));
							st.base_z = fma(p.eye_q1[2], fx, fma(p.eye_q2[2], fy, p.eye_q0[2]));
							st.base_w = fma(p.eye_w1, fx, fma(p.eye_w2, fy, p.eye_w0));
						}
						else
						{
							st.base_x = fx, st.base_y = fy;
							st.base_z = st.base_w = 0.0f;
						}
					}
					st.keys.function_key = pixel_function_key(p.frame_key_a, gy * p.width + lx); // image_view::position_of, image.hpp:155-159
					st.keys.stride = pixel_stride(p.frame_key_b, st.keys.function_key);
					const uint32_t first = chunk * item_samples, end = min(first + item_samples, p.samples_per_pixel);
					st.sample = first;
					st.sample_end = end;
					st.window = sample_counter(st.keys.stride, first);
					st.window_end = sample_counter(st.keys.stride, end);
					RT_HIP_BECOME(lane_restart);
				};
				// a free lane becomes the owner of item `item` of the tile at (x0, y0)
				const auto take_item = [&](uint32_t item, uint32_t x0, uint32_t y0)
				{
					const uint32_t pixel = item & ((1u << q.pixels_log2) - 1u); // chunk-major item order
					const uint32_t chunk = item >> q.pixels_log2;
					const uint32_t lx = x0 + (pixel & (tile_w - 1u));
					const uint32_t ly = y0 + (pixel >> q.tile_w_log2);
					if (lx < p.width && ly < p.local_rows)
						start_item(lx, ly, chunk);
					// else: a pixel outside the frame — the item is empty, ask again next trip
				};
				if (!ROLLING && asking != 0)
				{
					RT_HIP_REGION(6); // hand-out
					const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(asking >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(asking), 0u));
					if (RT_HIP_IS(lane_free))
					{
						slot = next_item + rank;
						if (slot >= items)
							RT_HIP_BECOME(lane_retired);
						else
						{
							RT_HIP_REGION(7); // a lane takes an item: pixel coordinates, stream keys
							take_item(slot, tile_x0, tile_y0);
						}
					}
					next_item += static_cast<uint32_t>(__builtin_popcountll(asking));
				}
				if (ROLLING && asking != 0)
				{
					const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(asking >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(asking), 0u));
					uint32_t want = static_cast<uint32_t>(__builtin_popcountll(asking));
					if (q.lane_cap < 64u) // (wave-uniform) a sparse launch: this wave holds at most lane_cap rays, see choose_queue
					{
						const uint32_t holding = static_cast<uint32_t>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(RT_HIP_IS(lane_trace) || RT_HIP_IS(lane_restart))));
						want = min(want, q.lane_cap > holding ? q.lane_cap - holding : 0u);
					}
					uint32_t served = 0; // asking lanes given an item so far, in rank order
					while (served < want)
					{
						if (block_next == block_end) // this wave's block is used up: on to the one drawn ahead
						{
							if (dry)
								break;
							if (prefetched >= total_items)
							{
								dry = true;
#ifdef RT_HIP_WAVE_CLOCKS
								clock_dry = wall_clock64();
#endif
								break;
							}
							block_next = prefetched;
							block_end = min(prefetched + prefetched_count, total_items);
							prefetched_count = q.block_items;
							prefetched = fetch_items(counters, prefetched_count);
						}
						const uint32_t take = static_cast<uint32_t>(min(static_cast<unsigned long long>(want - served), block_end - block_next)); // >= 1
						if (RT_HIP_IS(lane_free) && rank - served < take) // served <= rank < served + take
						{
							// item -> (pixel in hand-out order, chunk): pixel-major, so that a pixel's chunks are started together
							const unsigned long long item = block_next + (rank - served);
							uint32_t pixel, chunk;
							if ((total_items >> 32) == 0) // (wave-uniform)
							{
								pixel = static_cast<uint32_t>(item) / items_per_pixel;
								chunk = static_cast<uint32_t>(item) - pixel * items_per_pixel;
							}
							else
							{
								pixel = static_cast<uint32_t>(item / items_per_pixel);
								chunk = static_cast<uint32_t>(item - static_cast<unsigned long long>(pixel) * items_per_pixel);
							}
							const uint32_t row = pixel / p.width; // bottom row first
							slot = pixel;
							start_item(pixel - row * p.width, p.local_rows - 1u - row, chunk);
						}
						served += take;
						block_next += take;
					}
					if (dry && RT_HIP_IS(lane_free))
						RT_HIP_BECOME(lane_retired);
				}
				const bool queue_empty = __builtin_amdgcn_ballot_w64(!RT_HIP_IS(lane_retired)) == 0;
				if (NS == -1)
				{
					if (__syncthreads_and(queue_empty)) // the four waves leave together
						break;
				}
				else if (queue_empty)
					break;

				const bool restart = RT_HIP_IS(lane_restart);
				// (Every trip of a live wave has lanes here: no vote in front of the tail.  A restarting lane's stream already
				// stands at its sample's window — next_sample / start_item put it there — so both kinds of lane draw from
				// st.counter and nothing is selected or multiplied per trip.)
				{
					RT_HIP_REGION(8); // the fused tail: one generator step
					// ONE generator step for every lane (contract v4): a scattering lane's random<vec3>() (random_unit_vector,
					// random.hpp:57-66: the third word follows below) or random<float>(), a restarting lane's random<vec2>() (the
					// pixel jitter, :189).  As numerators k of u = k * 2^-24: the scaling is folded into the jitter's constants and
					// cancels in the unit vector.
					// (The step advances EVERY lane's counter in place — a free or retired lane's too, whose counter nobody reads before
					// start_item sets it.  Nothing below needs the value it had: "the stream stood at 0" reads "it now stands at one
					// stride".  Advanced in a copy and written back by the lanes that shade or restart, it cost a register move per trip.)
					// A restarting lane draws from the start of its sample's window, a scattering lane goes on where it stands: one select.
					// (Rounds 4-5 had next_sample() copy the window into the counter instead — a move per miss, and three more per trip
					// that carried the two versions of the counter through the hit / miss join.)
					st.counter = restart ? st.window : st.counter;
#ifdef RT_HIP_FAST_BUILD
					uint32_t word = next_step_word(st.counter, st.keys);
#else
					// (the addition written out with its result tied to its operand's register: hipcc otherwise adds into a second
					// register and moves the sum back at the end of the trip)
					asm("v_add_u32 %0, %0, %1" : "+v"(st.counter) : "v"(st.keys.stride));
					uint32_t word = step_word_at(st.counter, st.keys);
#endif
					uint32_t word_a = word * step_mul_a, word_b = word * step_mul_b;
					float d0 = step_numerator(word_a);
					float d1 = step_numerator(word_b);
					vec3 toward; // the vector whose direction the new ray takes (set on both paths below)
					// dot(toward, toward).  (sm's dielectric_scatter does not normalise what it returns: its lanes leave the 1 here,
					// whose reciprocal square root is 1, and x * 1 is x for every x.)
					float toward2;
					if (SM)
						toward2 = 1.0f;
					if (shade && SM && scatter_kind == scatter_dielectric)
					{
						// dielectric_scatter, sm_ray_tracer.cpp:181-219 (refract :161-172, schlick :174-179); shading.w = the
						// material's reflectivity, which the reference uses as the index of refraction.  ONE draw.
						const float refl = shading.w;
						const vec3 d = st.dir;
						const float dn = dot(d, normal);
						const float k = 2.0f * dn;
						const vec3 reflected = { fma(-k, normal.x, d.x), fma(-k, normal.y, d.y), fma(-k, normal.z, d.z) };
						const float len = sqrt_rn(dot(d, d)); // (== __builtin_sqrtf, bit for bit: contract.hpp)
						const bool inside = dn > 0.0f;
						const vec3 outward = inside ? vec3{ -normal.x, -normal.y, -normal.z } : normal;
						const float eta = inside ? refl : rcp_rn(refl);
						const float cosine = inside ? (refl * dn) / len : (-dn) / len;
						float reflect_prob = 1.0f;
						vec3 refracted = { 0.0f, 0.0f, 0.0f };
						const float cos_i = -dot(d, outward);
						const float sin2_t = (eta * eta) * fma(-cos_i, cos_i, 1.0f);
						if (!(sin2_t > 1.0f)) // refract() returned true
						{
							const float cos_t = sqrt_rn(1.0f - sin2_t);
							const float kk = fma(eta, cos_i, -cos_t);
							refracted = { fma(kk, outward.x, eta * d.x), fma(kk, outward.y, eta * d.y), fma(kk, outward.z, eta * d.z) };
							float r0 = (1.0f - refl) / (1.0f + refl);
							r0 = r0 * r0;
							const double x = static_cast<double>(1.0f - cosine); // pow(1 - cosine, 5) in double, by multiplication
							const double x2 = x * x;
							const double x5 = (x2 * x2) * x;
							reflect_prob = static_cast<float>(static_cast<double>(r0) + static_cast<double>(1.0f - r0) * x5);
						}
						toward = (d0 * random_scale < reflect_prob) ? reflected : refracted;
						st.throughput = st.throughput * vec3{ shading.x, shading.y, shading.z };
						const bool dead = mode == lane_trace; // no bounce left: the next trace() call would return {} at :157-158
						mode--;
						if (dead)
							end_sample({ 0.0f, 0.0f, 0.0f });
					}
					else if (shade)
					{
						RT_HIP_REGION(9); // scatter: third draw, unit vector, new direction
						uint32_t word_c = word * step_mul_c;
						while (((word_a | word_b | word_c) >> 8) == 0u) // `if (p == zero) continue` (random.hpp:61-62): the next step
						{
							word = next_step_word(st.counter, st.keys);
							word_a = word * step_mul_a, word_b = word * step_mul_b, word_c = word * step_mul_c;
							d0 = step_numerator(word_a);
							d1 = step_numerator(word_b);
						}
						const vec3 u = normalize_unit_cube_numerators({ d0, d1, step_numerator(word_c) });
						vec3 scatter = normal + u; // lambert (:117)
						bool absorbed = false;
						if (scatter_kind == scatter_metal)
						{
							RT_HIP_REGION(5); // metal: normalise the incoming direction, reflect, spread by the roughness
							// reflect(normalize(r.direction), n) + roughness * random_unit_vector() (:133-134, common.hpp:100-103)
							const vec3 v = normalize(st.dir);
							const float k = 2.0f * dot(v, normal);
							const vec3 reflected = { fma(-k, normal.x, v.x), fma(-k, normal.y, v.y), fma(-k, normal.z, v.z) };
							scatter = { fma(shading.w, u.x, reflected.x), fma(shading.w, u.y, reflected.y), fma(shading.w, u.z, reflected.z) };
							absorbed = dot(scatter, normal) <= 0.0f; // (:135-136)
							toward2 = dot(scatter, scatter);
						}
						else
						{
							toward2 = dot(scatter, scatter);
							// `if (scatter.approx_zero()) scatter = normal` (:118-119): all three components within 1e-6 of zero.  Then the
							// sum of their squares is at most 3e-12 (and a hair: three roundings) — a single comparison every lane can
							// afford; the three of the rule itself run behind a vote that practically never fires.
							if (__builtin_amdgcn_ballot_w64(toward2 <= 3.0001e-12f) != 0)
							{
								asm volatile("; lambert: some lane's scatter vector is next to zero" ::: "memory");
								if (__builtin_fabsf(scatter.x) <= approx_zero_epsilon && __builtin_fabsf(scatter.y) <= approx_zero_epsilon && __builtin_fabsf(scatter.z) <= approx_zero_epsilon)
								{
									scatter = normal;
									toward2 = dot(scatter, scatter);
								}
							}
						}
						st.throughput = st.throughput * vec3{ shading.x, shading.y, shading.z }; // attenuation * trace(...) (:171)
						toward = scatter;
						// absorbed, or the next trace() call would return {} at :157-158 (`if (!(max_bounces--))`: the count is kept
						// as the bounces still allowed after the segment in flight, and only a bounce touches it)
						const bool dead = absorbed || mode == lane_trace;
						mode--; // (a dead lane's mode is set by end_sample right below)
						if (dead)
						{
							RT_HIP_REGION(12); // absorbed or out of bounces: the sample is worth nothing
							end_sample({ 0.0f, 0.0f, 0.0f });
						}
					}
					else if (restart)
					{
						RT_HIP_REGION(10); // restart: primary ray
						// worker lambda :189-193 — jittered position, un-project to near and far, build the primary ray
						float jx = d0, jy = d1;
						// sample 0 goes through the pixel centre (0.5 = 2^23 * 2^-24) and draws nothing (:189).  Its window is the one
						// that starts at stream position 0.  One restart in spp is such a sample, and a wave meets them all in its first
						// trips: a vote keeps the two selects and the rewind out of every other trip.
						const bool at_sample_0 = st.window == 0u;
						if (__builtin_amdgcn_ballot_w64(at_sample_0) != 0)
						{
							asm volatile("; restart: some lane is at its pixel's sample 0" ::: "memory");
							if (at_sample_0)
							{
								jx = jy = 0x1.0p23f;
								st.counter = 0u; // (the sample draws nothing: back to the window's start)
							}
						}
						if (PINHOLE_ONLY || (!EYE_ONLY && !NEVER_PINHOLE && p.pinhole)) // (wave-uniform: a kernel argument)
						{
							// rt's camera: the near-to-far vector from the pixel's base and the jitter, the near point from it (contract
							// v4; constants from the host).  The LDS / big-scene kernels carry both forms; a scalar-register kernel is
							// built for ONE of them (GC): as kernel arguments the two sets of scalars together would cost its loop, which
							// lives on its scalar registers, a spill per lane mask.
							toward = { fma(p.ray_j1[0], jx, fma(p.ray_j2[0], jy, st.base_x)), fma(p.ray_j1[1], jx, fma(p.ray_j2[1], jy, st.base_y)),
									   fma(p.ray_j1[2], jx, fma(p.ray_j2[2], jy, st.base_z)) };
							st.origin = { p.ray_eye[0] + toward.x, p.ray_eye[1] + toward.y, p.ray_eye[2] + toward.z }; // (toward = near - eye)
						}
						else if (EYE_ONLY || p.eye_form) // (wave-uniform)
						{
							// a perspective matrix that is no pinhole's in binary32 — a camera that is not axis-aligned: rounding noise in
							// its w row.  Every near-to-far line passes through the eye E; the near point relative to it (times N.w) and
							// N.w itself move with the jitter like the pinhole's vector does: ONE reciprocal per sample, no far point
							// (round 4: two reciprocals and sixteen multiply-adds for the two points).
							toward = { fma(p.eye_jq1[0], jx, fma(p.eye_jq2[0], jy, st.base_x)), fma(p.eye_jq1[1], jx, fma(p.eye_jq2[1], jy, st.base_y)),
									   fma(p.eye_jq1[2], jx, fma(p.eye_jq2[2], jy, st.base_z)) };
							const float ws = fma(p.eye_jw1, jx, fma(p.eye_jw2, jy, st.base_w));
							// (wave-uniform) the host has shown ws in the band and N.w F.w > 0 for the whole frame — every frame of a camera
							// that looks at its scene; the scalar-register kernels are built for such frames only (choose_kernel), anything
							// else — near and far points on different sides of w = 0, a w at the band's ends — goes through the resident kernel
							const bool plain = EYE_ONLY || p.eye_form == 2u;
							float inv;
							if (plain)
								inv = rcp_in_band(ws);
							else
							{
								asm volatile("; restart: the guarded reciprocal" ::: "memory");
								inv = rcp_rn(ws);
							}
							st.origin = { fma(toward.x, inv, p.eye_e[0]), fma(toward.y, inv, p.eye_e[1]), fma(toward.z, inv, p.eye_e[2]) };
							// far - near = s N' |Z.w| / (N.w F.w): s N' unless the near and far points lie on different sides of w = 0
							if (!plain && __builtin_amdgcn_ballot_w64(ws * (ws + p.eye_zws) < 0.0f) != 0)
							{
								asm volatile("; restart: some lane's near and far points straddle w = 0" ::: "memory");
								if (ws * (ws + p.eye_zws) < 0.0f)
									toward = { -toward.x, -toward.y, -toward.z };
							}
						}
						else
						{
							// no finite eye (an orthographic frustum): un-project to depth 0 and depth 1 (camera.hpp:42-48) in homogeneous form, N and F.  The
							// near point needs its division; the direction does not — far / F.w - near / N.w is F N.w - N F.w over
							// N.w F.w, and normalize() removes a positive factor: one reciprocal per sample (round 4: two).
							const float px = fma(jx, random_scale, st.base_x); // == x + jx * 2^-24: the product is exact
							const float py = fma(jy, random_scale, st.base_y);
							const float ndc_x = fma(px, p.sx, -1.0f);
							const float ndc_y = fma(py, p.neg_sy, 1.0f);
							float N[4], F[4];
#pragma unroll
							for (int r = 0; r < 4; r++)
							{
								N[r] = fma(p.mx[r], ndc_x, fma(p.my[r], ndc_y, p.k_near[r]));
								F[r] = fma(p.mx[r], ndc_x, fma(p.my[r], ndc_y, p.k_far[r]));
							}
							const float inv_wn = rcp_rn(N[3]);
							st.origin = { N[0] * inv_wn, N[1] * inv_wn, N[2] * inv_wn };
							toward = { fma(F[0], N[3], -(N[0] * F[3])), fma(F[1], N[3], -(N[1] * F[3])), fma(F[2], N[3], -(N[2] * F[3])) };
							if (N[3] * F[3] < 0.0f)
								toward = { -toward.x, -toward.y, -toward.z };
						}
						toward2 = dot(toward, toward);
						st.throughput = { 1.0f, 1.0f, 1.0f };
						mode = lane_trace + (p.max_bounces - 1u); // tracing, max_bounces - 1 bounces to go after the primary segment
					}
					if (shade || restart)
					{
						RT_HIP_REGION(11); // normalise the new direction: toward * inv_sqrt(dot(toward, toward))
						const float inv = inv_sqrt_rn(toward2);
						st.dir = toward * inv;
					}
				}
			}

			// ---- one tile per wave: fold and write its pixels now (rolling tiles were folded as their last item came in) ----
			if (!ROLLING)
			{
				// the slots are private to the wave: no workgroup barrier (the other three waves may still be tracing, and
				// a wave that has written its pixels should free its slot), only ordering against this wave's own writes
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				fold_tile(slots, tile_x0, tile_y0);
			}
			if (!SCALAR_SEGMENTS)
			{
				wave_segments = lane_segments;
#pragma unroll
				for (int offset = 32; offset > 0; offset >>= 1)
					wave_segments += __shfl_down(wave_segments, offset, 64);
			}
			add_segments(counters, wave_segments);
#ifdef RT_HIP_WAVE_CLOCKS
			if (ROLLING && lane == 0)
			{
				const uint32_t id = blockIdx.x * 4u + wave;
				if (id < device_counters::clocked_waves)
				{
					counters->wave_clocks[id][0] = clock_start;
					counters->wave_clocks[id][1] = clock_dry;
					counters->wave_clocks[id][2] = wall_clock64();
				}
			}
#endif
#ifdef RT_HIP_REGION_COUNTERS
			if (lane == 0)
				for (unsigned i = 0; i < device_counters::regions; i++)
				{
					atomicAdd(&counters->region_runs[i], static_cast<unsigned long long>(region_runs[i]));
					atomicAdd(&counters->region_lanes[i], static_cast<unsigned long long>(region_lanes[i]));
				}
#endif
		}

#ifndef RT_HIP_FAST_BUILD
		// ---- preview: reference src/renderers/rasterizer.cpp:24-85 ----------------------------------------------------
		// One thread per pixel, one ray through the pixel centre; every lane of a wave walks the same primitive, so the
		// reads below are wave-uniform (scalar loads through the constant cache / L2).  4 bytes written per pixel and a
		// few dozen flops per primitive: a frame of a small scene is bound by the store and the launch, nothing to tune.
		__global__ __launch_bounds__(block_threads) void preview_frame(const frame_params p,
																	   const device_scene s,
																	   uint32_t* __restrict__ out_rgba,
																	   float* __restrict__ out_rgb,
																	   device_counters* __restrict__ counters)
		{
			const uint32_t lx = blockIdx.x * 64u + (threadIdx.x & 63u);
			const uint32_t ly = blockIdx.y * (block_threads / 64u) + (threadIdx.x >> 6);
			const bool alive = lx < p.width && ly < p.local_rows;
			const uint32_t gy = global_row(alive ? ly : 0u, p);

			// :30-32 — near and far points of the pixel centre
			const float ndc_x = fma(static_cast<float>(lx) + 0.5f, p.sx, -1.0f);
			const float ndc_y = fma(static_cast<float>(gy) + 0.5f, p.neg_sy, 1.0f);
			float near_row[4], far_row[4];
#pragma unroll
			for (int r = 0; r < 4; r++)
			{
				near_row[r] = fma(p.mx[r], ndc_x, fma(p.my[r], ndc_y, p.k_near[r]));
				far_row[r] = fma(p.mx[r], ndc_x, fma(p.my[r], ndc_y, p.k_far[r]));
			}
			const float inv_wn = rcp_rn(near_row[3]), inv_wf = rcp_rn(far_row[3]);
			const vec3 near_pos = { near_row[0] * inv_wn, near_row[1] * inv_wn, near_row[2] * inv_wn };
			const vec3 far_pos = { far_row[0] * inv_wf, far_row[1] * inv_wf, far_row[2] * inv_wf };
			const vec3 delta = far_pos - near_pos;
			const vec3 dir = normalize(delta);					// vec3::direction(near, far) (:39)
			float dist = sqrt_rn(dot(delta, delta)) + 1.0f;		// max_dist + 1 (:33,35)
			bool hit = false;
			uint32_t material = 0;
			vec3 hit_pos = { 0.0f, 0.0f, 0.0f };
			vec3 normal = { 0.0f, 1.0f, 0.0f };					// vec3::constants::up (:38)

			// hit_tests (:41-61): a candidate replaces the current one only if strictly nearer; planes, boxes, spheres
			for (uint32_t i = 0; i < s.n_planes; i++)
			{
				const float4 g = s.primitive_geometry[s.n_spheres + i];
				float t;
				if (hits_plane(near_pos, dir, { g.x, g.y, g.z }, g.w, t) && t < dist)
				{
					dist = t, hit = true, material = s.plane_material[i];
					hit_pos = ray_at(near_pos, dir, t);
					normal = { g.x, g.y, g.z };
				}
			}
			for (uint32_t i = 0; i < s.n_boxes; i++)
			{
				const float4 lo = s.box_bounds[2u * i], hi = s.box_bounds[2u * i + 1u];
				float t;
				if (hits_box(near_pos, dir, { lo.x, lo.y, lo.z }, { hi.x, hi.y, hi.z }, t) && t < dist)
				{
					dist = t, hit = true, material = __float_as_uint(lo.w);
					hit_pos = ray_at(near_pos, dir, t); // the normal stays what it was: the reference sets none for a box (:56-59)
				}
			}
			for (uint32_t i = 0; i < s.n_spheres; i++)
			{
				const float4 g = s.primitive_geometry[i];
				float t;
				if (hits_sphere(near_pos, dir, { g.x, g.y, g.z }, g.w, t) && t < dist)
				{
					dist = t, hit = true, material = s.sphere_material[i];
					hit_pos = ray_at(near_pos, dir, t);
					normal = normalize(hit_pos - vec3{ g.x, g.y, g.z });
				}
			}

			vec3 colour;
			if (hit)
			{
				// min(0.25 + lambert(N, direction to the eye, albedo) * 0.75, 1) (:66-73); lambert = (L . N) * albedo (:13-20)
				const float4 albedo = s.material_albedo[material];
				const float k = dot(normalize(near_pos - hit_pos), normal);
				const vec3 v = { (k * albedo.x) * 0.75f + 0.25f, (k * albedo.y) * 0.75f + 0.25f, (k * albedo.z) * 0.75f + 0.25f };
				colour = { select_min(v.x, 1.0f), select_min(v.y, 1.0f), select_min(v.z, 1.0f) };
			}
			else
			{
				// lerp(sky_start, sky_end, y / (H - 1)) (:76-80).  Both sky colours are built from integers through
				// colour's clamping constructor (colour.hpp:72-91), which turns every non-zero byte into 1.0: the sky is
				// white, and NaN for a one-row frame (0 / 0), which packs to black.
				const float t = static_cast<float>(gy) / static_cast<float>(p.height - 1u);
				const float c = fma(1.0f - 1.0f, t, 1.0f);
				colour = { c, c, c };
			}
			if (alive)
			{
				const size_t o = static_cast<size_t>(p.frame_rows ? gy : ly) * p.width + lx;
				if (out_rgb)
				{
					out_rgb[o * 3 + 0] = colour.x;
					out_rgb[o * 3 + 1] = colour.y;
					out_rgb[o * 3 + 2] = colour.z;
				}
				out_rgba[o] = pack_rgba8888(colour);
			}
			// one closest-hit query per pixel: the count is known, so a single store stands in for the per-wave atomics of the
			// render kernels (at one ray per pixel their serialisation on one address would be most of the launch: 0.40 vs 0.02 ms)
			if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
				counters->segments[0] = static_cast<unsigned long long>(p.local_rows) * p.width;
		}

		// ---- multi-GPU assemble: rank-major compact stripes -> frame ------------------------------------------------
		// `width` counts 32-bit words per row: pixels for the RGBA8888 frame, 3 x pixels for the float mean.
		// Rows of ranks below `first_rank` are left alone (the root's own stripes, when its kernel has already stored them
		// straight into the frame).  TO_HOST: the frame is the caller's page-locked back buffer — system-scope stores, written
		// through at once, so that the words cross PCIe while the kernel runs instead of behind its end-of-kernel write-back
		// (finish_pixel has the measurement).
		template <bool TO_HOST>
		__global__ __launch_bounds__(block_threads) void assemble_stripes(uint32_t width,
																		  uint32_t height,
																		  uint32_t world,
																		  uint32_t stripe_rows,
																		  uint32_t padded_local_rows,
																		  uint32_t first_rank,
																		  const uint32_t* __restrict__ gathered,
																		  uint32_t* __restrict__ frame)
		{
			const uint32_t x = blockIdx.x * block_threads + threadIdx.x;
			const uint32_t y = blockIdx.y;
			if (x >= width || y >= height)
				return;
			const uint32_t stripe = y / stripe_rows;
			const uint32_t rank = stripe % world;
			if (rank < first_rank) // (block-uniform: a block lies within one row)
				return;
			const uint32_t local_row = (stripe / world) * stripe_rows + (y % stripe_rows);
			const uint32_t word = gathered[(static_cast<size_t>(rank) * padded_local_rows + local_row) * width + x];
			if (TO_HOST)
				__hip_atomic_store(&frame[static_cast<size_t>(y) * width + x], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			else
				frame[static_cast<size_t>(y) * width + x] = word;
		}

#endif // !RT_HIP_FAST_BUILD

		template <int NS, bool SM, int NP = 0, bool GC = false>
		void launch_queue_sm(const frame_params& frame,
							 const queue_params& queue,
							 const small_scene& small,
							 const device_scene& scene,
							 dim3 grid, // small scenes: the grid of tiles; big scenes: x = the most workgroups the tiles can occupy
							 size_t lds_bytes,
							 uint32_t* d_rgba8,
							 float* d_rgb_f32,
							 device_counters* d_counters,
							 const rolling_buffers& rolling,
							 uint32_t compute_units,
							 launch_cache& cache,
							 hipStream_t stream)
		{
			if (NS < 0)
			{
				// persistent launch: exactly what the device keeps resident (surplus workgroups would only find the queue dry).
				// The answer is remembered per context (= per device and host thread of use), per kernel and LDS size.
#ifdef RT_HIP_FAST_BUILD
				constexpr bool fast_arithmetic = true;
#else
				constexpr bool fast_arithmetic = false;
#endif
				const bool sub_chunk_items = !SM && queue.halves;
				launch_cache::entry& known = cache.persistent[launch_cache::slot(NS == -1 ? 0u : (NS == -2 ? 1u : 2u), SM, fast_arithmetic, sub_chunk_items)];
				if (known.lds_bytes != lds_bytes || known.per_cu < 1)
				{
					int per_cu = 0;
					hipError_t asked;
					if constexpr (!SM)
						asked = sub_chunk_items ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, render_queue<NS, SM, true>, static_cast<int>(block_threads), lds_bytes)
												: hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, render_queue<NS, SM, false>, static_cast<int>(block_threads), lds_bytes); // (persistent kernels: never with scalar-register planes)
					else
						asked = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, render_queue<NS, SM>, static_cast<int>(block_threads), lds_bytes);
					if (asked != hipSuccess || per_cu < 1)
						per_cu = 4;
					// Not more workgroups than the kernels were compiled for (5 waves per SIMD = 5 workgroups per CU), even when the
					// register allocation of a build happens to leave room for a sixth (round 4 saw 6 144 waves instead of 5 120
					// on one build): the launch's shape should not depend on that.  Measured, it makes no difference either way
					// (config 5: 5.479 against 5.476 s, profiles/r04/persistent_waves_ab.txt).
					per_cu = std::min(per_cu, NS == -3 ? RT_HIP_WAVES_DENSE : RT_HIP_PERSISTENT_WAVES_CAP);
					(void)hipGetLastError();
					known.lds_bytes = lds_bytes;
					known.per_cu = per_cu;
				}
				grid = dim3(std::min(grid.x, compute_units * static_cast<uint32_t>(known.per_cu)));
			}
			if constexpr (!SM) // (the sm table keeps whole chunks: one set of kernels fewer to build)
			{
				if (queue.halves)
				{
					hipLaunchKernelGGL((render_queue<NS, SM, true, NP, GC>), grid, dim3(block_threads), lds_bytes, stream, frame, queue, small, scene, scene.primitive_geometry, d_rgba8, d_rgb_f32, d_counters, rolling.item_sums, rolling.pixel_done);
					return;
				}
			}
			hipLaunchKernelGGL((render_queue<NS, SM, false, NP, GC>), grid, dim3(block_threads), lds_bytes, stream, frame, queue, small, scene, scene.primitive_geometry, d_rgba8, d_rgb_f32, d_counters, rolling.item_sums, rolling.pixel_done);
		}

		template <int NS, int NP = 0, bool GC = false>
		void launch_queue(bool sm,
						  const frame_params& frame,
						  const queue_params& queue,
						  const small_scene& small,
						  const device_scene& scene,
						  dim3 grid,
						  size_t lds_bytes,
						  uint32_t* d_rgba8,
						  float* d_rgb_f32,
						  device_counters* d_counters,
						  const rolling_buffers& rolling,
						  uint32_t compute_units,
						  launch_cache& cache,
						  hipStream_t stream)
		{
#ifndef RT_HIP_FAST_BUILD // (the API refuses RT_HIP_FLAG_FAST together with RT_HIP_FLAG_SM_MATERIALS)
			if (sm)
				launch_queue_sm<NS, true, NP, GC>(frame, queue, small, scene, grid, lds_bytes, d_rgba8, d_rgb_f32, d_counters, rolling, compute_units, cache, stream);
			else
#endif
				launch_queue_sm<NS, false, NP, GC>(frame, queue, small, scene, grid, lds_bytes, d_rgba8, d_rgb_f32, d_counters, rolling, compute_units, cache, stream);
		}
	}

#ifndef RT_HIP_FAST_BUILD
	uint32_t choose_kernel(const device_scene& scene, uint32_t flags, uint32_t samples_per_pixel, bool perspective, uint64_t pixels)
	{
		const uint32_t primitives = scene.n_spheres + scene.n_planes;
		if (flags & RT_HIP_FLAG_FORCE_STREAMED)
			return RT_HIP_KERNEL_STREAMED;
		if (flags & RT_HIP_FLAG_FORCE_TILED)
			return RT_HIP_KERNEL_TILED;
		// up to 8 primitives: at least one sphere, at most three planes (round 4: neither a plane nor a camera whose w varies
		// over the frame pushes a scene off this kernel any more)
		// ... through a camera with an eye (`perspective`: the pinhole or the plain eye form; the scalar-register kernels are built for those)
		// ... and planes whose normals are of ordinary size (device_scene::planes_tame)
		if (!(flags & RT_HIP_FLAG_FORCE_RESIDENT) && perspective && scene.n_spheres >= 1 && scene.n_planes <= scalar_max_planes && primitives <= scalar_max_spheres && (scene.n_planes == 0 || scene.planes_tame))
			return RT_HIP_KERNEL_SMALL;
		// The LDS-resident kernel (one tile per wave) up to streamed_from_primitives (kernels.hpp has the measurements), or whatever
		// its LDS can hold when forced; beyond that a trip is a long scan and the rolling hand-out of the big-scene kernels wins.
		const uint32_t staged = (scene.n_spheres < resident_scalar_scan_from ? scene.n_spheres : 0u) + scene.n_planes; // what the resident kernel keeps in LDS
		// (Scenes beyond its LDS capacity only in frames that fill the device — 4M samples, 64 for every lane it holds: in a small
		// frame of a big scene every wave is a sparse one, and the streamed kernel scans those with all 64 lanes per ray.)
		const bool fits = primitives <= resident_max_primitives || (primitives <= streamed_from_primitives && pixels * samples_per_pixel >= (1ull << 22));
		if (staged <= resident_max_primitives && ((flags & RT_HIP_FLAG_FORCE_RESIDENT) || fits))
			return RT_HIP_KERNEL_RESIDENT;
		// Big scenes: the scalar-streamed kernel (no staging, no barriers).  Rounds 1-2 chose the LDS-tiled kernel below
		// 32 samples per pixel, where it was 2 % ahead; since the group prefetch, the cooperative scan of sparse waves and
		// the one-sample items the streamed kernel is 10-30 % ahead at every sample count from 1 to 24 and every size from
		// 1 100 to 100 000 spheres (profiles/r03/tiled_vs_streamed.txt).  The tiled kernel stays behind its flag.
		(void)samples_per_pixel;
		return RT_HIP_KERNEL_STREAMED;
	}

	queue_params choose_queue(uint32_t samples_per_pixel, uint32_t width, uint32_t local_rows, bool big_scene, bool host_frame, int half_chunks, uint32_t primitives, bool sparse_launch)
	{
		queue_params q{};
		q.chunks = (samples_per_pixel + sample_chunk - 1u) / sample_chunk; // K chunks per pixel
		uint32_t pixels_log2;
		if (big_scene)
		{
			// rolling items: no tiles at all (the fields below describe the frame as 1 x 1 tiles and are not used); a wave
			// draws 8 items at a time, one block ahead — small enough that a wave sits on at most 15 reserved items when
			// the sequence runs dry, large enough that the counter sees one atomic per wave every few trips
			pixels_log2 = 0;
			q.block_items = 8u;
			q.lane_cap = 64u;
			q.sparse_rays = sparse_wave_rays;
			// Sub-chunk items.  A trip of a big-scene wave costs the same with one lane holding a ray as with 64, and from the
			// moment the launch-wide sequence runs dry every lane still owes the rest of its item: with whole chunks the
			// last 10 % of config 5's launch ran on thinning waves (4.9 % of all wave-time after the waves' retirement alone,
			// profiles/r03/config5_streamed/wave_tail_items_sparse.txt), and its 8-way share — 2.6 chunks per lane — took
			// twice its share of the time.  Items may be ANY run of consecutive samples if every sample's VALUE is handed
			// over instead of a chunk's sum (16 bytes per sample through HBM: nothing next to a scan of the scene per path
			// segment) and the lane that brings a pixel's last item adds them up as the contract says.  How small: every item
			// costs an arrival (an atomic in HBM: the device does ~0.66 G of them per second, profiles/r03/item_sweep.txt),
			// which stays in the shadow of the tracing while samples-per-item x primitives >= 8192 — one sample per item from
			// 8192 primitives upwards, eight at 1025.  Measured: config 5 5.55 -> 5.10 s, its 1/8 share 1203 -> 636 ms;
			// 30 000 x 64 spp 1756 -> 1616; 10 000 x 32 spp 276 -> 244; 2 000 x 64 spp 96.4 -> 90.5; 1 025 x 64 spp 49.7 -> 47.1.
			q.item_samples = sample_chunk;
			if (half_chunks && samples_per_pixel > 1u && primitives)
			{
				uint32_t smallest = 1u;
				while (smallest < sample_chunk && static_cast<uint64_t>(smallest) * primitives < 8192u)
					smallest *= 2u;
				// (a sample's slot is 16 bytes: frames whose samples would need more than 8 GiB keep whole chunks)
				if (static_cast<uint64_t>(width) * local_rows * samples_per_pixel * 16u <= (8ull << 30))
					q.item_samples = smallest;
				if (half_chunks == 2 && q.item_samples == sample_chunk)
					q.item_samples = sample_chunk / 2u;
#ifdef RT_HIP_QUEUE_KNOBS
				if (const char* knob = std::getenv("RT_HIP_ITEM_SAMPLES")) // experiment builds only (tools/gpu_item_sweep.py)
					q.item_samples = static_cast<uint32_t>(std::atoi(knob));
#endif
				q.halves = q.item_samples < sample_chunk ? 1u : 0u;
#ifdef RT_HIP_QUEUE_KNOBS
				if (const char* knob = std::getenv("RT_HIP_BLOCK_ITEMS"))
					q.block_items = static_cast<uint32_t>(std::atoi(knob));
#endif
			}
			// Sparse launches of the streamed kernel.  A trip costs one sequential scan of the scene whether the wave holds 64
			// rays or one; the cooperative scan (scan_spheres_together) costs a wave about 1/40 of that PER RAY.  At the end of a
			// full launch, where the device is busy, it pays up to 8 rays (sparse_wave_rays: an A/B of round 3); in a launch
			// that cannot fill the device at all — fewer work items than 32 per wave it can hold — it pays all the way:
			// the launch is spread THIN, every wave taking at most ceil(items / waves) rays at a time, and scans
			// cooperatively throughout (profiles/r03/sparse_launch.txt).
			if (sparse_launch)
			{
				constexpr uint64_t launch_waves = 256ull * 4ull * 5ull;
				const uint64_t items = static_cast<uint64_t>(width) * local_rows * (q.halves ? (samples_per_pixel + q.item_samples - 1u) / q.item_samples : q.chunks);
				const uint64_t per_wave = (items + launch_waves - 1u) / launch_waves;
				if (per_wave <= 32u)
				{
					q.lane_cap = static_cast<uint32_t>(std::max<uint64_t>(per_wave, 1u));
					q.block_items = std::min(q.block_items, q.lane_cap);
					q.sparse_rays = std::max(q.sparse_rays, q.lane_cap);
				}
			}
		}
		else
		{
			// one tile per wave.  A wave lives as long as its longest lane and the launch ends with about one wave lifetime
			// of tail, so waves should be short — but with fewer than two items per lane the lanes of a wave end at very
			// different times.  Measured on the headline frame and on its 1/2, 1/4, 1/8 shares at 256, 64 and 16 spp
			// (profiles/r01/queue_shape_sweep.txt): 128 items per wave, and 64 when that would give fewer than 6 x 8192
			// waves (8192 = what the device holds at a time), win or tie every case.
			pixels_log2 = 7; // 128 pixels = 16 x 8
			while (pixels_log2 > 2 && (q.chunks << pixels_log2) > 128u)
				pixels_log2--;
			const uint64_t pixels = static_cast<uint64_t>(width) * local_rows;
			if (pixels_log2 > 2 && (pixels >> pixels_log2) < 49152u)
				pixels_log2--;
		}
		// Half-chunks (render_queue<.., HALF>): a launch that has only a few chunks per lane of the device ends unevenly —
		// a rank's 1/8 share of a 64-spp frame holds two per lane and took 0.21 ms for 0.09 ms of work; with 8-sample items
		// 0.13-0.14 (profiles/r03/chunk_probe.txt: 1/4 share -11 %, nothing from eight chunks per lane upwards, where the
		// parking would only cost).  64 half-chunks per wave, one per lane; tiles of at least four pixels.
		// (Not for pixels of ONE chunk, forced aside: their tiles would hold half as many pixels and the wave's fold — a
		// division and three square roots per pixel — runs on half its lanes: 1080p x 16 spp 0.345 against 0.323 ms.  Between
		// four and seven chunks per lane the gain fades: 800 x 600 x 64 spp -16 %, 1280 x 720 x 64 spp +16 %;
		// profiles/r03/half_threshold.txt.)
		if (half_chunks && !big_scene && samples_per_pixel > sample_chunk / 2u && q.chunks <= 16u)
		{
			constexpr uint64_t resident_lanes = 256ull * 4ull * 8ull * 64ull; // an MI355X at 8 waves per SIMD
			if (half_chunks == 2 || (q.chunks >= 2u && static_cast<uint64_t>(width) * local_rows * q.chunks < 5ull * resident_lanes))
			{
				q.halves = 1u;
				pixels_log2 = 2u;
				while (pixels_log2 < 7u && ((2u * q.chunks) << (pixels_log2 + 1u)) <= 64u)
					pixels_log2++;
			}
		}
		// frames of headline size and beyond at 256 spp: 16 pixels (256 items) per wave beat 8 — half as many waves to start
		// and to fold (HBM: 2.62 against 2.63 ms at 1080p, 10.4 against 10.6 at 4K; profiles/r03/tile_shapes.txt) — while a
		// half frame still prefers 8 (1.39 against 1.34)
		if (!big_scene && !q.halves && pixels_log2 == 3u && (q.chunks << 4u) <= 256u && ((static_cast<uint64_t>(width) * local_rows) >> 4u) >= 98304u)
			pixels_log2 = 4u;
		q.pixels_log2 = pixels_log2;
		q.tile_w_log2 = (pixels_log2 + 1u) / 2u; // 16x8, 8x8, 8x4, 4x4, 4x2, 2x2, 2x1, 1x1
		if (host_frame && !big_scene && q.halves)
			q.tile_w_log2 = std::min(pixels_log2, 4u); // (rows as wide as the tile allows: see below)
		else if (host_frame && !big_scene)
		{
			// The finished pixels of a tile leave the wave as one store per tile, a row fragment of tile_w pixels per tile
			// row; into page-locked host memory every fragment is a PCIe write.  Fragments of 8 and 16 bytes (2 x 2, 4 x 4,
			// 4 x 2 tiles) cost nothing in HBM and a lot over PCIe as soon as there are many of them per microsecond — a 1/8
			// share of the headline frame took 0.60 ms as 2 x 2 tiles against 0.35 ms into HBM, 0.37 ms as 4 x 1; config 2's
			// whole frame 1.36 ms into memory of the other socket as 8 x 4, 0.86 ms as 16 x 2 — and so does memory on the far
			// socket (profiles/r03/tile_shapes.txt).  So: rows as wide as the tile allows, up to 64 bytes.  One exception,
			// where the launch has waves to spare: 256-spp frames of headline size take 16 pixels as 8 x 2 instead of 8 as
			// 8 x 1 (half as many store instructions: as fast as the frame left in HBM whichever socket the memory is on).
			const uint64_t pixels = static_cast<uint64_t>(width) * local_rows;
			if (pixels_log2 == 3u && (q.chunks << 4u) <= 256u && (pixels >> 4u) >= 49152u)
				q.pixels_log2 = pixels_log2 = 4u;
			// (16 pixels at 256 spp: 8 x 2 for the 1..4-sphere kernels — basic.toml 2.639 against 2.664 ms as 16 x 1 — and 16 x 1
			// for the 5..8-sphere and the resident ones — dielectric.toml 3.011 against 3.040 as 8 x 2; tile_sweep_drop_in.txt)
			q.tile_w_log2 = (pixels_log2 == 4u && q.chunks == 16u && primitives < 5u) ? 3u : std::min(pixels_log2, 4u);
		}
#ifdef RT_HIP_QUEUE_KNOBS
		// experiment builds only (tools/gpu_tile_shapes.py): tile size and width from the environment, per launch
		if (!big_scene)
		{
			if (const char* knob = std::getenv("RT_HIP_TILE_LOG2"))
				q.pixels_log2 = pixels_log2 = static_cast<uint32_t>(std::atoi(knob));
			q.tile_w_log2 = (pixels_log2 + 1u) / 2u;
			if (const char* knob = std::getenv("RT_HIP_TILE_W_LOG2"))
				q.tile_w_log2 = std::min<uint32_t>(static_cast<uint32_t>(std::atoi(knob)), pixels_log2);
		}
#endif
		const uint32_t tile_w = 1u << q.tile_w_log2, tile_h = (1u << pixels_log2) >> q.tile_w_log2;
		q.tiles_x = (width + tile_w - 1u) / tile_w;
		q.tiles_y = (local_rows + tile_h - 1u) / tile_h;
		return q;
	}

	void rolling_buffer_bytes(const queue_params& queue, uint32_t samples_per_pixel, uint32_t width, uint32_t local_rows, bool big_scene, size_t& item_sums_bytes, size_t& pixel_done_bytes)
	{
		item_sums_bytes = pixel_done_bytes = 0;
		if (!big_scene || (queue.chunks <= 1u && !queue.halves)) // one chunk per pixel: the lane that traced it writes the pixel
			return;
		const size_t pixels = static_cast<size_t>(width) * local_rows;
		item_sums_bytes = queue.halves ? pixels * samples_per_pixel * 16u : pixels * queue.chunks * 16u;
		pixel_done_bytes = pixels * sizeof(uint32_t);
	}

#endif // !RT_HIP_FAST_BUILD

	uint32_t launch_render(const frame_params& frame,
						   const device_scene& scene,
						   const small_scene& small,
						   uint32_t flags,
						   uint32_t* d_rgba8,
						   float* d_rgb_f32,
						   device_counters* d_counters,
						   const rolling_buffers& rolling,
						   uint32_t compute_units,
						   launch_cache& cache,
						   hipStream_t stream)
	{
		if (!frame.width || !frame.local_rows)
			return RT_HIP_KERNEL_NONE;
		const uint32_t variant = choose_kernel(scene, flags, frame.samples_per_pixel, frame.pinhole != 0 || frame.eye_form == 2u, static_cast<uint64_t>(frame.width) * frame.local_rows);
		const bool sm = (flags & RT_HIP_FLAG_SM_MATERIALS) != 0;
		const bool big_scene = variant == RT_HIP_KERNEL_TILED || variant == RT_HIP_KERNEL_STREAMED;
		const queue_params queue = choose_queue(frame.samples_per_pixel, frame.width, frame.local_rows, big_scene, (flags & launch_flag_host_frame) != 0u, half_chunk_choice(flags), scene.n_spheres + scene.n_planes,
												variant == RT_HIP_KERNEL_STREAMED && scene.n_spheres >= sparse_launch_min_spheres);

		// small scenes: one wave per tile, four tiles side by side per workgroup.  Big scenes: a persistent launch — what
		// the device keeps resident, and no more lanes than items
		const uint64_t total_items = static_cast<uint64_t>(frame.width) * frame.local_rows * ((big_scene && queue.halves) ? (frame.samples_per_pixel + queue.item_samples - 1u) / queue.item_samples : queue.chunks);
		const uint64_t items_per_workgroup = big_scene ? static_cast<uint64_t>(block_threads / 64u) * queue.lane_cap : block_threads; // (a sparse launch: lane_cap rays per wave)
		const dim3 grid = big_scene ? dim3(static_cast<uint32_t>(std::min<uint64_t>(0x7FFFFFFFull, (total_items + items_per_workgroup - 1u) / items_per_workgroup))) // capped to the resident count at launch
									: dim3((queue.tiles_x + 3u) / 4u, queue.tiles_y);
		const size_t slot_bytes = big_scene ? 0u : static_cast<size_t>(block_threads / 64u) * tile_slot_bytes(queue);
		if (variant == RT_HIP_KERNEL_SMALL)
		{
			const size_t lds_bytes = small_table_float4s * sizeof(float4) + slot_bytes;
#define RT_HIP_LAUNCH_SMALL(N, P)                                                                                                    \
	(frame.pinhole ? launch_queue<N, P, false>(sm, frame, queue, small, scene, grid, lds_bytes, d_rgba8, d_rgb_f32, d_counters, rolling, compute_units, cache, stream)  \
					 : launch_queue<N, P, true>(sm, frame, queue, small, scene, grid, lds_bytes, d_rgba8, d_rgb_f32, d_counters, rolling, compute_units, cache, stream))
#define RT_HIP_LAUNCH_SMALL_SPHERES(P, N_MAX)                                                                                        \
	switch (scene.n_spheres)                                                                                                         \
	{                                                                                                                                \
		case 1: RT_HIP_LAUNCH_SMALL(1, P); break;                                                                                    \
		case 2: RT_HIP_LAUNCH_SMALL(2, P); break;                                                                                    \
		case 3: RT_HIP_LAUNCH_SMALL(3, P); break;                                                                                    \
		case 4: RT_HIP_LAUNCH_SMALL(4, P); break;                                                                                    \
		case 5: RT_HIP_LAUNCH_SMALL(5, P); break;                                                                                    \
		case 6: if constexpr (N_MAX >= 6) RT_HIP_LAUNCH_SMALL(6, P); break;                                                          \
		case 7: if constexpr (N_MAX >= 7) RT_HIP_LAUNCH_SMALL(7, P); break;                                                          \
		default: if constexpr (N_MAX >= 8) RT_HIP_LAUNCH_SMALL(8, P); break;                                                         \
	}
			switch (scene.n_planes) // (choose_kernel admits n_spheres + n_planes <= 8 only)
			{
				case 0: RT_HIP_LAUNCH_SMALL_SPHERES(0, 8); break;
				case 1: RT_HIP_LAUNCH_SMALL_SPHERES(1, 7); break;
				case 2: RT_HIP_LAUNCH_SMALL_SPHERES(2, 6); break;
				default: RT_HIP_LAUNCH_SMALL_SPHERES(3, 5); break;
			}
#undef RT_HIP_LAUNCH_SMALL_SPHERES
#undef RT_HIP_LAUNCH_SMALL
			return variant;
		}
		if (variant == RT_HIP_KERNEL_RESIDENT)
		{
			const size_t lds_bytes = static_cast<size_t>((scene.n_spheres < resident_scalar_scan_from ? scene.n_spheres : 0u) + scene.n_planes) * sizeof(float4) + slot_bytes;
			if (scene.n_spheres >= resident_scalar_scan_from) // (the scalar-load scan: one build for every camera form)
				launch_queue<0, 1, false>(sm, frame, queue, small, scene, grid, lds_bytes, d_rgba8, d_rgb_f32, d_counters, rolling, compute_units, cache, stream);
			else if (frame.pinhole)
				launch_queue<0, 0, false>(sm, frame, queue, small, scene, grid, lds_bytes, d_rgba8, d_rgb_f32, d_counters, rolling, compute_units, cache, stream);
			else
				launch_queue<0, 0, true>(sm, frame, queue, small, scene, grid, lds_bytes, d_rgba8, d_rgb_f32, d_counters, rolling, compute_units, cache, stream);
			return variant;
		}
		if (variant == RT_HIP_KERNEL_STREAMED)
		{
			// A frame that fills the device — not a thin launch, and at least 4M samples — takes the build without the cooperative scan of
			// sparse waves: only its last waves run sparse, and the registers that scan costs every other trip are worth 9 % (config 5).
			const bool dense = queue.lane_cap == 64u && static_cast<uint64_t>(frame.width) * frame.local_rows * frame.samples_per_pixel >= (1ull << 22);
			if (dense)
				launch_queue<-3>(sm, frame, queue, small, scene, grid, 0, d_rgba8, d_rgb_f32, d_counters, rolling, compute_units, cache, stream);
			else
				launch_queue<-2>(sm, frame, queue, small, scene, grid, 0, d_rgba8, d_rgb_f32, d_counters, rolling, compute_units, cache, stream);
			return variant;
		}
		launch_queue<-1>(sm, frame, queue, small, scene, grid, tile_primitives * sizeof(float4), d_rgba8, d_rgb_f32, d_counters, rolling, compute_units, cache, stream);
		return variant;
	}

#ifndef RT_HIP_FAST_BUILD
	void launch_preview(const frame_params& frame, const device_scene& scene, uint32_t* d_rgba8, float* d_rgb_f32, device_counters* d_counters, hipStream_t stream)
	{
		if (!frame.width || !frame.local_rows)
			return;
		const uint32_t rows_per_block = block_threads / 64u;
		const dim3 grid((frame.width + 63u) / 64u, (frame.local_rows + rows_per_block - 1u) / rows_per_block);
		hipLaunchKernelGGL(preview_frame, grid, dim3(block_threads), 0, stream, frame, scene, d_rgba8, d_rgb_f32, d_counters);
	}

	void launch_assemble(uint32_t width,
						 uint32_t height,
						 uint32_t world,
						 uint32_t stripe_rows,
						 uint32_t padded_local_rows,
						 const uint32_t* d_gathered,
						 uint32_t* d_frame,
						 uint32_t first_rank,
						 bool frame_is_host_memory,
						 hipStream_t stream)
	{
		const dim3 grid((width + block_threads - 1) / block_threads, height);
		if (frame_is_host_memory)
			hipLaunchKernelGGL(assemble_stripes<true>, grid, dim3(block_threads), 0, stream, width, height, world, stripe_rows, padded_local_rows, first_rank, d_gathered, d_frame);
		else
			hipLaunchKernelGGL(assemble_stripes<false>, grid, dim3(block_threads), 0, stream, width, height, world, stripe_rows, padded_local_rows, first_rank, d_gathered, d_frame);
	}

#endif // !RT_HIP_FAST_BUILD
}
