// rt_amd/csrc/delivery.cpp — see delivery.hpp.
#include "delivery.hpp"

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>

#if defined(__linux__)
#include <sched.h>
#endif

#if (defined(__x86_64__) || defined(_M_X64)) && !defined(RT_HIP_CARRIER_PORTABLE) // (the portable path: other hosts, and the ThreadSanitizer build of the test)
#include <emmintrin.h>
#define RT_HIP_CARRIER_SSE2 1
#endif

namespace rt_hip
{
	namespace
	{
		constexpr size_t line_bytes = 64;			  // what is copied, and checked for pending pixels, at a time
		constexpr size_t pixel_band_bytes = 64u << 10; // a band of a frame in flight: 8.5 rows of a 1920-pixel frame
		constexpr size_t copy_band_bytes = 256u << 10;
		constexpr size_t small_frame_bytes = 128u << 10; // frames up to this size are not worth waking anybody for

		// a wait that may be long (the peer it waits for can have been descheduled, or its cgroup throttled): spin briefly, then
		// give the CPU away between looks — the waiter burns the same quota the one it waits for needs
		template <typename Done>
		inline void wait_until(Done done)
		{
			for (unsigned spins = 0; !done(); spins++)
			{
				if (spins < 4096u)
				{
#ifdef RT_HIP_CARRIER_SSE2
					_mm_pause();
#endif
					asm volatile("" ::: "memory");
				}
				else
					std::this_thread::yield();
			}
		}

		inline void relax()
		{
#ifdef RT_HIP_CARRIER_SSE2
			_mm_pause();
#endif
			asm volatile("" ::: "memory"); // whatever is polled is loaded again
		}

		// the 64-byte line at `from` if all of its sixteen pixels are there: copied to `to`, zeros put back; else nothing
		inline bool take_line(unsigned char* from, unsigned char* to)
		{
#ifdef RT_HIP_CARRIER_SSE2
			const __m128i zero = _mm_setzero_si128();
			const __m128i a = _mm_load_si128(reinterpret_cast<const __m128i*>(from));
			const __m128i b = _mm_load_si128(reinterpret_cast<const __m128i*>(from + 16));
			const __m128i c = _mm_load_si128(reinterpret_cast<const __m128i*>(from + 32));
			const __m128i d = _mm_load_si128(reinterpret_cast<const __m128i*>(from + 48));
			const __m128i pending = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi32(a, zero), _mm_cmpeq_epi32(b, zero)), _mm_or_si128(_mm_cmpeq_epi32(c, zero), _mm_cmpeq_epi32(d, zero)));
			if (_mm_movemask_epi8(pending) != 0)
				return false;
			if ((reinterpret_cast<uintptr_t>(to) & 15u) == 0)
			{
				// a 16-byte aligned destination (rt's images are 64-byte aligned, src/image.hpp:11): streaming stores — the
				// line is written whole, nothing of it needs to be read into a cache first
				_mm_stream_si128(reinterpret_cast<__m128i*>(to), a);
				_mm_stream_si128(reinterpret_cast<__m128i*>(to + 16), b);
				_mm_stream_si128(reinterpret_cast<__m128i*>(to + 32), c);
				_mm_stream_si128(reinterpret_cast<__m128i*>(to + 48), d);
			}
			else
			{
				_mm_storeu_si128(reinterpret_cast<__m128i*>(to), a);
				_mm_storeu_si128(reinterpret_cast<__m128i*>(to + 16), b);
				_mm_storeu_si128(reinterpret_cast<__m128i*>(to + 32), c);
				_mm_storeu_si128(reinterpret_cast<__m128i*>(to + 48), d);
			}
			// (the zeros behind the copy are streamed too: a line left MODIFIED in this core's cache would have to be snooped
			// out when the device stores the next frame's pixels into it)
			_mm_stream_si128(reinterpret_cast<__m128i*>(from), zero);
			_mm_stream_si128(reinterpret_cast<__m128i*>(from + 16), zero);
			_mm_stream_si128(reinterpret_cast<__m128i*>(from + 32), zero);
			_mm_stream_si128(reinterpret_cast<__m128i*>(from + 48), zero);
			return true;
#else
			uint32_t words[line_bytes / 4];
			for (size_t k = 0; k < line_bytes / 4; k++)
				words[k] = __atomic_load_n(reinterpret_cast<const uint32_t*>(from) + k, __ATOMIC_RELAXED);
			for (const uint32_t w : words)
				if (!w)
					return false;
			std::memcpy(to, words, line_bytes);
			for (size_t k = 0; k < line_bytes / 4; k++)
				__atomic_store_n(reinterpret_cast<uint32_t*>(from) + k, 0u, __ATOMIC_RELAXED);
			return true;
#endif
		}

		// whole words of [from, from + bytes) as they are now: a word that is still 0 was never stored and leaves the caller's
		// word alone (the reference's caller has pre-cleared its frame, src/main.cpp:318)
		inline void take_words(unsigned char* from, unsigned char* to, size_t bytes)
		{
			for (size_t at = 0; at + 4 <= bytes; at += 4)
			{
				const uint32_t w = __atomic_load_n(reinterpret_cast<const uint32_t*>(from + at), __ATOMIC_RELAXED);
				if (w)
				{
					std::memcpy(to + at, &w, 4);
					__atomic_store_n(reinterpret_cast<uint32_t*>(from + at), 0u, __ATOMIC_RELAXED);
				}
			}
		}

		inline bool all_there(const unsigned char* from, size_t bytes)
		{
			for (size_t at = 0; at + 4 <= bytes; at += 4)
				if (!__atomic_load_n(reinterpret_cast<const uint32_t*>(from + at), __ATOMIC_RELAXED))
					return false;
			return true;
		}
	}

	pixel_carrier::pixel_carrier(unsigned helpers, int numa_node) : numa_node_(numa_node)
	{
		helpers_wanted_ = helpers;
		if (const char* knob = std::getenv("RT_HIP_CARRIER_STAY_HOT_US")) // (experiments: 0 = helpers sleep between any two frames)
		{
			char* end = nullptr;
			const long v = std::strtol(knob, &end, 10);
			if (end != knob && v >= 0 && v <= 1000000)
				stay_hot_ = std::chrono::microseconds(v);
		}
		threads_.reserve(helpers);
		try
		{
			for (unsigned i = 0; i < helpers; i++)
				threads_.emplace_back([this] { helper_main(); });
		}
		catch (const std::exception&)
		{
			// (no more threads to be had: the carrier works with however many came up — with none, the caller's thread
			// carries the whole frame inside finish())
		}
	}

	pixel_carrier::~pixel_carrier()
	{
		if (in_flight_)
			abandon();
		{
			const std::lock_guard<std::mutex> lock(mutex_);
			quit_ = true;
		}
		quit_hint_.store(true, std::memory_order_release);
		wake_.notify_all();
		for (std::thread& t : threads_)
			t.join();
	}

	// The helpers poll and copy memory that lives on one host node (the GPU's): they stay on that node's CPUs — those of
	// them this process may use at all.  Best effort; RT_HIP_CARRIER_PIN=0 leaves the threads where the scheduler puts them.
	void pixel_carrier::stay_near_the_frame()
	{
#if defined(__linux__)
		if (numa_node_ < 0)
			return;
		if (const char* knob = std::getenv("RT_HIP_CARRIER_PIN"))
			if (knob[0] == '0')
				return;
		char path[96];
		std::snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", numa_node_);
		std::FILE* f = std::fopen(path, "r");
		if (!f)
			return;
		char list[4096] = {};
		const size_t got = std::fread(list, 1, sizeof(list) - 1, f);
		std::fclose(f);
		list[got] = 0;
		cpu_set_t allowed, wanted;
		CPU_ZERO(&allowed);
		CPU_ZERO(&wanted);
		if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0)
			return;
		int count = 0;
		for (const char* p = list; *p;) // "0-63,128-191"
		{
			char* end = nullptr;
			const long a = std::strtol(p, &end, 10);
			if (end == p)
				break;
			long b = a;
			if (*end == '-')
			{
				p = end + 1;
				b = std::strtol(p, &end, 10);
			}
			for (long c = a; c <= b && c < CPU_SETSIZE; c++)
				if (c >= 0 && CPU_ISSET(static_cast<int>(c), &allowed))
				{
					CPU_SET(static_cast<int>(c), &wanted);
					count++;
				}
			if (*end != ',')
				break;
			p = end + 1;
		}
		// only if that leaves every helper AND the caller's thread a CPU of their own: helpers spin while a frame is in
		// flight, and seven of them squeezed onto the three CPUs a container happens to have on that node would take
		// turns — with the caller's thread, which the runtime has spinning on the stream — instead of working
		if (count >= static_cast<int>(helpers_wanted_) + 2)
			(void)sched_setaffinity(0, sizeof(wanted), &wanted); // (0 = the calling THREAD)
#endif
	}

	void pixel_carrier::helper_main()
	{
		stay_near_the_frame();
		uint64_t seen = 0;
		for (;;)
		{
			// a renderer is asked for frame after frame: stay awake for a moment after each
			bool fresh = false;
			const auto until = std::chrono::steady_clock::now() + stay_hot_;
			for (unsigned spins = 0;; spins++)
			{
				if (posted_.load(std::memory_order_acquire) != seen)
				{
					fresh = true;
					break;
				}
				if (quit_hint_.load(std::memory_order_acquire))
					return;
				if ((spins & 63u) == 63u && std::chrono::steady_clock::now() >= until)
					break;
				relax();
			}
			if (!fresh)
			{
				sleepers_.fetch_add(1, std::memory_order_seq_cst);
				{
					std::unique_lock<std::mutex> lock(mutex_);
					wake_.wait(lock, [&] { return quit_ || posted_.load(std::memory_order_seq_cst) != seen; });
					if (quit_)
					{
						sleepers_.fetch_sub(1, std::memory_order_seq_cst);
						return;
					}
				}
				sleepers_.fetch_sub(1, std::memory_order_seq_cst);
			}
			seen = posted_.load(std::memory_order_acquire);
			// enter only an OPEN job (its fields are valid then): this thread announces itself first and looks second, the
			// caller's thread closes first and waits for the announced second — one of the two always sees the other
			inside_.fetch_add(1, std::memory_order_seq_cst);
			if (open_.load(std::memory_order_seq_cst))
				work();
			inside_.fetch_sub(1, std::memory_order_seq_cst);
		}
	}

	void pixel_carrier::post()
	{
		open_.store(true, std::memory_order_seq_cst);
		posted_.fetch_add(1, std::memory_order_seq_cst);
		if (sleepers_.load(std::memory_order_seq_cst) != 0)
		{
			{
				const std::lock_guard<std::mutex> lock(mutex_); // (a helper between its predicate and its wait holds this lock)
			}
			wake_.notify_all();
		}
	}

	void pixel_carrier::close()
	{
		open_.store(false, std::memory_order_seq_cst);
		wait_until([&] { return inside_.load(std::memory_order_seq_cst) == 0; });
	}

	// What every thread does with a frame in flight.  A thread keeps a handful of bands OPEN at a time and sweeps over them,
	// taking every line that has become complete and leaving the others for the next sweep — it never waits on one line
	// while others are ready.  (The first version claimed one band and waited on its lines in order: fine while the device
	// finishes the frame in a narrow front, as the headline launch does — every band was delivered before the drain — but a
	// 64-spp launch keeps 8 192 waves x 64 pixels = a quarter of the frame in flight at once, the threads sat on the oldest
	// bands of that front, and 2 MB were still to be copied when the kernel ended: + 0.07 ms on config 2.)
	void pixel_carrier::work()
	{
		if (kind_ == copy_bytes)
		{
			for (;;)
			{
				const size_t claimed = next_band_.fetch_add(1, std::memory_order_relaxed);
				if (claimed >= bands_)
					return;
				const size_t at = claimed * band_bytes_;
				std::memcpy(to_ + at, from_ + at, std::min(band_bytes_, bytes_ - at));
				bands_done_.fetch_add(1, std::memory_order_release);
			}
		}
		constexpr size_t open_bands = 6;
		struct open_band
		{
			size_t begin = 0, lines = 0, left = 0; // byte offset of the band, its lines (the last may be short), lines not yet taken
			uint64_t pending[pixel_band_bytes / line_bytes / 64] = {};
			size_t probe = 0; // the first pending line: the only one looked at until it is complete (then the band is swept)
		} open[open_bands];
		size_t n_open = 0;
		bool more = true, progress = true;
		for (;;)
		{
			// One band is open at a time while it keeps yielding lines; a thread opens another (up to `open_bands`) only when a
			// whole sweep over what it holds yielded nothing — so a narrow front of finishing tiles (the headline launch: 8
			// bands) is followed band by band, a wide one (64 spp: 32 bands) is covered, and no thread sweeps over bands the
			// device has not reached yet.
			while (more && (n_open == 0 || (!progress && n_open < open_bands)))
			{
				progress = true; // (one more band per fruitless sweep)
				const size_t claimed = next_band_.fetch_add(1, std::memory_order_relaxed);
				if (claimed >= bands_)
				{
					more = false;
					break;
				}
				open_band& b = open[n_open++];
				const size_t band = bottom_first_ ? bands_ - 1u - claimed : claimed;
				b.begin = band * band_bytes_;
				const size_t bytes = std::min(bytes_, b.begin + band_bytes_) - b.begin;
				b.lines = b.left = (bytes + line_bytes - 1u) / line_bytes;
				for (size_t w = 0; w < sizeof(b.pending) / sizeof(b.pending[0]); w++)
				{
					const size_t first = w * 64u;
					b.pending[w] = first >= b.lines ? 0ull : (b.lines - first >= 64u ? ~0ull : ((1ull << (b.lines - first)) - 1ull));
				}
			}
			if (n_open == 0)
				return;
			const uint32_t state = state_.load(std::memory_order_acquire);
			if (state == abandoned)
			{
				bands_done_.fetch_add(n_open, std::memory_order_release); // (nothing further is copied; the owner wipes the frame)
				n_open = 0;
				continue; // (claims what is left, to the same end, so that the counts add up)
			}
			const bool final_sweep = state == drained; // (loads below come after the acquire: what the device stored, all of it)
			progress = false;
			for (size_t i = 0; i < n_open;)
			{
				open_band& b = open[i];
				// Poll ONE line per band.  Every line a thread keeps reading sits in its cache, and every store of the device
				// into such a line has to be probed out of that cache first: sweeping all pending lines of the open bands all
				// the time showed in the KERNEL's own time on long frames.  So: the band's first pending line is its probe; only
				// when that one has arrived is the band swept for everything else that has.
				if (!final_sweep)
				{
					const size_t at = b.begin + b.probe * line_bytes;
					const size_t n = std::min(line_bytes, bytes_ - at);
					if (!all_there(from_ + at, n))
					{
						i++;
						continue;
					}
				}
				for (size_t w = 0; w < sizeof(b.pending) / sizeof(b.pending[0]) && b.left; w++)
				{
					uint64_t bits = b.pending[w];
					while (bits)
					{
						const size_t line = w * 64u + static_cast<size_t>(__builtin_ctzll(bits));
						bits &= bits - 1u;
						const size_t at = b.begin + line * line_bytes;
						const size_t n = std::min(line_bytes, bytes_ - at);
						bool taken;
						if (n == line_bytes)
							taken = take_line(from_ + at, to_ + at);
						else if ((taken = all_there(from_ + at, n)))
							take_words(from_ + at, to_ + at, n);
						if (!taken && final_sweep)
						{
							take_words(from_ + at, to_ + at, n); // whatever is there now is all there will be
							taken = true;
						}
						if (taken)
						{
							b.pending[w] &= ~(1ull << (line & 63u));
							b.left--;
							progress = true;
						}
					}
				}
				if (b.left != 0) // the next probe: the first line still pending
					for (size_t w = 0; w < sizeof(b.pending) / sizeof(b.pending[0]); w++)
						if (b.pending[w])
						{
							b.probe = w * 64u + static_cast<size_t>(__builtin_ctzll(b.pending[w]));
							break;
						}
				if (b.left == 0)
				{
#ifdef RT_HIP_CARRIER_SSE2
					_mm_sfence(); // (streaming stores are weakly ordered: the band is in memory before it is reported)
#endif
					bands_done_.fetch_add(1, std::memory_order_release);
					open[i] = open[--n_open];
					progress = true;
				}
				else
					i++;
			}
			if (!progress)
				for (int k = 0; k < 32; k++)
					relax();
		}
	}

	void pixel_carrier::begin(uint32_t* from, uint32_t* to, size_t words, bool bottom_first, bool announce_now)
	{
		if (in_flight_)
			abandon();
		close();
		kind_ = carry_pixels;
		bottom_first_ = bottom_first;
		from_ = reinterpret_cast<unsigned char*>(from);
		to_ = reinterpret_cast<unsigned char*>(to);
		bytes_ = words * sizeof(uint32_t);
		band_bytes_ = pixel_band_bytes;
		bands_ = (bytes_ + band_bytes_ - 1u) / band_bytes_;
		next_band_.store(0, std::memory_order_relaxed);
		bands_done_.store(0, std::memory_order_relaxed);
		state_.store(storing, std::memory_order_relaxed);
		early_bands_ = 0;
		in_flight_ = true;
		announced_ = false;
		if (announce_now)
			announce();
	}

	void pixel_carrier::announce()
	{
		if (!in_flight_ || announced_ || kind_ != carry_pixels)
			return;
		announced_ = true;
		if (!threads_.empty() && bytes_ > small_frame_bytes)
			post();
	}

	void pixel_carrier::finish()
	{
		if (!in_flight_)
			return;
		early_bands_ = bands_done_.load(std::memory_order_acquire);
		state_.store(drained, std::memory_order_release);
		work();
		wait_until([&] { return bands_done_.load(std::memory_order_acquire) == bands_; });
		close();
		in_flight_ = false;
	}

	void pixel_carrier::abandon()
	{
		if (!in_flight_)
			return;
		state_.store(abandoned, std::memory_order_release);
		close();
		in_flight_ = false;
	}

	void pixel_carrier::copy(void* to, const void* from, size_t bytes)
	{
		if (!bytes)
			return;
		// (a frame in flight owns the helpers: the caller's thread copies alone — never "nothing copied, nothing said")
		if (in_flight_ || threads_.empty() || bytes <= copy_band_bytes)
		{
			std::memcpy(to, from, bytes);
			return;
		}
		close();
		kind_ = copy_bytes;
		from_ = static_cast<unsigned char*>(const_cast<void*>(from));
		to_ = static_cast<unsigned char*>(to);
		bytes_ = bytes;
		band_bytes_ = copy_band_bytes;
		bands_ = (bytes_ + band_bytes_ - 1u) / band_bytes_;
		next_band_.store(0, std::memory_order_relaxed);
		bands_done_.store(0, std::memory_order_relaxed);
		post();
		work();
		wait_until([&] { return bands_done_.load(std::memory_order_acquire) == bands_; });
		close();
	}
}
