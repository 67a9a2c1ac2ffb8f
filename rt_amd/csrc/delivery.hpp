// rt_amd/csrc/delivery.hpp — host threads that carry finished pixels from the module's own page-locked frame into the
// caller's frame buffer WHILE the device is still storing the rest (frame.hip owns the memory; this is plain C++17, no HIP,
// so that it is tested on the CPU with a thread playing the device: tests/native/pixel_carrier_test.cpp).
//
// The reference's render() fills a pageable host buffer it was handed (image_view, src/image.cpp:9-13) and the caller
// presents it at once (src/window.cpp:215-216).  The kernels here store every finished pixel — one aligned 32-bit word,
// written through to host memory — into a frame the MODULE owns; a packed pixel is never 0 (alpha is always 255,
// src/colour.hpp:63-65,101-106), and the frame is all zero when a frame starts, so "this word is not 0" IS "this pixel is
// finished": no flag, no counter and no change to any kernel.  The threads take the frame in bands of 64 KB, from the bottom
// of the image up (the order the launch hands its tiles out), each keeping a few bands open and sweeping over them: every
// 64-byte line whose sixteen words are all there is copied and zeros are put back behind it — the invariant the next frame
// starts from — and a line that is not complete yet is left for the next sweep.  When the caller's thread has seen the
// stream drain it says so (finish): from then on whatever is in a line is final, and what is left — the tiles that were
// in flight at the very end — is carried over by all threads and the caller's own.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <thread>
#include <vector>

namespace rt_hip
{
	class pixel_carrier
	{
	  public:
		// `helpers`: threads besides the caller's own (0 = the caller's thread carries everything inside finish()).
		// `numa_node` >= 0: the helpers run on that host node's CPUs (where the frame they poll lives: a thread that polls
		// lines of memory homed on the other socket drags every store of the device into them across the socket link).
		explicit pixel_carrier(unsigned helpers, int numa_node = -1);
		~pixel_carrier();
		pixel_carrier(const pixel_carrier&) = delete;
		pixel_carrier& operator=(const pixel_carrier&) = delete;

		// A frame of `words` pixels begins: `from` (64-byte aligned, module-owned, ALL ZERO) is about to be stored into by
		// the device; `to` is the caller's buffer.  Returns at once.
		// `bottom_first`: the order the bands are looked at — the render launches hand their tiles out bottom row first; a frame
		// whose bulk arrives through the multi-GPU assemble kernel fills from the top.
		// `announce_now` = false: the helpers are told by announce() — the owner calls it once the launch has been issued,
		// so that waking sleeping helpers (a futex call and a scheduler's latency each) delays no kernel.
		void begin(uint32_t* from, uint32_t* to, size_t words, bool bottom_first = true, bool announce_now = true);
		// The helpers start looking at the frame begun with announce_now = false.  (Never called: finish() carries everything.)
		void announce();
		// Everything the device stored is visible to this thread now (the stream has drained): carry over what is left.
		// Returns when every word is in `to` and `from` is all zero again.
		void finish();
		// Instead of finish(): nothing further is copied.  Returns when no thread touches `from` or `to` any more; `from`
		// is NOT clean afterwards (the owner wipes it).
		void abandon();
		// `bytes` from module-owned memory to the caller's, split over the helpers and this thread (no frame in flight).
		void copy(void* to, const void* from, size_t bytes);

		unsigned helpers() const { return static_cast<unsigned>(threads_.size()); }
		// (tests, RT_HIP_DEBUG_FRAME) bands of the last frame that were complete before finish() was called
		size_t early_bands() const { return early_bands_; }
		size_t bands() const { return bands_; } // of the frame begun last

	  private:
		enum : uint32_t { storing = 0, drained = 1, abandoned = 2 };
		enum : uint32_t { carry_pixels = 0, copy_bytes = 1 };

		void helper_main();
		void stay_near_the_frame(); // (the first act of a helper: its CPU affinity)
		void post(); // make the job visible to the helpers and wake the sleeping ones
		void work(); // what every thread does with the job in flight
		void close(); // no thread may enter the job any more; waits for those inside

		std::vector<std::thread> threads_;
		int numa_node_ = -1;
		unsigned helpers_wanted_ = 0;
		std::mutex mutex_;
		std::condition_variable wake_;
		bool quit_ = false;					 // (under mutex_)
		std::atomic<bool> quit_hint_{ false }; // the same for helpers that are awake
		std::atomic<uint64_t> posted_{ 0 };	 // jobs posted so far
		std::atomic<uint32_t> sleepers_{ 0 };
		std::atomic<bool> open_{ false };	 // the job's fields are valid and threads may enter
		std::atomic<uint32_t> inside_{ 0 };	 // threads inside work()

		// the job in flight: written by the caller's thread while the job is closed, read by threads that entered an open job
		uint32_t kind_ = carry_pixels;
		bool bottom_first_ = true;
		unsigned char* from_ = nullptr;
		unsigned char* to_ = nullptr;
		size_t bytes_ = 0, bands_ = 0, band_bytes_ = 0;
		std::atomic<size_t> next_band_{ 0 }, bands_done_{ 0 };
		std::atomic<uint32_t> state_{ storing };
		size_t early_bands_ = 0;
		bool in_flight_ = false;
		bool announced_ = false;
		std::chrono::microseconds stay_hot_{ 150 }; // how long a helper keeps looking for the next frame before it sleeps
	};
}
