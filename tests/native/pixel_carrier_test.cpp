// tests/native/pixel_carrier_test.cpp — the pixel carrier of rt_amd/csrc/delivery.cpp on the CPU, with a thread playing the
// device: it stores the pixels of a frame into the "module-owned" staging frame tile by tile, in a scrambled bottom-up order
// and with pauses, as aligned 32-bit words (what the kernels' write-through stores are to the host), while the carrier's
// threads copy finished lines into the "caller's" buffer.  Checked per frame: the caller's buffer holds exactly the frame,
// the staging frame is all zero again, nothing outside the buffer was touched; abandoned frames copy nothing further.
// Built plain and with -fsanitize=thread (tests/test_pixel_carrier.py).
#include "../../rt_amd/csrc/delivery.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

using rt_hip::pixel_carrier;

static uint32_t pixel_of(uint64_t frame, size_t i)
{
	uint64_t z = (frame << 40) ^ i ^ 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return static_cast<uint32_t>(z >> 32) | 0xFFu; // alpha 255: never 0
}

struct frame_case
{
	size_t width, height;
	unsigned tile_w, tile_h;
};

// the "device": stores tile after tile, roughly bottom row first, each pixel one relaxed atomic 32-bit store
static void device_stores(uint32_t* staging, const frame_case& f, uint64_t frame, unsigned seed, size_t stop_after_tiles, bool bottom_first)
{
	const size_t tiles_x = (f.width + f.tile_w - 1) / f.tile_w, tiles_y = (f.height + f.tile_h - 1) / f.tile_h;
	std::vector<size_t> order(tiles_x * tiles_y);
	for (size_t i = 0; i < order.size(); i++)
		order[i] = i;
	std::mt19937 rng(seed);
	// bottom-up with a window of disorder, like a launch whose waves finish out of order
	for (size_t i = 0; i < order.size(); i++)
		std::swap(order[i], order[std::min(order.size() - 1, i + rng() % 97)]);
	size_t done = 0;
	for (const size_t t : order)
	{
		if (done++ == stop_after_tiles)
			return;
		const size_t ty = bottom_first ? tiles_y - 1 - t / tiles_x : t / tiles_x, tx = t % tiles_x; // (top first: the multi-GPU assemble kernel)
		for (size_t y = ty * f.tile_h; y < std::min(f.height, (ty + 1) * f.tile_h); y++)
			for (size_t x = tx * f.tile_w; x < std::min(f.width, (tx + 1) * f.tile_w); x++)
				__atomic_store_n(&staging[y * f.width + x], pixel_of(frame, y * f.width + x), __ATOMIC_RELAXED);
		if ((rng() & 255u) == 0)
			std::this_thread::sleep_for(std::chrono::microseconds(rng() % 40));
	}
}

int main(int argc, char** argv)
{
	const unsigned helpers = argc > 1 ? static_cast<unsigned>(std::atoi(argv[1])) : 3u;
	const unsigned rounds = argc > 2 ? static_cast<unsigned>(std::atoi(argv[2])) : 3u;
	const frame_case cases[] = { { 1, 1, 1, 1 }, { 7, 3, 4, 2 }, { 333, 187, 8, 2 }, { 640, 360, 16, 1 }, { 1000, 601, 4, 4 }, { 1920, 270, 16, 1 }, { 33, 5000, 8, 2 } };
	pixel_carrier carrier(helpers);
	size_t frames = 0, early = 0, bands = 0;
	for (unsigned round = 0; round < rounds; round++)
		for (const frame_case& f : cases)
		{
			const size_t words = f.width * f.height;
			const size_t guard = 64;
			uint32_t* const staging = static_cast<uint32_t*>(std::aligned_alloc(64, ((words * 4 + 63) / 64) * 64));
			std::memset(staging, 0, ((words * 4 + 63) / 64) * 64);
			std::vector<uint32_t> caller(words + 2 * guard, 0xA5A5A5A5u);
			uint32_t* const to = caller.data() + guard + (round % 3); // not 16-byte aligned in two rounds of three
			for (uint64_t frame = 1; frame <= 3; frame++)
			{
				const bool abandon = frame == 2 && (round & 1);
				for (size_t i = 0; i < words; i++)
					to[i] = 0x000000FFu; // the caller's pre-cleared black
				const bool bottom_first = ((round + frame) & 1) == 0;
				// the helpers are told at once, behind the "launch" (as frame_delivery does), or never (finish carries everything)
				const unsigned told = static_cast<unsigned>((round + frame + f.tile_w) % 3);
				carrier.begin(staging, to, words, bottom_first, told == 0);
				std::thread device(device_stores, staging, std::cref(f), frame + 10 * round, static_cast<unsigned>(frame * 7919 + round), abandon ? (words > 64 ? 17 : 0) : ~size_t(0), bottom_first);
				if (told == 1)
					carrier.announce();
				device.join(); // = the stream has drained
				if (abandon)
				{
					carrier.abandon();
					for (size_t i = 0; i < words; i++)
						if (to[i] != 0x000000FFu && to[i] != pixel_of(frame + 10 * round, i))
							return std::printf("FAIL: abandoned frame wrote a foreign word at %zu\n", i), 1;
					std::memset(staging, 0, words * 4); // what frame_delivery::begin does with a dirty frame
					continue;
				}
				carrier.finish();
				frames++, early += carrier.early_bands(), bands += (words * 4 + 65535) / 65536;
				for (size_t i = 0; i < words; i++)
					if (to[i] != pixel_of(frame + 10 * round, i))
						return std::printf("FAIL: %zux%zu frame %llu word %zu is %08x, wanted %08x\n", f.width, f.height, static_cast<unsigned long long>(frame), i, to[i], pixel_of(frame + 10 * round, i)), 1;
				for (size_t i = 0; i < words; i++)
					if (staging[i])
						return std::printf("FAIL: staging word %zu of a %zux%zu frame is not zero after delivery\n", i, f.width, f.height), 1;
			}
			for (size_t i = 0; i < caller.size(); i++)
				if ((i < guard + (round % 3) || i >= guard + (round % 3) + words) && caller[i] != 0xA5A5A5A5u)
					return std::printf("FAIL: word %zu outside the caller's buffer was touched\n", i), 1;
			// the plain copy (the float mean's way)
			std::vector<unsigned char> src(words * 12 + 5), dst(words * 12 + 5, 0);
			for (size_t i = 0; i < src.size(); i++)
				src[i] = static_cast<unsigned char>(i * 131 + round);
			carrier.copy(dst.data(), src.data(), src.size());
			if (src != dst)
				return std::printf("FAIL: copy of %zu bytes differs\n", src.size()), 1;
			std::free(staging);
		}
	std::printf("OK: %zu frames with %u helpers; %zu of %zu bands were complete before the drain\n", frames, helpers, early, bands);
	return 0;
}
