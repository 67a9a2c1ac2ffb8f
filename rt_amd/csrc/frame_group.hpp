// rt_amd/csrc/frame_group.hpp — the ranks of ONE renderer that live in different processes and all map the SAME back
// buffer (rt_hip_join_frame_group, include/rt_hip.h).
//
// With one process per GPU the frame's pixels can reach rank 0's caller in two ways: over xGMI to rank 0's GPU and from
// there over ONE PCIe link to the host (the gathered form: ncclGather + assemble, serialised behind the tracing), or —
// when the caller's back buffer is a shared mapping that every rank process maps and page-locks — straight from every
// GPU into their image rows of that buffer, each over its own PCIe link, while the frame is still being traced.  The
// second form has no data-path collective at all.  What it does need is what this file holds: a control block in POSIX
// shared memory through which the ranks agree on the call (same frame size, seed, flags, scene), check once per buffer
// that their mappings really are one memory, and tell each other when their stripes are in place — two counters bumped
// once per rank per frame.  Nothing in the reference corresponds (it is one process: src/renderers/mg_ray_tracer.cpp:203-204
// is a thread pool's for_range + wait()).
//
// Pure POSIX + C++17: no HIP in here, so that the protocol is tested on the CPU with forked processes
// (tests/native/frame_group_ranks.cpp).
#pragma once

#include <fcntl.h>
#include <sched.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>

namespace rt_hip
{
	constexpr uint32_t frame_group_magic = 0x47465452u; // "RTFG"
	constexpr uint32_t frame_group_max_ranks = 64u;

	// what every rank must have been called with (rank 0's values are the reference ones)
	struct frame_group_call
	{
		uint32_t width, height, flags, samples_per_pixel;
		uint64_t seed, scene_fingerprint;
	};

	inline bool same_call(const frame_group_call& a, const frame_group_call& b)
	{
		return a.width == b.width && a.height == b.height && a.flags == b.flags && a.samples_per_pixel == b.samples_per_pixel && a.seed == b.seed && a.scene_fingerprint == b.scene_fingerprint;
	}

	// one rank's line of the block: written by that rank alone, read by the others between two barriers
	struct alignas(64) frame_group_rank
	{
		int32_t device;
		int32_t pid;
		int32_t numa_node;	 // host NUMA node that rank's GPU hangs off (-1: unknown)
		uint32_t new_buffer; // this frame's pixel buffer was page-locked in this call (not seen before)
		uint32_t kernel_variant;
		uint64_t primary_samples, segments, sphere_tests, plane_tests;
		float render_ms, upload_ms;
	};

	struct frame_group_block
	{
		std::atomic<uint32_t> magic; // stored last by the creator
		uint32_t world;
		int32_t creator_pid; // rank 0's process: a block whose creator is gone (or whose join is over) is a leftover, not this group's
		// ... as far as a reader can tell: a pid means something only inside its PID namespace, and may be in use again by another
		// process.  The creator's namespace (inode of /proc/self/ns/pid) and its start time (field 22 of /proc/<pid>/stat,
		// clock ticks since boot) say whether "no such process" is to be believed and whether "such a process" is still THAT one.
		uint64_t creator_pidns;
		uint64_t creator_start;
		alignas(64) std::atomic<uint64_t> joined;
		alignas(64) std::atomic<uint64_t> entered;	 // + 1 per rank per frame: everybody is inside rt_hip_render
		alignas(64) std::atomic<uint64_t> nonce_set; // + 1 per rank per buffer check: rank 0's mark is in the buffer
		alignas(64) std::atomic<uint64_t> nonce_seen; // + 1 per rank per buffer check: everybody has looked
		alignas(64) std::atomic<uint64_t> finished;	 // + 1 per rank per frame: this rank's stripes are in the buffer
		alignas(64) std::atomic<uint32_t> broken;	 // 0, or 1 + the rank that broke the group; sticky
		char reason[240];
		alignas(64) frame_group_call call; // rank 0's, written before it arrives at `entered`
		uint32_t nonce;
		frame_group_rank ranks[frame_group_max_ranks];
	};
	static_assert(std::atomic<uint64_t>::is_always_lock_free && std::atomic<uint32_t>::is_always_lock_free, "the block's counters are shared between processes");

	class frame_group
	{
	  public:
		enum class outcome
		{
			ok,
			broken,
			timed_out,
			failed
		};

		frame_group_block* block = nullptr;
		uint32_t rank = 0, world = 1;
		uint32_t deadline_ms = 120000u; // how long a rank waits for the others inside one frame
		uint64_t frames = 0, checks = 0; // barriers passed so far (every rank counts the same)
		std::string error;				 // why the last call of this object failed

		frame_group() = default;
		frame_group(const frame_group&) = delete;
		frame_group& operator=(const frame_group&) = delete;
		~frame_group()
		{
			if (block)
			{
				break_group("rank %u left the group", rank); // the others stop waiting for it at once
				(void)munmap(block, sizeof(frame_group_block));
			}
			unlink_name();
		}

		// Collective over the `world` processes that name the same object: rank 0 creates it, the others wait for it to
		// appear; returns when all have mapped it (then rank 0 removes the name: nothing is left behind in /dev/shm).
		outcome join(const char* name, uint32_t my_rank, uint32_t my_world, uint32_t timeout_ms)
		{
			if (!name || name[0] != '/' || std::strchr(name + 1, '/') || std::strlen(name) > 200 || !name[1])
				return fail("the group's name must look like \"/something\" (shm_open)");
			if (!my_world || my_world > frame_group_max_ranks || my_rank >= my_world)
				return fail("invalid rank %u of %u (at most %u ranks)", my_rank, my_world, frame_group_max_ranks);
			rank = my_rank;
			world = my_world;
			const auto t0 = std::chrono::steady_clock::now();
			const auto late = [&] { return std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms); };
			int fd = -1;
			if (rank == 0)
			{
				fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
				if (fd < 0 && errno == EEXIST && is_leftover(name))
				{
					// a block of that name from a run that crashed (its creator is gone) or is over (its join completed and the
					// name should have been removed): not this group's — replaced
					(void)shm_unlink(name);
					fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
				}
				if (fd < 0)
					return fail("shm_open(%s, O_CREAT | O_EXCL) failed: %s", name, std::strerror(errno));
				name_ = name;
				if (ftruncate(fd, static_cast<off_t>(sizeof(frame_group_block))) != 0)
				{
					const int e = errno;
					(void)close(fd);
					return fail("ftruncate(%s) failed: %s", name, std::strerror(e));
				}
			}
			else
			{
				for (;;)
				{
					fd = shm_open(name, O_RDWR, 0600);
					if (fd >= 0)
					{
						struct stat st;
						// created AND sized — and not a leftover of an earlier run under the same name (rank 0 replaces those: wait
						// for the replacement instead of joining a block whose counters are already at their targets)
						if (fstat(fd, &st) == 0 && static_cast<size_t>(st.st_size) >= sizeof(frame_group_block) && !is_leftover_fd(fd))
							break;
						(void)close(fd);
						fd = -1;
					}
					else if (errno != ENOENT)
						return fail("shm_open(%s) failed: %s", name, std::strerror(errno));
					if (late())
						return fail_as(outcome::timed_out, "rank %u of %u waited %u ms for rank 0 to create %s", rank, world, timeout_ms, name);
					nap(200);
				}
			}
			void* const mapping = mmap(nullptr, sizeof(frame_group_block), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
			const int map_errno = errno;
			(void)close(fd);
			if (mapping == MAP_FAILED)
				return fail("mmap of %s failed: %s", name, std::strerror(map_errno));
			block = static_cast<frame_group_block*>(mapping); // (fresh shared memory reads as zeros: every counter starts at 0)
			if (rank == 0)
			{
				block->world = world;
				block->creator_pid = static_cast<int32_t>(getpid());
				block->creator_pidns = pid_namespace();
				block->creator_start = process_start_time(static_cast<int32_t>(getpid()));
				block->magic.store(frame_group_magic, std::memory_order_release);
			}
			else
			{
				while (block->magic.load(std::memory_order_acquire) != frame_group_magic)
				{
					if (late())
						return fail_as(outcome::timed_out, "rank %u of %u waited %u ms for rank 0 to initialise %s", rank, world, timeout_ms, name);
					nap(50);
				}
				if (block->world != world)
					return fail("rank %u was told a world of %u, rank 0 created %s for %u", rank, world, name, block->world);
			}
			block->ranks[rank].pid = static_cast<int32_t>(getpid());
			block->joined.fetch_add(1, std::memory_order_acq_rel);
			const uint32_t frame_deadline = deadline_ms;
			deadline_ms = timeout_ms;
			const outcome all_here = wait_for(block->joined, world, "to join", true);
			deadline_ms = frame_deadline;
			unlink_name(); // (on every outcome: a group that did not form leaves nothing behind either)
			if (all_here == outcome::ok)
			{
				// Can this process see the other ranks' process ids at all (one pid namespace)?  Everybody has just arrived, so
				// everybody is alive: if a rank looks dead NOW, pids mean nothing here and the waits below rely on the deadline alone.
				for (uint32_t r = 0; r < world; r++)
					if (r != rank && !process_exists(block->ranks[r].pid))
						pids_visible_ = false;
			}
			return all_here;
		}

		bool is_broken() const
		{
			return block && block->broken.load(std::memory_order_acquire) != 0;
		}

		// Sticky: every wait of every rank returns `broken` from now on; the group is destroyed and made anew to go on.
		void break_group(const char* format, ...)
		{
			if (!block)
				return;
			uint32_t expected = 0;
			if (block->broken.compare_exchange_strong(expected, 0x80000000u, std::memory_order_acq_rel))
			{
				va_list args;
				va_start(args, format);
				std::vsnprintf(block->reason, sizeof(block->reason), format, args);
				va_end(args);
				block->broken.store(1u + rank, std::memory_order_release); // (readers wait for the text: see why_broken)
			}
		}

		std::string why_broken() const
		{
			if (!block)
				return "no group";
			for (int i = 0; i < 100000 && block->broken.load(std::memory_order_acquire) == 0x80000000u; i++)
				relax(); // the rank that broke it is still writing the reason
			char text[sizeof(block->reason) + 1];
			std::memcpy(text, block->reason, sizeof(block->reason));
			text[sizeof(block->reason)] = '\0';
			return text;
		}

		// the frame's three meeting points (every rank calls each of them once per frame / per buffer check, in this order)
		outcome enter_frame()
		{
			if (is_broken()) // (before arriving: a broken group starts no further frame)
			{
				error = "the frame group is broken: " + why_broken();
				return outcome::broken;
			}
			frames++;
			block->entered.fetch_add(1, std::memory_order_acq_rel);
			return wait_for(block->entered, frames * world, "to enter the frame", false);
		}
		bool any_new_buffer() const // (between enter_frame and finish_frame: every rank reads the same answer)
		{
			bool any = false;
			for (uint32_t r = 0; r < world; r++)
				any = any || block->ranks[r].new_buffer != 0;
			return any;
		}
		// Rank 0 has put `block->nonce` into the first word of ITS mapping of the pixel buffer; every other rank looks at the
		// first word of its own.  `mine` is that word's address.  Breaks the group if the mappings are not one memory.
		outcome check_buffer(volatile uint32_t* mine)
		{
			checks++;
			if (rank == 0)
			{
				block->nonce = 0x9E3779B9u * static_cast<uint32_t>(checks) ^ static_cast<uint32_t>(getpid());
				*mine = block->nonce;
			}
			block->nonce_set.fetch_add(1, std::memory_order_acq_rel);
			if (const outcome o = wait_for(block->nonce_set, checks * world, "to mark the pixel buffer", false); o != outcome::ok)
				return o;
			if (rank != 0 && *mine != block->nonce)
				break_group("rank %u's pixel buffer is not a mapping of the memory rank 0 renders into (rt_hip_join_frame_group: every rank passes its own mapping of ONE shared buffer)", rank);
			block->nonce_seen.fetch_add(1, std::memory_order_acq_rel);
			return wait_for(block->nonce_seen, checks * world, "to look at the pixel buffer", false); // (everybody has looked: a verdict of "broken" is in by now)
		}
		// A frame whose every rank has arrived here IS complete, whatever happens to the group afterwards (a rank that
		// leaves right behind its last frame must not turn that frame into a failure).
		outcome finish_frame()
		{
			block->finished.fetch_add(1, std::memory_order_acq_rel);
			return wait_for(block->finished, frames * world, "to finish their stripes", true);
		}

	  private:
		std::string name_; // rank 0: the object's name while it still exists
		bool pids_visible_ = true; // the ranks' process ids can be probed from this process (checked when the group forms)

		// alive, as far as this process can tell: the pid exists and is not a zombie (a rank that died is a zombie until its
		// launcher reaps it, and kill(pid, 0) still succeeds on those)
		static bool process_exists(int32_t pid)
		{
			if (pid <= 0 || !(kill(static_cast<pid_t>(pid), 0) == 0 || errno == EPERM))
				return false;
			char path[64], text[512];
			std::snprintf(path, sizeof(path), "/proc/%d/stat", static_cast<int>(pid));
			const int fd = open(path, O_RDONLY | O_CLOEXEC);
			if (fd < 0)
				return true; // (no /proc to ask: the pid exists, that is all that can be said)
			const ssize_t n = read(fd, text, sizeof(text) - 1);
			(void)close(fd);
			if (n <= 0)
				return true;
			text[n] = '\0';
			const char* const after_name = std::strrchr(text, ')'); // "pid (comm) S ...": comm may contain anything but ends at the LAST ')'
			return !(after_name && after_name[1] == ' ' && (after_name[2] == 'Z' || after_name[2] == 'X'));
		}
		// 0 = cannot be told
		static uint64_t pid_namespace()
		{
			struct stat st;
			return stat("/proc/self/ns/pid", &st) == 0 ? static_cast<uint64_t>(st.st_ino) : 0u;
		}
		static uint64_t process_start_time(int32_t pid)
		{
			char path[64], text[1024];
			std::snprintf(path, sizeof(path), "/proc/%d/stat", static_cast<int>(pid));
			const int fd = open(path, O_RDONLY | O_CLOEXEC);
			if (fd < 0)
				return 0u;
			const ssize_t n = read(fd, text, sizeof(text) - 1);
			(void)close(fd);
			if (n <= 0)
				return 0u;
			text[n] = '\0';
			const char* at = std::strrchr(text, ')'); // fields 3.. follow the LAST ')' (comm may contain anything)
			for (int field = 2; at && field < 22; field++)
				at = std::strchr(at + 1, ' ');
			return at ? std::strtoull(at + 1, nullptr, 10) : 0u;
		}
		// the creator of a found block is gone: only a reader in the creator's PID namespace can say so (from another namespace a
		// live creator looks like "no such process": such a reader relies on joined >= world, `broken` and its own deadline), and a
		// process that exists under that pid but started at another time is not the creator
		static bool creator_is_gone(const frame_group_block* found)
		{
			const uint64_t mine = pid_namespace();
			if (!mine || !found->creator_pidns || mine != found->creator_pidns)
				return false;
			if (!process_exists(found->creator_pid))
				return true;
			const uint64_t started = process_start_time(found->creator_pid);
			return started && found->creator_start && started != found->creator_start;
		}
		// A block found under the group's name that cannot be THIS group's: initialised, and either its creator is gone or
		// everybody of ITS world had joined (a live group removes its name at that moment) or it was broken.
		static bool is_leftover_fd(int fd)
		{
			void* const mapping = mmap(nullptr, sizeof(frame_group_block), PROT_READ, MAP_SHARED, fd, 0);
			if (mapping == MAP_FAILED)
				return false;
			const frame_group_block* const found = static_cast<const frame_group_block*>(mapping);
			const bool initialised = found->magic.load(std::memory_order_acquire) == frame_group_magic;
			const bool leftover = initialised && (creator_is_gone(found) || found->joined.load(std::memory_order_acquire) >= found->world || found->broken.load(std::memory_order_acquire) != 0);
			(void)munmap(mapping, sizeof(frame_group_block));
			return leftover;
		}
		static bool is_leftover(const char* name)
		{
			const int fd = shm_open(name, O_RDWR, 0600);
			if (fd < 0)
				return false;
			struct stat st;
			const bool leftover = fstat(fd, &st) == 0 && static_cast<size_t>(st.st_size) >= sizeof(frame_group_block) && is_leftover_fd(fd);
			(void)close(fd);
			return leftover;
		}

		void unlink_name()
		{
			if (!name_.empty())
				(void)shm_unlink(name_.c_str());
			name_.clear();
		}

		static void relax()
		{
#if defined(__x86_64__) || defined(__i386__)
			__builtin_ia32_pause();
#else
			std::atomic_signal_fence(std::memory_order_seq_cst);
#endif
		}
		static void nap(long microseconds)
		{
			timespec ts{ 0, microseconds * 1000L };
			(void)nanosleep(&ts, nullptr);
		}

		outcome fail(const char* format, ...)
		{
			char text[400];
			va_list args;
			va_start(args, format);
			std::vsnprintf(text, sizeof(text), format, args);
			va_end(args);
			error = text;
			return outcome::failed;
		}
		outcome fail_as(outcome o, const char* format, ...)
		{
			char text[400];
			va_list args;
			va_start(args, format);
			std::vsnprintf(text, sizeof(text), format, args);
			va_end(args);
			error = text;
			return o;
		}

		// Spin (the other ranks are normally a microsecond away), then yield, then sleep.  `arrival_wins`: a counter that
		// has reached its target is a success even if the group has been broken since; otherwise a broken group is
		// reported even when everybody arrived (the verdict of a check travels that way).
		outcome wait_for(const std::atomic<uint64_t>& counter, uint64_t target, const char* what, bool arrival_wins)
		{
			const auto t0 = std::chrono::steady_clock::now();
			const auto broken = [&]
			{
				error = "the frame group is broken: " + why_broken();
				return outcome::broken;
			};
			for (uint64_t spins = 0;; spins++)
			{
				if (counter.load(std::memory_order_acquire) >= target)
				{
					if (!arrival_wins && block->broken.load(std::memory_order_acquire))
						return broken();
					return outcome::ok;
				}
				if (block->broken.load(std::memory_order_acquire))
					return broken();
				if (spins < 20000)
					relax();
				else
				{
					if ((spins & 15u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(deadline_ms))
					{
						const uint64_t have = counter.load(std::memory_order_acquire);
						break_group("rank %u waited %u ms for the other ranks %s (%llu of %u there)", rank, deadline_ms, what, static_cast<unsigned long long>(have - (target - world)), world);
						error = "the frame group is broken: " + why_broken();
						return outcome::timed_out;
					}
					if (spins < 200000)
						(void)sched_yield();
					else
					{
						// the slow path: somebody is late by milliseconds.  A rank whose PROCESS is gone (an HSA abort, a kill) will
						// never arrive: say so now, not at the deadline
						if (pids_visible_ && (spins & 63u) == 0)
							for (uint32_t r = 0; r < world; r++)
								if (r != rank && block->ranks[r].pid > 0 && !process_exists(block->ranks[r].pid))
								{
									break_group("rank %u's process (pid %d) is gone; rank %u was waiting for the other ranks %s", r, block->ranks[r].pid, rank, what);
									return broken();
								}
						nap(50);
					}
				}
			}
		}
	};
}
