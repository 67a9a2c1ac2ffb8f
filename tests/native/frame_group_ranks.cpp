// tests/native/frame_group_ranks.cpp — the frame group's protocol (rt_amd/csrc/frame_group.hpp) with forked processes
// on the CPU: no GPU, no HIP.  Each "rank" is a child process; the "pixel buffer" is a second shared-memory object that
// every rank maps for itself (or, in the failing scenario, does not).
//
//   frame_group_ranks <scenario> <world>
// scenarios:
//   frames    200 frames: every rank fills its stripes of the buffer after enter_frame, rank 0 checks the whole buffer
//             after finish_frame — a rank that ran ahead or a barrier that let somebody through early shows as a wrong word
//   private   rank 1 passes a private buffer: the buffer check must break the group, on every rank
//   mismatch  rank 1 is called with another seed: every rank must see the group broken
//   leaves    the last rank exits after 3 frames: the others must stop waiting at once (not at the deadline)
//   absent    the last rank never joins: join must time out on the others and leave no name behind
//   silent    the last rank stops calling after 3 frames without leaving: the deadline must end the wait
//   killed    the last rank's process is killed (SIGKILL: no destructor runs) after 3 frames: the others must be told that
//             its process is gone within a fraction of a second, long before the 20 s deadline
//   stale     a block of the same name was left behind by an earlier run (initialised, every counter at its target, its
//             creator gone): rank 0 must replace it and no rank may join the leftover — then 200 frames as in `frames`
// Prints "OK: ..." and exits 0 when every rank behaved as the scenario demands.
#include "../../rt_amd/csrc/frame_group.hpp"

#include <signal.h>
#include <sys/wait.h>

#include <cstdlib>
#include <string>
#include <vector>

using rt_hip::frame_group;
using outcome = rt_hip::frame_group::outcome;

namespace
{
	constexpr uint32_t words = 64 * 1024; // the "frame"
	constexpr uint32_t stripe = 512;	  // words per stripe, round-robin over the ranks

	uint32_t* map_buffer(const char* name, bool create)
	{
		const int fd = shm_open(name, create ? (O_CREAT | O_RDWR) : O_RDWR, 0600);
		if (fd < 0)
			return nullptr;
		if (create && ftruncate(fd, words * sizeof(uint32_t)) != 0)
			return nullptr;
		void* p = mmap(nullptr, words * sizeof(uint32_t), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		close(fd);
		return p == MAP_FAILED ? nullptr : static_cast<uint32_t*>(p);
	}

	uint32_t pixel(uint32_t frame, uint32_t index)
	{
		return frame * 0x01000193u ^ index * 0x9E3779B9u;
	}

	// one rank's life; returns the process's exit code (0 = behaved as the scenario demands)
	int run_rank(const std::string& scenario, uint32_t rank, uint32_t world, const std::string& group_name, const std::string& buffer_name)
	{
		const uint32_t last = world - 1;
		if (scenario == "absent" && rank == last)
			return 0;
		frame_group group;
		group.deadline_ms = scenario == "silent" ? 400u : 20000u;
		const auto t_join = std::chrono::steady_clock::now();
		const outcome joined = group.join(group_name.c_str(), rank, world, scenario == "absent" ? 500u : 20000u);
		if (scenario == "absent")
		{
			const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_join).count();
			// (the first rank whose deadline passes breaks the group; the others see that a moment before their own passes)
			const bool gave_up = joined == outcome::timed_out || (joined == outcome::broken && group.error.find("waited 500 ms") != std::string::npos);
			if (!gave_up || waited < 0.4 || waited > 5.0)
			{
				std::fprintf(stderr, "rank %u: join returned %d after %.2f s: %s\n", rank, static_cast<int>(joined), waited, group.error.c_str());
				return 1;
			}
			return 0;
		}
		if (joined != outcome::ok)
		{
			std::fprintf(stderr, "rank %u: join failed: %s\n", rank, group.error.c_str());
			return 1;
		}
		std::vector<uint32_t> private_words(words, 0u);
		uint32_t* buffer = (scenario == "private" && rank == 1) ? private_words.data() : map_buffer(buffer_name.c_str(), false);
		if (!buffer)
			return 2;
		const uint32_t n_frames = 200;
		for (uint32_t f = 1; f <= n_frames; f++)
		{
			if (scenario == "killed" && rank == last && f == 4)
				kill(getpid(), SIGKILL);
			if ((scenario == "leaves" || scenario == "silent") && rank == last && f == 4)
			{
				if (scenario == "silent")
				{
					timespec ts{ 2, 0 }; // stays in the group, says nothing
					nanosleep(&ts, nullptr);
				}
				return 0;
			}
			rt_hip::frame_group_call call{ 1920, 1080, 0, 256, 1, 0xABCDEF };
			if (scenario == "mismatch" && rank == 1 && f == 2)
				call.seed = 2;
			group.block->ranks[rank].new_buffer = f == 1 ? 1u : 0u;
			if (rank == 0)
				group.block->call = call;
			const auto t0 = std::chrono::steady_clock::now();
			outcome o = group.enter_frame();
			if (o == outcome::ok && rank != 0 && !rt_hip::same_call(group.block->call, call))
			{
				group.break_group("rank %u was called with other arguments than rank 0", rank);
				o = outcome::broken;
			}
			if (o == outcome::ok && group.any_new_buffer())
				o = group.check_buffer(buffer);
			if (o == outcome::ok)
			{
				for (uint32_t s = rank; s * stripe < words; s += world) // "render" this rank's stripes
					for (uint32_t i = s * stripe; i < (s + 1) * stripe; i++)
						buffer[i] = pixel(f, i);
				o = group.finish_frame();
			}
			const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			if (o != outcome::ok)
			{
				// which scenarios end here, when, and with what
				const bool expected = (scenario == "private" && f == 1 && group.why_broken().find("not a mapping") != std::string::npos)
									  || (scenario == "mismatch" && f == 2 && group.why_broken().find("other arguments") != std::string::npos)
									  || (scenario == "leaves" && f == 4 && waited < 2.0 && group.why_broken().find("left the group") != std::string::npos)
									  || (scenario == "killed" && f == 4 && waited < 2.0 && group.why_broken().find("is gone") != std::string::npos)
									  || (scenario == "silent" && f == 4 && waited >= 0.35 && waited < 1.9 && group.why_broken().find("waited 400 ms") != std::string::npos);
				if (!expected)
					std::fprintf(stderr, "rank %u frame %u: outcome %d after %.3f s: %s\n", rank, f, static_cast<int>(o), waited, group.why_broken().c_str());
				return expected ? 0 : 1;
			}
			if (rank == 0) // the whole frame is there when rank 0 returns to its caller
				for (uint32_t i = 0; i < words; i++)
					if (buffer[i] != pixel(f, i))
					{
						std::fprintf(stderr, "frame %u: word %u is 0x%08x, not 0x%08x\n", f, i, buffer[i], pixel(f, i));
						return 1;
					}
		}
		if (scenario != "frames" && scenario != "stale")
		{
			std::fprintf(stderr, "rank %u: scenario %s ran to the end\n", rank, scenario.c_str());
			return 1;
		}
		return 0;
	}
}

int main(int argc, char** argv)
{
	if (argc != 3)
		return 64;
	const std::string scenario = argv[1];
	const uint32_t world = static_cast<uint32_t>(std::atoi(argv[2]));
	const std::string group_name = "/rt_hip_test_group_" + std::to_string(getpid());
	const std::string buffer_name = "/rt_hip_test_frame_" + std::to_string(getpid());
	if (!map_buffer(buffer_name.c_str(), true))
		return 65;
	if (scenario == "stale")
	{
		// what a run that crashed after its join would leave under this name: a finished child's pid as the creator
		const pid_t ghost = fork();
		if (ghost == 0)
			_exit(0);
		waitpid(ghost, nullptr, 0);
		const int fd = shm_open(group_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
		if (fd < 0 || ftruncate(fd, sizeof(rt_hip::frame_group_block)) != 0)
			return 66;
		auto* old = static_cast<rt_hip::frame_group_block*>(mmap(nullptr, sizeof(rt_hip::frame_group_block), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
		close(fd);
		old->world = world;
		old->creator_pid = static_cast<int32_t>(ghost);
		old->joined.store(world);
		old->entered.store(1000 * world), old->finished.store(1000 * world);
		old->magic.store(rt_hip::frame_group_magic);
		munmap(old, sizeof(rt_hip::frame_group_block));
	}
	std::vector<pid_t> children;
	for (uint32_t rank = 0; rank < world; rank++)
	{
		const pid_t pid = fork();
		if (pid == 0)
			_exit(run_rank(scenario, rank, world, group_name, buffer_name));
		children.push_back(pid);
	}
	int bad = 0;
	for (const pid_t pid : children)
	{
		int status = 0;
		waitpid(pid, &status, 0);
		const bool killed_on_purpose = scenario == "killed" && pid == children.back() && WIFSIGNALED(status) && WTERMSIG(status) == SIGKILL;
		if (!killed_on_purpose && (!WIFEXITED(status) || WEXITSTATUS(status) != 0))
			bad++;
	}
	shm_unlink(buffer_name.c_str());
	// the group's own name must be gone whatever happened (rank 0 removes it when the join is over)
	const int leftover = shm_open(group_name.c_str(), O_RDWR, 0600);
	if (leftover >= 0)
	{
		close(leftover);
		shm_unlink(group_name.c_str());
		std::fprintf(stderr, "%s was left behind\n", group_name.c_str());
		bad++;
	}
	if (bad)
		return 1;
	std::printf("OK: %s with %u ranks\n", scenario.c_str(), world);
	return 0;
}
