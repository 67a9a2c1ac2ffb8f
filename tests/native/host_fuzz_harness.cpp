// ASan / UBSan harness over the host-side scene front end (rt_amd/host: TOML reader, scene loader, camera) and the CPU oracle:
// every document named on the command line goes through rt_host_toml_to_json (full and short buffer), rt_host_scene_parse and,
// if it loads, rt_host_scene_describe and a tiny oracle render.  Built and run by tests/test_host_sanitizers.py.
#include "../../rt_amd/host/host_capi.h"
#include "../../oracle/cpu_ref.h"

#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

int main(int argc, char** argv)
{
	int loaded = 0, rejected = 0;
	std::vector<char> json(1 << 20);
	std::vector<uint32_t> frame(16 * 9);
	for (int i = 1; i < argc; i++)
	{
		std::ifstream file(argv[i], std::ios::binary);
		std::stringstream buffer;
		buffer << file.rdbuf();
		const std::string text = buffer.str();
		(void)rt_host_toml_to_json(text.c_str(), json.data(), json.size());
		(void)rt_host_toml_to_json(text.c_str(), json.data(), 7);
		rt_host_scene* scene = rt_host_scene_parse(text.c_str());
		if (!scene)
		{
			rejected++;
			continue;
		}
		loaded++;
		rt_host_scene_set_sampling(scene, 2, 3);
		rt_hip_scene pod;
		if (rt_host_scene_describe(scene, 16, 9, &pod) == 0)
			(void)oracle_render(&pod, 16, 9, 1, ORACLE_TRACE_ITERATIVE, nullptr, frame.data(), nullptr, 1, nullptr);
		rt_host_scene_free(scene);
	}
	std::printf("%d loaded, %d rejected, %d documents\n", loaded, rejected, argc - 1);
	return 0;
}
