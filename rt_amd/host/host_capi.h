/* rt_amd/host/host_capi.h — C entry points of the host-side scene/camera code (librt_host.so), for the
 * Python test and benchmark harness (ctypes).  Pure CPU: no HIP dependency.  NOT the drop-in boundary
 * (that is include/rt_hip.h); in a real rt build the reference's own scene/camera code plays this role. */
#ifndef RT_HOST_CAPI_H
#define RT_HOST_CAPI_H

#include "../../include/rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_host_scene rt_host_scene;

/* message of the last failure on this thread ("" if none) */
const char* rt_host_last_error(void);

/* scene::parse / scene::load / scene::synthetic; NULL on failure (see rt_host_last_error) */
rt_host_scene* rt_host_scene_parse(const char* toml_text);
rt_host_scene* rt_host_scene_load(const char* path);
rt_host_scene* rt_host_scene_synthetic(unsigned sphere_count);
void rt_host_scene_free(rt_host_scene* scene);

/* override scene.samples_per_pixel / scene.max_bounces (0 = keep).  The reference has no such flags;
 * the benchmark configurations need them. */
void rt_host_scene_set_sampling(rt_host_scene* scene, unsigned samples_per_pixel, unsigned max_bounces);
/* camera.pose(position, direction) */
void rt_host_scene_set_camera(rt_host_scene* scene, const float position[3], const float direction[3]);

/* Fill the ABI scene for a width x height frame: column pointers into `scene` (valid until it is freed or
 * modified) and camera.viewport({width,height}).inverse_view_projection as m(r,c) -> [r*4+c]. */
int rt_host_scene_describe(const rt_host_scene* scene, unsigned width, unsigned height, rt_hip_scene* out);

/* viewport::screen_to_world for tests */
void rt_host_screen_to_world(const rt_host_scene* scene, unsigned width, unsigned height, float x, float y, float depth, float out[3]);

/* The TOML reader on its own (tests cross-check it against an independent parser): parse `toml_text` and write the
 * document as JSON into `out` (size `capacity`, always NUL-terminated).  Integers as JSON numbers, floats with 17
 * significant digits (inf / nan as strings "inf", "-inf", "nan"), tables as objects in insertion order.
 * Returns the number of bytes needed (excluding the NUL), or -1 on a parse error (see rt_host_last_error). */
long rt_host_toml_to_json(const char* toml_text, char* out, unsigned long capacity);

/* named colour lookup (1 = found) */
int rt_host_named_colour(const char* name, float out_rgba[4]);

#ifdef __cplusplus
}
#endif
#endif
