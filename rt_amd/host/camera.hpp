// rt_amd/host/camera.hpp — the camera the renderer consumes, mirroring reference src/camera.hpp.
//
// Same interface for the members the path uses: rt::camera{pose, position, rotation, viewport(size)} and
// rt::viewport{position, size, view, projection, view_projection, inverse_view_projection, screen_to_world}
// (reference src/camera.hpp:7-49, 51-138).  Defaults match src/camera.hpp:54-58: vfov pi/4, near 0.01, far 1000,
// position (0, 1, 0).
#pragma once

#include "math.hpp"

namespace rt
{
	struct viewport
	{
		vec3 position;
		vec2u size;
		mat4 view;
		mat4 projection;
		mat4 view_projection;
		mat4 inverse_view_projection;

		// reference src/camera.hpp:42-48
		vec3 screen_to_world(float screen_x, float screen_y, float depth = 0.0f) const noexcept
		{
			const vec3 pos_in_view_space = { 2.0f * (screen_x / static_cast<float>(size.x)) - 1.0f,
											 -2.0f * (screen_y / static_cast<float>(size.y)) + 1.0f,
											 depth };
			return inverse_view_projection.transform_position(pos_in_view_space);
		}
	};

	class camera
	{
	  private:
		float vfov_ = 0.78539816339744830962f; // pi / 4
		vec3 pos_ = { 0, 1, 0 };
		mat3 rot_{};
		float near_ = 0.01f;
		float far_ = 1000.0f;

	  public:
		const vec3& position() const noexcept { return pos_; }
		const mat3& rotation() const noexcept { return rot_; }

		// reference src/camera.hpp:103-113 (its assertions on NaN / infinity left out)
		camera& pose(const vec3& pos, const mat3& rot) noexcept
		{
			pos_ = pos;
			rot_ = mat3::orthonormalize(rot);
			return *this;
		}

		// reference src/camera.hpp:116-119
		camera& pose(const vec3& pos, const vec3& dir) noexcept { return pose(pos, mat3::from_3d_direction(vec3::normalize(dir))); }

		// reference src/camera.hpp:122-137
		rt::viewport viewport(vec2u size) const noexcept
		{
			rt::viewport vp{};
			vp.position = pos_;
			vp.size = size;
			vp.view = mat4::invert(mat4::from_translation(pos_) * mat4::from_3d_rotation(rot_));
			vp.projection = mat4::perspective_projection(vfov_, static_cast<float>(size.x) / static_cast<float>(size.y), near_, far_);
			vp.view_projection = vp.projection * vp.view;
			vp.inverse_view_projection = mat4::invert(vp.view_projection);
			return vp;
		}
	};
}
