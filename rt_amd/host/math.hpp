// rt_amd/host/math.hpp — the few vector/matrix operations the host side of the path needs.
//
// Stands in for the parts of marzer/muu that reference src/camera.hpp:116-137 and src/scene.cpp:534-597 call
// (muu::vector, muu::matrix: from_translation, from_3d_rotation, from_3d_direction, perspective_projection,
// invert).  muu is not available offline, so its conventions are CHOSEN here:
// right-handed world, forward = -Z, up = +Y, right = +X; clip-space depth 0 at the near plane, 1 at the far plane.
// All of it is plain binary32 arithmetic like muu's (no double intermediates).
#pragma once

#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>

namespace rt
{
	struct vec2u
	{
		unsigned x{}, y{};
	};

	struct vec3
	{
		float x{}, y{}, z{};

		struct constants
		{
			static constexpr std::array<float, 3> forward{ 0, 0, -1 };
		};

		static vec3 normalize(vec3 v) noexcept
		{
			const float inv = 1.0f / std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
			return { v.x * inv, v.y * inv, v.z * inv };
		}
		static float dot(vec3 a, vec3 b) noexcept { return a.x * b.x + a.y * b.y + a.z * b.z; }
		static vec3 cross(vec3 a, vec3 b) noexcept { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
	};

	inline vec3 operator+(vec3 a, vec3 b) noexcept { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
	inline vec3 operator-(vec3 a, vec3 b) noexcept { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
	inline vec3 operator-(vec3 a) noexcept { return { -a.x, -a.y, -a.z }; }
	inline vec3 operator*(vec3 a, float s) noexcept { return { a.x * s, a.y * s, a.z * s }; }

	inline constexpr vec3 forward_axis{ 0, 0, -1 };
	inline constexpr vec3 up_axis{ 0, 1, 0 };
	inline constexpr vec3 right_axis{ 1, 0, 0 };

	// 3x3, m(r, c)
	struct mat3
	{
		float m[3][3]{ { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };

		float& operator()(size_t r, size_t c) noexcept { return m[r][c]; }
		float operator()(size_t r, size_t c) const noexcept { return m[r][c]; }

		vec3 transform_direction(vec3 v) const noexcept
		{
			return { m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z,
					 m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
					 m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z };
		}

		// muu::orthonormalize(mat3) as rt's camera::pose applies it (reference src/camera.hpp:108): the columns — the images of
		// +X, +Y, +Z — made orthogonal and of unit length by Gram-Schmidt, in that order.  (muu itself is not readable here:
		// the ORDER is this mirror's choice.  For a matrix that already is a rotation up to rounding the result differs from the
		// input in the last bits only; what matters to the plug-in is that the pose is re-normalised at all, as the reference's is —
		// a rotation accumulated over many mouse moves would otherwise drift.)
		static mat3 orthonormalize(const mat3& in) noexcept
		{
			vec3 x = { in.m[0][0], in.m[1][0], in.m[2][0] }, y = { in.m[0][1], in.m[1][1], in.m[2][1] }, z = { in.m[0][2], in.m[1][2], in.m[2][2] };
			const auto dot = [](vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; };
			x = vec3::normalize(x);
			const float yx = dot(y, x);
			y = vec3::normalize({ y.x - yx * x.x, y.y - yx * x.y, y.z - yx * x.z });
			const float zx = dot(z, x), zy = dot(z, y);
			z = vec3::normalize({ z.x - zx * x.x - zy * y.x, z.y - zx * x.y - zy * y.y, z.z - zx * x.z - zy * y.z });
			mat3 out;
			out.m[0][0] = x.x, out.m[1][0] = x.y, out.m[2][0] = x.z;
			out.m[0][1] = y.x, out.m[1][1] = y.y, out.m[2][1] = y.z;
			out.m[0][2] = z.x, out.m[1][2] = z.y, out.m[2][2] = z.z;
			return out;
		}

		// rotation that takes `forward` (-Z) to `dir` and keeps +Y as "up" as far as possible
		static mat3 from_3d_direction(vec3 dir) noexcept
		{
			const vec3 f = vec3::normalize(dir);
			vec3 r = vec3::cross(f, up_axis);
			if (r.x * r.x + r.y * r.y + r.z * r.z < 1.0e-12f) // looking straight up or down
				r = right_axis;
			r = vec3::normalize(r);
			const vec3 u = vec3::cross(r, f);
			mat3 out;
			// columns: image of +X, +Y, +Z
			out.m[0][0] = r.x, out.m[1][0] = r.y, out.m[2][0] = r.z;
			out.m[0][1] = u.x, out.m[1][1] = u.y, out.m[2][1] = u.z;
			out.m[0][2] = -f.x, out.m[1][2] = -f.y, out.m[2][2] = -f.z;
			return out;
		}
	};

	// 4x4, m(r, c); transforms column vectors: (M v)_r = sum_c m(r,c) v_c
	struct mat4
	{
		float m[4][4]{ { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 0, 1 } };

		float& operator()(size_t r, size_t c) noexcept { return m[r][c]; }
		float operator()(size_t r, size_t c) const noexcept { return m[r][c]; }

		static mat4 from_translation(vec3 t) noexcept
		{
			mat4 out;
			out.m[0][3] = t.x, out.m[1][3] = t.y, out.m[2][3] = t.z;
			return out;
		}

		static mat4 from_3d_rotation(const mat3& rot) noexcept
		{
			mat4 out;
			for (size_t r = 0; r < 3; r++)
				for (size_t c = 0; c < 3; c++)
					out.m[r][c] = rot(r, c);
			return out;
		}

		// right-handed, looking down -Z, depth 0 (near) .. 1 (far)
		static mat4 perspective_projection(float vertical_fov, float aspect_ratio, float near_clip, float far_clip) noexcept
		{
			const float f = 1.0f / std::tan(vertical_fov * 0.5f);
			mat4 out;
			out.m[0][0] = f / aspect_ratio;
			out.m[1][1] = f;
			out.m[2][2] = far_clip / (near_clip - far_clip);
			out.m[2][3] = (near_clip * far_clip) / (near_clip - far_clip);
			out.m[3][2] = -1.0f;
			out.m[3][3] = 0.0f;
			return out;
		}

		friend mat4 operator*(const mat4& a, const mat4& b) noexcept
		{
			mat4 out;
			for (size_t r = 0; r < 4; r++)
				for (size_t c = 0; c < 4; c++)
				{
					float acc = 0.0f;
					for (size_t k = 0; k < 4; k++)
						acc += a.m[r][k] * b.m[k][c];
					out.m[r][c] = acc;
				}
			return out;
		}

		// general inverse by cofactors
		static mat4 invert(const mat4& in) noexcept
		{
			const float* a = &in.m[0][0];
			float inv[16];
			inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
			inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
			inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
			inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
			inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
			inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
			inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
			inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
			inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
			inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
			inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
			inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
			inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
			inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
			inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
			inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
			const float det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
			const float inv_det = 1.0f / det;
			mat4 out;
			for (size_t i = 0; i < 16; i++)
				(&out.m[0][0])[i] = inv[i] * inv_det;
			return out;
		}

		// (M * (v, 1)).xyz / .w
		vec3 transform_position(vec3 v) const noexcept
		{
			float row[4];
			for (size_t r = 0; r < 4; r++)
				row[r] = m[r][0] * v.x + m[r][1] * v.y + m[r][2] * v.z + m[r][3];
			const float inv_w = 1.0f / row[3];
			return { row[0] * inv_w, row[1] * inv_w, row[2] * inv_w };
		}
	};
}
