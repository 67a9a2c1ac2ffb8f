// rt_amd/host/soa.hpp — struct-of-arrays tables with the column names of the reference's soagen tables.
//
// Mirrors rt::materials / rt::planes / rt::spheres / rt::boxes as the path reads them (reference src/soa.toml:6-45;
// generated accessors src/soa.hpp:177-199): named column accessors returning raw pointers, size(), push_back
// with one argument per column in declaration order (as used at reference src/scene.cpp:558-562,583,594-595).
// Float and index columns are 32-byte aligned like the reference's (`alignment = 32`, src/soa.toml:17-22,27-32).
// The AoS `value` column of the reference (muu::bounding_sphere / muu::plane) is kept as a 4-float struct.
#pragma once

#include "math.hpp"

#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace rt
{
	// material kinds, reference src/common.hpp:105-115
	enum class material_type : unsigned
	{
		lambert,
		metal,
		dielectric,
		air,
		vacuum,
		water,
		ice,
		diamond,
	};

	// rgba floats, reference src/colour.hpp:17-57
	struct colour
	{
		float r{}, g{}, b{}, a{ 1.0f };
	};

	struct sphere // muu::bounding_sphere<float>
	{
		vec3 center;
		float radius;
	};

	struct box // muu::bounding_box<float>: centre and half sizes
	{
		vec3 center;
		vec3 extents;
	};

	struct plane // muu::plane<float>: normal . p + d = 0
	{
		vec3 normal;
		float d;

		plane() = default;
		plane(vec3 n, float d_) : normal{ n }, d{ d_ } {}
		// from a point on the plane and a unit normal (reference src/scene.cpp:580-581)
		plane(vec3 position, vec3 unit_normal) : normal{ unit_normal }, d{ -vec3::dot(position, unit_normal) } {}
	};

	namespace detail
	{
		// growable, 32-byte aligned column of a trivially copyable type
		template <typename T>
		class column
		{
			T* data_ = nullptr;
			size_t size_ = 0, capacity_ = 0;

			void grow()
			{
				const size_t cap = capacity_ ? capacity_ * 2 : 8; // soagen rounds capacity to 8 rows for these columns
				void* p = std::aligned_alloc(32, ((cap * sizeof(T) + 31) / 32) * 32);
				if (!p)
					throw std::bad_alloc{};
				if (size_)
					std::memcpy(p, data_, size_ * sizeof(T));
				std::free(data_);
				data_ = static_cast<T*>(p);
				capacity_ = cap;
			}

		  public:
			column() = default;
			column(const column& o) { *this = o; }
			column(column&& o) noexcept : data_{ o.data_ }, size_{ o.size_ }, capacity_{ o.capacity_ } { o.data_ = nullptr, o.size_ = o.capacity_ = 0; }
			column& operator=(const column& o)
			{
				if (this != &o)
				{
					size_ = 0;
					for (size_t i = 0; i < o.size_; i++)
						push_back(o.data_[i]);
				}
				return *this;
			}
			column& operator=(column&& o) noexcept
			{
				if (this != &o)
				{
					std::free(data_);
					data_ = o.data_, size_ = o.size_, capacity_ = o.capacity_;
					o.data_ = nullptr, o.size_ = o.capacity_ = 0;
				}
				return *this;
			}
			~column() { std::free(data_); }

			void push_back(const T& v)
			{
				if (size_ == capacity_)
					grow();
				data_[size_++] = v;
			}
			T* data() noexcept { return data_; }
			const T* data() const noexcept { return data_; }
			size_t size() const noexcept { return size_; }
		};
	}

	class materials
	{
		std::vector<std::string> name_;
		detail::column<material_type> type_;
		detail::column<colour> albedo_;
		detail::column<float> roughness_, reflectivity_;

	  public:
		size_t size() const noexcept { return type_.size(); }
		bool empty() const noexcept { return !size(); }
		void push_back(std::string name, material_type type, colour albedo, float roughness, float reflectivity)
		{
			name_.push_back(std::move(name));
			type_.push_back(type);
			albedo_.push_back(albedo);
			roughness_.push_back(roughness);
			reflectivity_.push_back(reflectivity);
		}
		const std::string* name() const noexcept { return name_.data(); }
		const material_type* type() const noexcept { return type_.data(); }
		const colour* albedo() const noexcept { return albedo_.data(); }
		const float* roughness() const noexcept { return roughness_.data(); }
		const float* reflectivity() const noexcept { return reflectivity_.data(); }
	};

	class planes
	{
		detail::column<plane> value_;
		detail::column<unsigned> material_;
		detail::column<float> normal_x_, normal_y_, normal_z_, d_;

	  public:
		size_t size() const noexcept { return value_.size(); }
		void push_back(const plane& value, unsigned material, float nx, float ny, float nz, float d)
		{
			value_.push_back(value);
			material_.push_back(material);
			normal_x_.push_back(nx), normal_y_.push_back(ny), normal_z_.push_back(nz), d_.push_back(d);
		}
		const plane* value() const noexcept { return value_.data(); }
		const unsigned* material() const noexcept { return material_.data(); }
		const float* normal_x() const noexcept { return normal_x_.data(); }
		const float* normal_y() const noexcept { return normal_y_.data(); }
		const float* normal_z() const noexcept { return normal_z_.data(); }
		const float* d() const noexcept { return d_.data(); }
	};

	class spheres
	{
		detail::column<sphere> value_;
		detail::column<unsigned> material_;
		detail::column<float> center_x_, center_y_, center_z_, radius_;

	  public:
		size_t size() const noexcept { return value_.size(); }
		void push_back(const sphere& value, unsigned material, float cx, float cy, float cz, float radius)
		{
			value_.push_back(value);
			material_.push_back(material);
			center_x_.push_back(cx), center_y_.push_back(cy), center_z_.push_back(cz), radius_.push_back(radius);
		}
		const sphere* value() const noexcept { return value_.data(); }
		const unsigned* material() const noexcept { return material_.data(); }
		const float* center_x() const noexcept { return center_x_.data(); }
		const float* center_y() const noexcept { return center_y_.data(); }
		const float* center_z() const noexcept { return center_z_.data(); }
		const float* radius() const noexcept { return radius_.data(); }
	};

	class boxes // drawn by the preview only; mg_ray_tracer's test_boxes never hits (mg_ray_tracer.cpp:89-93)
	{
		detail::column<box> value_;
		detail::column<unsigned> material_;
		detail::column<float> center_x_, center_y_, center_z_, extents_x_, extents_y_, extents_z_;

	  public:
		size_t size() const noexcept { return value_.size(); }
		void push_back(const box& value, unsigned material, float cx, float cy, float cz, float ex, float ey, float ez)
		{
			value_.push_back(value);
			material_.push_back(material);
			center_x_.push_back(cx), center_y_.push_back(cy), center_z_.push_back(cz);
			extents_x_.push_back(ex), extents_y_.push_back(ey), extents_z_.push_back(ez);
		}
		const box* value() const noexcept { return value_.data(); }
		const unsigned* material() const noexcept { return material_.data(); }
		const float* center_x() const noexcept { return center_x_.data(); }
		const float* center_y() const noexcept { return center_y_.data(); }
		const float* center_z() const noexcept { return center_z_.data(); }
		const float* extents_x() const noexcept { return extents_x_.data(); }
		const float* extents_y() const noexcept { return extents_y_.data(); }
		const float* extents_z() const noexcept { return extents_z_.data(); }
	};
}
