// rt_amd/host/image.hpp — the pixel buffer contract of the path, mirroring reference src/image.hpp / image.cpp.
//
// image: owning, 64-byte aligned buffer of uint32 RGBA8888 pixels (src/image.hpp:7-95, image.cpp:9-31);
// image_view: non-owning {pointer, size} handed to renderers (src/image.hpp:97-162) — row-major, no pitch:
// pixel (x, y) lives at data()[y * size().x + x] (:143-147) and position_of(i) = (i % w, i / w) (:155-159).
#pragma once

#include "math.hpp"

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <utility>

namespace rt
{
	class image_view;

	class image
	{
	  public:
		using pixel_type = uint32_t;
		static constexpr size_t buffer_alignment = 64;

	  private:
		pixel_type* data_ = {};
		vec2u size_ = {};

	  public:
		image() noexcept = default;

		explicit image(vec2u sz) noexcept : size_{ sz }
		{
			const size_t bytes = static_cast<size_t>(sz.x) * sz.y * sizeof(pixel_type);
			if (bytes)
				data_ = static_cast<pixel_type*>(std::aligned_alloc(buffer_alignment, (bytes + buffer_alignment - 1) / buffer_alignment * buffer_alignment));
		}

		image(image&& other) noexcept : data_{ std::exchange(other.data_, {}) }, size_{ std::exchange(other.size_, {}) } {}

		image& operator=(image&& rhs) noexcept
		{
			if (this != &rhs)
			{
				std::free(data_);
				data_ = std::exchange(rhs.data_, {});
				size_ = std::exchange(rhs.size_, {});
			}
			return *this;
		}

		~image() noexcept { std::free(data_); }

		explicit operator bool() const noexcept { return data_ && size_.x > 0 && size_.y > 0; }
		const vec2u& size() const noexcept { return size_; }
		pixel_type* data() const noexcept { return data_; }

		image& clear(uint32_t colour) noexcept
		{
			std::fill(data_, data_ + static_cast<size_t>(size_.x) * size_.y, colour);
			return *this;
		}
	};

	class image_view
	{
	  public:
		using pixel_type = image::pixel_type;

	  private:
		pixel_type* data_ = {};
		vec2u size_ = {};

	  public:
		constexpr image_view() noexcept = default;
		image_view(pixel_type* img, vec2u sz) noexcept : data_{ img }, size_{ sz } {}
		image_view(image& img) noexcept : data_{ img.data() }, size_{ img.size() } {}

		explicit constexpr operator bool() const noexcept { return data_ && size_.x > 0 && size_.y > 0; }
		constexpr const vec2u& size() const noexcept { return size_; }
		constexpr pixel_type* data() const noexcept { return data_; }
		constexpr pixel_type& operator()(unsigned x, unsigned y) const noexcept { return *(data_ + (static_cast<size_t>(y) * size_.x + x)); }
		constexpr vec2u position_of(unsigned idx) const noexcept { return { idx % size_.x, idx / size_.x }; }

		image_view& clear(uint32_t colour) noexcept
		{
			std::fill(data_, data_ + static_cast<size_t>(size_.x) * size_.y, colour);
			return *this;
		}
	};
}
