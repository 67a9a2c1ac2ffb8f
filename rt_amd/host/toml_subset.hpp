// rt_amd/host/toml_subset.hpp — a small TOML reader for rt scene files.
//
// The reference parses scenes with toml++ (reference src/scene.cpp:3,527), a network-fetched dependency that is
// not available offline.  This reader covers the TOML constructs rt scene files use — and the common ones around
// them — with toml++'s observable behaviour for well-formed input:
//   key = value pairs (bare and quoted keys, dotted keys), [table] and [[array-of-tables]] headers,
//   integers (dec/hex/oct/bin, '_' separators), floats (incl. exponents, inf, nan), booleans,
//   basic "..." strings with escapes, literal '...' strings, multi-line """...""" / '''...''' strings,
//   arrays (multi-line, trailing comma, comments inside), inline tables, '#' comments.
// Not covered: dates/times.  Errors throw toml::parse_error carrying line and column.
#pragma once

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

namespace rt::toml
{
	struct parse_error : std::runtime_error
	{
		size_t line, column;
		parse_error(const std::string& what, size_t line_, size_t column_)
			: std::runtime_error{ what + " (line " + std::to_string(line_) + ", column " + std::to_string(column_) + ")" },
			  line{ line_ },
			  column{ column_ }
		{}
	};

	enum class node_type
	{
		none,
		table,
		array,
		string,
		integer,
		floating_point,
		boolean
	};

	inline std::string_view to_string(node_type t) noexcept
	{
		switch (t)
		{
			case node_type::table: return "table";
			case node_type::array: return "array";
			case node_type::string: return "string";
			case node_type::integer: return "integer";
			case node_type::floating_point: return "floating-point";
			case node_type::boolean: return "boolean";
			default: return "none";
		}
	}

	struct node
	{
		node_type type = node_type::none;
		std::string string_value;
		int64_t integer_value = 0;
		double float_value = 0.0;
		bool boolean_value = false;
		std::vector<node> elements;							// array
		std::vector<std::pair<std::string, node>> members;	// table, insertion order
		size_t line = 0, column = 0;						// where the value started
		bool inline_table = false;							// closed for extension
		bool defined_by_header = false;

		bool is_table() const noexcept { return type == node_type::table; }
		bool is_array() const noexcept { return type == node_type::array; }
		bool is_string() const noexcept { return type == node_type::string; }
		bool is_integer() const noexcept { return type == node_type::integer; }
		bool is_float() const noexcept { return type == node_type::floating_point; }
		bool is_number() const noexcept { return is_integer() || is_float(); }
		bool is_boolean() const noexcept { return type == node_type::boolean; }

		const node* get(std::string_view key) const noexcept
		{
			if (!is_table())
				return nullptr;
			for (const auto& m : members)
				if (m.first == key)
					return &m.second;
			return nullptr;
		}
		node* get(std::string_view key) noexcept { return const_cast<node*>(static_cast<const node*>(this)->get(key)); }

		size_t size() const noexcept { return is_array() ? elements.size() : (is_table() ? members.size() : 0); }
	};

	// parse a whole document; `source_name` only decorates error messages
	node parse(std::string_view document, std::string_view source_name = "<string>");
	node parse_file(const std::string& path);
}
