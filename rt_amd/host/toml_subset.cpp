// rt_amd/host/toml_subset.cpp — see toml_subset.hpp.
#include "toml_subset.hpp"

#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <limits>
#include <sstream>

namespace rt::toml
{
	namespace
	{
		class parser
		{
			std::string_view src_;
			std::string source_name_;
			size_t pos_ = 0, line_ = 1, col_ = 1;

			[[noreturn]] void fail(const std::string& what) const { throw parse_error{ source_name_ + ": " + what, line_, col_ }; }

			bool eof() const noexcept { return pos_ >= src_.size(); }
			char peek(size_t ahead = 0) const noexcept { return pos_ + ahead < src_.size() ? src_[pos_ + ahead] : '\0'; }
			char advance() noexcept
			{
				const char c = src_[pos_++];
				if (c == '\n')
					line_++, col_ = 1;
				else
					col_++;
				return c;
			}
			bool starts_with(std::string_view s) const noexcept { return src_.substr(pos_, s.size()) == s; }

			void skip_spaces() noexcept
			{
				while (!eof() && (peek() == ' ' || peek() == '\t'))
					advance();
			}
			void skip_comment() noexcept
			{
				if (peek() == '#')
					while (!eof() && peek() != '\n')
						advance();
			}
			// whitespace, newlines and comments (inside arrays)
			void skip_blank() noexcept
			{
				while (!eof())
				{
					const char c = peek();
					if (c == ' ' || c == '\t' || c == '\r' || c == '\n')
						advance();
					else if (c == '#')
						skip_comment();
					else
						break;
				}
			}
			void expect_line_end()
			{
				skip_spaces();
				skip_comment();
				if (eof())
					return;
				if (peek() == '\r')
					advance();
				if (eof())
					return;
				if (peek() != '\n')
					fail(std::string{ "expected end of line, saw '" } + peek() + "'");
				advance();
			}

			static bool is_bare_key_char(char c) noexcept
			{
				return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_' || c == '-';
			}

			void append_utf8(std::string& out, uint32_t cp)
			{
				if (cp < 0x80)
					out += static_cast<char>(cp);
				else if (cp < 0x800)
					out += static_cast<char>(0xC0 | (cp >> 6)), out += static_cast<char>(0x80 | (cp & 0x3F));
				else if (cp < 0x10000)
					out += static_cast<char>(0xE0 | (cp >> 12)), out += static_cast<char>(0x80 | ((cp >> 6) & 0x3F)),
						out += static_cast<char>(0x80 | (cp & 0x3F));
				else
					out += static_cast<char>(0xF0 | (cp >> 18)), out += static_cast<char>(0x80 | ((cp >> 12) & 0x3F)),
						out += static_cast<char>(0x80 | ((cp >> 6) & 0x3F)), out += static_cast<char>(0x80 | (cp & 0x3F));
			}

			std::string parse_basic_string()
			{
				const bool multi = starts_with("\"\"\"");
				advance();
				if (multi)
				{
					advance(), advance();
					if (peek() == '\r')
						advance();
					if (peek() == '\n')
						advance();
				}
				std::string out;
				while (true)
				{
					if (eof())
						fail("unterminated string");
					if (multi && starts_with("\"\"\""))
					{
						advance(), advance(), advance();
						// up to two quotes directly in front of the closing delimiter belong to the string
						for (int extra = 0; extra < 2 && peek() == '"'; extra++)
						{
							out += '"';
							advance();
						}
						return out;
					}
					char c = advance();
					if (!multi && c == '"')
						return out;
					if (!multi && c == '\n')
						fail("newline in single-line string");
					if (c != '\\')
					{
						out += c;
						continue;
					}
					if (eof())
						fail("unterminated escape");
					c = advance();
					switch (c)
					{
						case 'b': out += '\b'; break;
						case 't': out += '\t'; break;
						case 'n': out += '\n'; break;
						case 'f': out += '\f'; break;
						case 'r': out += '\r'; break;
						case 'e': out += '\x1B'; break;
						case '"': out += '"'; break;
						case '\\': out += '\\'; break;
						case 'u':
						case 'U':
						{
							const int digits = c == 'u' ? 4 : 8;
							uint32_t cp = 0;
							for (int i = 0; i < digits; i++)
							{
								const char h = eof() ? '\0' : advance();
								cp <<= 4;
								if (h >= '0' && h <= '9')
									cp |= static_cast<uint32_t>(h - '0');
								else if (h >= 'a' && h <= 'f')
									cp |= static_cast<uint32_t>(h - 'a' + 10);
								else if (h >= 'A' && h <= 'F')
									cp |= static_cast<uint32_t>(h - 'A' + 10);
								else
									fail("bad unicode escape");
							}
							append_utf8(out, cp);
							break;
						}
						case ' ':
						case '\t':
						case '\r':
						case '\n':
							if (!multi)
								fail("bad escape");
							// line-ending backslash: trim following whitespace
							while (!eof() && (peek() == ' ' || peek() == '\t' || peek() == '\r' || peek() == '\n'))
								advance();
							break;
						default: fail(std::string{ "unknown escape '\\" } + c + "'");
					}
				}
			}

			std::string parse_literal_string()
			{
				const bool multi = starts_with("'''");
				advance();
				if (multi)
				{
					advance(), advance();
					if (peek() == '\r')
						advance();
					if (peek() == '\n')
						advance();
				}
				std::string out;
				while (true)
				{
					if (eof())
						fail("unterminated string");
					if (multi && starts_with("'''"))
					{
						advance(), advance(), advance();
						return out;
					}
					const char c = advance();
					if (!multi && c == '\'')
						return out;
					if (!multi && c == '\n')
						fail("newline in single-line string");
					out += c;
				}
			}

			std::string parse_key_part()
			{
				skip_spaces();
				if (peek() == '"')
					return parse_basic_string();
				if (peek() == '\'')
					return parse_literal_string();
				std::string out;
				while (!eof() && is_bare_key_char(peek()))
					out += advance();
				if (out.empty())
					fail(std::string{ "expected a key, saw '" } + peek() + "'");
				return out;
			}

			std::vector<std::string> parse_key()
			{
				std::vector<std::string> parts;
				parts.push_back(parse_key_part());
				skip_spaces();
				while (peek() == '.')
				{
					advance();
					parts.push_back(parse_key_part());
					skip_spaces();
				}
				return parts;
			}

			node parse_number_or_keyword()
			{
				node n;
				n.line = line_, n.column = col_;
				size_t end = pos_;
				while (end < src_.size())
				{
					const char c = src_[end];
					if ((c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '+' || c == '-' || c == '.' || c == '_')
						end++;
					else
						break;
				}
				const std::string_view tok = src_.substr(pos_, end - pos_);
				if (tok.empty())
					fail(std::string{ "expected a value, saw '" } + peek() + "'");
				const auto consume = [&]()
				{
					while (pos_ < end)
						advance();
				};
				if (tok == "true" || tok == "false")
				{
					n.type = node_type::boolean;
					n.boolean_value = tok == "true";
					consume();
					return n;
				}
				std::string_view body = tok;
				bool negative = false;
				if (body[0] == '+' || body[0] == '-')
				{
					negative = body[0] == '-';
					body.remove_prefix(1);
				}
				if (body == "inf" || body == "nan")
				{
					n.type = node_type::floating_point;
					n.float_value = body == "inf" ? std::numeric_limits<double>::infinity() : std::numeric_limits<double>::quiet_NaN();
					if (negative)
						n.float_value = -n.float_value;
					consume();
					return n;
				}
				if (body.empty() || !(body[0] >= '0' && body[0] <= '9'))
					fail("malformed value '" + std::string{ tok } + "'");
				const auto is_digit_like = [](char c) noexcept
				{ return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'f') || (c >= 'A' && c <= 'F'); };
				std::string clean;
				for (size_t i = 0; i < body.size(); i++)
				{
					if (body[i] == '_')
					{
						// TOML 1.0: an underscore sits between two digits
						if (i == 0 || i + 1 == body.size() || !is_digit_like(body[i - 1]) || !is_digit_like(body[i + 1]))
							fail("misplaced '_' in number '" + std::string{ tok } + "'");
						continue;
					}
					clean += body[i];
				}
				int base = 10;
				if (clean.size() > 2 && clean[0] == '0' && (clean[1] == 'x' || clean[1] == 'o' || clean[1] == 'b'))
				{
					base = clean[1] == 'x' ? 16 : (clean[1] == 'o' ? 8 : 2);
					if (tok[0] == '+' || tok[0] == '-')
						fail("prefixed integers may not have a sign");
					if (body[2] == '_')
						fail("misplaced '_' in number '" + std::string{ tok } + "'");
					clean = clean.substr(2);
				}
				if (base == 10)
				{
					// TOML 1.0 decimal grammar: int = "0" | [1-9][0-9]*; frac = "." [0-9]+; exp = [eE] [+-]? [0-9]+
					size_t i = 0;
					const auto digits = [&]()
					{
						const size_t start = i;
						while (i < clean.size() && clean[i] >= '0' && clean[i] <= '9')
							i++;
						return i - start;
					};
					const size_t int_digits = digits();
					bool ok = int_digits > 0 && (int_digits == 1 || clean[0] != '0');
					if (ok && i < clean.size() && clean[i] == '.')
					{
						i++;
						ok = digits() > 0;
					}
					if (ok && i < clean.size() && (clean[i] == 'e' || clean[i] == 'E'))
					{
						i++;
						if (i < clean.size() && (clean[i] == '+' || clean[i] == '-'))
							i++;
						ok = digits() > 0;
					}
					if (!ok || i != clean.size())
						fail("malformed number '" + std::string{ tok } + "'");
				}
				const bool is_float = base == 10 && clean.find_first_of(".eE") != std::string::npos;
				errno = 0;
				char* parsed_end = nullptr;
				if (is_float)
				{
					n.type = node_type::floating_point;
					n.float_value = std::strtod(clean.c_str(), &parsed_end);
					if (negative)
						n.float_value = -n.float_value;
				}
				else
				{
					n.type = node_type::integer;
					const unsigned long long v = std::strtoull(clean.c_str(), &parsed_end, base);
					if (errno == ERANGE || v > (negative ? 9223372036854775808ull : 9223372036854775807ull))
						fail("integer out of range '" + std::string{ tok } + "'");
					n.integer_value = negative ? static_cast<int64_t>(0ull - v) : static_cast<int64_t>(v);
				}
				if (!parsed_end || *parsed_end != '\0' || parsed_end == clean.c_str())
					fail("malformed number '" + std::string{ tok } + "'");
				consume();
				return n;
			}

			node parse_array()
			{
				node n;
				n.type = node_type::array;
				n.line = line_, n.column = col_;
				advance(); // [
				while (true)
				{
					skip_blank();
					if (eof())
						fail("unterminated array");
					if (peek() == ']')
					{
						advance();
						return n;
					}
					n.elements.push_back(parse_value());
					skip_blank();
					if (peek() == ',')
					{
						advance();
						continue;
					}
					if (peek() == ']')
					{
						advance();
						return n;
					}
					fail(std::string{ "expected ',' or ']' in array, saw '" } + peek() + "'");
				}
			}

			node parse_inline_table()
			{
				node n;
				n.type = node_type::table;
				n.inline_table = true;
				n.line = line_, n.column = col_;
				advance(); // {
				skip_spaces(); // TOML 1.0 (what toml++ reads by default): an inline table stays on one line, no trailing comma
				if (peek() == '}')
				{
					advance();
					return n;
				}
				while (true)
				{
					skip_spaces();
					const auto key = parse_key();
					skip_spaces();
					if (peek() != '=')
						fail("expected '=' after key in inline table");
					advance();
					skip_spaces();
					insert(n, key, parse_value());
					skip_spaces();
					if (peek() == ',')
					{
						advance();
						continue;
					}
					if (peek() == '}')
					{
						advance();
						return n;
					}
					fail(std::string{ "expected ',' or '}' in inline table, saw '" } + peek() + "'");
				}
			}

			node parse_value()
			{
				skip_spaces();
				node n;
				n.line = line_, n.column = col_;
				const char c = peek();
				if (c == '"')
				{
					n.type = node_type::string;
					n.string_value = parse_basic_string();
					return n;
				}
				if (c == '\'')
				{
					n.type = node_type::string;
					n.string_value = parse_literal_string();
					return n;
				}
				if (c == '[')
					return parse_array();
				if (c == '{')
					return parse_inline_table();
				return parse_number_or_keyword();
			}

			// walk/create intermediate tables for a dotted key, then insert
			void insert(node& table, const std::vector<std::string>& key, node value)
			{
				node* cur = &table;
				for (size_t i = 0; i + 1 < key.size(); i++)
				{
					node* next = cur->get(key[i]);
					if (!next)
					{
						node t;
						t.type = node_type::table;
						t.line = line_, t.column = col_;
						cur->members.emplace_back(key[i], std::move(t));
						next = &cur->members.back().second;
					}
					else if (!next->is_table() || next->inline_table)
						fail("key '" + key[i] + "' is not an extendable table");
					cur = next;
				}
				if (cur->get(key.back()))
					fail("duplicate key '" + key.back() + "'");
				cur->members.emplace_back(key.back(), std::move(value));
			}

			node* open_table(node& root, const std::vector<std::string>& key, bool array_of_tables)
			{
				node* cur = &root;
				for (size_t i = 0; i < key.size(); i++)
				{
					const bool last = i + 1 == key.size();
					node* next = cur->get(key[i]);
					if (!next)
					{
						node t;
						t.line = line_, t.column = col_;
						if (last && array_of_tables)
						{
							t.type = node_type::array;
							node first;
							first.type = node_type::table;
							first.line = line_, first.column = col_;
							t.elements.push_back(std::move(first));
							cur->members.emplace_back(key[i], std::move(t));
							return &cur->members.back().second.elements.back();
						}
						t.type = node_type::table;
						t.defined_by_header = last;
						cur->members.emplace_back(key[i], std::move(t));
						cur = &cur->members.back().second;
						continue;
					}
					if (next->is_array())
					{
						if (next->elements.empty() || !next->elements.back().is_table())
							fail("key '" + key[i] + "' is not an array of tables");
						if (last)
						{
							if (!array_of_tables)
								fail("'" + key[i] + "' was already defined as an array");
							node t;
							t.type = node_type::table;
							t.line = line_, t.column = col_;
							next->elements.push_back(std::move(t));
						}
						cur = &next->elements.back();
						continue;
					}
					if (!next->is_table() || next->inline_table)
						fail("key '" + key[i] + "' is not a table");
					if (last)
					{
						if (array_of_tables)
							fail("'" + key[i] + "' was already defined as a table");
						if (next->defined_by_header)
							fail("table '" + key[i] + "' defined twice");
						next->defined_by_header = true;
					}
					cur = next;
				}
				return cur;
			}

		  public:
			parser(std::string_view src, std::string_view source_name) : src_{ src }, source_name_{ source_name }
			{
				if (src_.substr(0, 3) == "\xEF\xBB\xBF")
					pos_ = 3;
			}

			node run()
			{
				node root;
				root.type = node_type::table;
				root.line = 1, root.column = 1;
				node* current = &root;
				while (true)
				{
					skip_blank();
					if (eof())
						break;
					if (peek() == '[')
					{
						const bool aot = peek(1) == '[';
						advance();
						if (aot)
							advance();
						const auto key = parse_key();
						skip_spaces();
						if (peek() != ']')
							fail("expected ']' closing table header");
						advance();
						if (aot)
						{
							if (peek() != ']')
								fail("expected ']]' closing array-of-tables header");
							advance();
						}
						current = open_table(root, key, aot);
						expect_line_end();
						continue;
					}
					const auto key = parse_key();
					skip_spaces();
					if (peek() != '=')
						fail("expected '=' after key '" + key.back() + "'");
					advance();
					insert(*current, key, parse_value());
					expect_line_end();
				}
				return root;
			}
		};
	}

	node parse(std::string_view document, std::string_view source_name)
	{
		return parser{ document, source_name }.run();
	}

	node parse_file(const std::string& path)
	{
		std::ifstream file{ path, std::ios::binary };
		if (!file)
			throw parse_error{ "could not open '" + path + "'", 0, 0 };
		std::ostringstream text;
		text << file.rdbuf();
		return parse(text.str(), path);
	}
}
