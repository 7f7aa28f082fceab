#include "json.h"

#include <cctype>
#include <cstdlib>

namespace rt {

json_value const* json_value::at_path(std::string const& path) const {
    json_value const* cur = this;
    size_t pos = 0;
    while (pos <= path.size() && cur) {
        size_t const slash = path.find('/', pos);
        std::string const token =
            path.substr(pos, slash == std::string::npos ? std::string::npos : slash - pos);
        if (!token.empty()) {
            if (cur->type == kind::object) {
                cur = cur->find(token);
            } else if (cur->type == kind::array) {
                char* end = nullptr;
                unsigned long const idx = std::strtoul(token.c_str(), &end, 10);
                cur = (*end == '\0' && idx < cur->array.size()) ? &cur->array[idx] : nullptr;
            } else {
                cur = nullptr;
            }
        }
        if (slash == std::string::npos) {
            break;
        }
        pos = slash + 1;
    }
    return cur;
}

namespace {

struct parser {
    std::string const& text;
    size_t pos = 0;

    [[noreturn]] void fail(char const* what) const {
        throw std::runtime_error(
            std::string{"json: "} + what + " at offset " + std::to_string(pos));
    }

    void skip_space() {
        while (pos < text.size() && std::isspace((unsigned char) text[pos])) {
            ++pos;
        }
    }

    char peek() {
        skip_space();
        if (pos >= text.size()) {
            fail("unexpected end of input");
        }
        return text[pos];
    }

    void expect(char c) {
        if (peek() != c) {
            fail("unexpected character");
        }
        ++pos;
    }

    static void append_utf8(std::string& out, unsigned cp) {
        if (cp < 0x80) {
            out.push_back((char) cp);
        } else if (cp < 0x800) {
            out.push_back((char) (0xC0 | (cp >> 6)));
            out.push_back((char) (0x80 | (cp & 0x3F)));
        } else {
            out.push_back((char) (0xE0 | (cp >> 12)));
            out.push_back((char) (0x80 | ((cp >> 6) & 0x3F)));
            out.push_back((char) (0x80 | (cp & 0x3F)));
        }
    }

    std::string parse_string() {
        expect('"');
        std::string out;
        while (true) {
            if (pos >= text.size()) {
                fail("unterminated string");
            }
            char const c = text[pos++];
            if (c == '"') {
                break;
            }
            if (c != '\\') {
                out.push_back(c);
                continue;
            }
            if (pos >= text.size()) {
                fail("unterminated escape");
            }
            char const e = text[pos++];
            switch (e) {
                case '"': out.push_back('"'); break;
                case '\\': out.push_back('\\'); break;
                case '/': out.push_back('/'); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'n': out.push_back('\n'); break;
                case 'r': out.push_back('\r'); break;
                case 't': out.push_back('\t'); break;
                case 'u': {
                    if (pos + 4 > text.size()) {
                        fail("short unicode escape");
                    }
                    unsigned const cp =
                        (unsigned) std::strtoul(text.substr(pos, 4).c_str(), nullptr, 16);
                    pos += 4;
                    append_utf8(out, cp);
                    break;
                }
                default: fail("bad escape");
            }
        }
        return out;
    }

    json_value parse_value() {
        json_value v;
        char const c = peek();
        if (c == '{') {
            ++pos;
            v.type = json_value::kind::object;
            if (peek() == '}') {
                ++pos;
                return v;
            }
            while (true) {
                std::string key = parse_string();
                expect(':');
                v.object[std::move(key)] = parse_value();
                char const d = peek();
                ++pos;
                if (d == '}') {
                    break;
                }
                if (d != ',') {
                    fail("expected ',' or '}'");
                }
            }
        } else if (c == '[') {
            ++pos;
            v.type = json_value::kind::array;
            if (peek() == ']') {
                ++pos;
                return v;
            }
            while (true) {
                v.array.push_back(parse_value());
                char const d = peek();
                ++pos;
                if (d == ']') {
                    break;
                }
                if (d != ',') {
                    fail("expected ',' or ']'");
                }
            }
        } else if (c == '"') {
            v.type = json_value::kind::string;
            v.string = parse_string();
        } else if (text.compare(pos, 4, "true") == 0) {
            v.type = json_value::kind::boolean;
            v.boolean = true;
            pos += 4;
        } else if (text.compare(pos, 5, "false") == 0) {
            v.type = json_value::kind::boolean;
            v.boolean = false;
            pos += 5;
        } else if (text.compare(pos, 4, "null") == 0) {
            pos += 4;
        } else {
            char const* begin = text.c_str() + pos;
            char* end = nullptr;
            double const d = std::strtod(begin, &end);
            if (end == begin) {
                fail("unexpected token");
            }
            v.type = json_value::kind::number;
            v.number = d;
            pos += (size_t) (end - begin);
        }
        return v;
    }
};

}  // namespace

json_value parse_json(std::string const& text) {
    parser p{text};
    json_value v = p.parse_value();
    p.skip_space();
    if (p.pos != text.size()) {
        p.fail("trailing characters");
    }
    return v;
}

}  // namespace rt
