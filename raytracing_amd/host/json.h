// Minimal JSON reader for scene files.
//
// The reference reads scenes with nlohmann::json (un-vendored submodule) and
// iterates objects with items() (src/asset/scene.cpp:109,188,209,259).  That
// container is a std::map, so members come back sorted by key (byte-wise), not
// in file order.  Instance, material and light numbering depend on it, so
// objects here are std::map as well.
#pragma once

#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace rt {

struct json_value {
    enum class kind { null, boolean, number, string, array, object };
    kind type = kind::null;
    bool boolean = false;
    double number = 0.0;
    std::string string;
    std::vector<json_value> array;
    std::map<std::string, json_value> object;

    bool is_object() const { return type == kind::object; }
    bool is_array() const { return type == kind::array; }

    // Member lookup; nullptr when absent or when *this is not an object.
    json_value const* find(std::string const& key) const {
        if (type != kind::object) {
            return nullptr;
        }
        auto const it = object.find(key);
        return it == object.end() ? nullptr : &it->second;
    }

    // Resolve "a/b/0" style paths (object keys and array indices).
    json_value const* at_path(std::string const& path) const;

    bool contains(std::string const& path) const { return at_path(path) != nullptr; }

    // Throwing accessors (the reference's .at()): missing keys are errors.
    json_value const& at(std::string const& path) const {
        json_value const* v = at_path(path);
        if (!v) {
            throw std::runtime_error("json: missing key '" + path + "'");
        }
        return *v;
    }
    float as_float() const {
        if (type != kind::number) {
            throw std::runtime_error("json: expected a number");
        }
        return (float) number;
    }
    uint32_t as_uint() const {
        if (type != kind::number) {
            throw std::runtime_error("json: expected a number");
        }
        return (uint32_t) number;
    }
    std::string const& as_string() const {
        if (type != kind::string) {
            throw std::runtime_error("json: expected a string");
        }
        return string;
    }

    // Defaulting accessors (the reference's .value(pointer, default)).
    float value(std::string const& path, float fallback) const {
        json_value const* v = at_path(path);
        return (v && v->type == kind::number) ? (float) v->number : fallback;
    }
    bool value(std::string const& path, bool fallback) const {
        json_value const* v = at_path(path);
        return (v && v->type == kind::boolean) ? v->boolean : fallback;
    }
    std::string value(std::string const& path, char const* fallback) const {
        json_value const* v = at_path(path);
        return (v && v->type == kind::string) ? v->string : std::string{fallback};
    }
};

// Parses a complete JSON document; throws std::runtime_error on malformed input.
json_value parse_json(std::string const& text);

}  // namespace rt
