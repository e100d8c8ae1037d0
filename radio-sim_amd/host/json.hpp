// json.hpp -- the small JSON layer of the native radio-link server (rsim_server.cpp), with the number and
// string text the reference's wire carries.  The reference uses minimal-json 0.9.2 (radio-medium/lib, class
// files only); what matters on the wire and is reproduced here:
//   * an object keeps its members in insertion order, set() replaces in place, the writer emits no blanks;
//   * a parsed number keeps its literal text: JsonValue.toString() of the node id `1` is "1", of `1.0` is
//     "1.0", of a string "n1" is "\"n1\"" WITH the quotes -- that text is the key of Simulator's node table
//     (net/SimulatorJSONHandler.java:70,98,109);
//   * asLong / asInt parse the literal as a decimal integer (Long.parseLong: "1.0" or "1e3" throw), asDouble as
//     a double; an accessor of the wrong type throws -- in the reference that ends the connection's reader thread;
//   * a double is written as Java's Double.toString with a trailing ".0" cut off (JsonValue.valueOf(double),
//     cutOffPointZero): -100.0 -> -100, -99.99 -> -99.99, 1.0E10 stays 1.0E10; NaN / infinity are refused.
#pragma once

#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace rsim {

struct JsonError : std::runtime_error {
    explicit JsonError(const std::string &m) : std::runtime_error(m) {}
};

// java.lang.Double.toString: shortest digits that identify the double; decimal notation for 1e-3 <= |d| < 1e7,
// else "d.dddE[-]n"; always at least one digit after the point.  Written into `dst` (at least 32 bytes), no allocation:
// the server formats a double per delivery and per changed node -- tens of thousands per step.  Finite d only.
inline size_t java_double_chars(double d, char *dst)
{
    char *o = dst;
    if (d == 0.0) {
        if (std::signbit(d)) *o++ = '-';
        *o++ = '0', *o++ = '.', *o++ = '0';
        return size_t(o - dst);
    }
    const double a = std::fabs(d);
    char sci[40]; // d[.ddd]e[+-]XX, shortest round-trip digits
    const auto r = std::to_chars(sci, sci + sizeof(sci), a, std::chars_format::scientific);
    char digits[24];
    int nd = 0;
    const char *p = sci;
    for (; p < r.ptr && *p != 'e'; ++p)
        if (*p != '.') digits[nd++] = *p;
    int exp10 = 0;
    std::from_chars(p + (p + 1 < r.ptr && p[1] == '+' ? 2 : 1), r.ptr, exp10);
    if (nd == 1) { // Java always writes two digits, the pair closest to the exact value (4.9E-324, not 5.0E-324)
        char two[40]; // d.de[+-]XX
        const auto r2 = std::to_chars(two, two + sizeof(two), a, std::chars_format::scientific, 1);
        digits[0] = two[0];
        nd = 1;
        if (two[2] != '0') digits[nd++] = two[2];
        exp10 = 0;
        std::from_chars(two + (two[4] == '+' ? 5 : 4), r2.ptr, exp10);
    }
    if (std::signbit(d)) *o++ = '-';
    if (a >= 1e-3 && a < 1e7) {
        if (exp10 >= 0) {
            for (int i = 0; i <= exp10; ++i) *o++ = i < nd ? digits[i] : '0';
            *o++ = '.';
            if (nd > exp10 + 1)
                for (int i = exp10 + 1; i < nd; ++i) *o++ = digits[i];
            else
                *o++ = '0';
        } else {
            *o++ = '0', *o++ = '.';
            for (int i = 0; i < -exp10 - 1; ++i) *o++ = '0';
            for (int i = 0; i < nd; ++i) *o++ = digits[i];
        }
    } else {
        *o++ = digits[0];
        *o++ = '.';
        if (nd > 1)
            for (int i = 1; i < nd; ++i) *o++ = digits[i];
        else
            *o++ = '0';
        *o++ = 'E';
        o = std::to_chars(o, o + 8, exp10).ptr;
    }
    return size_t(o - dst);
}
inline std::string java_double_to_string(double d)
{
    if (std::isnan(d)) return "NaN";
    if (std::isinf(d)) return d > 0 ? "Infinity" : "-Infinity";
    char buf[48];
    return std::string(buf, java_double_chars(d, buf));
}

// JsonValue.valueOf(double): cutOffPointZero(Double.toString(value))
inline size_t json_double_chars(double d, char *dst) // (finite d; dst: at least 32 bytes)
{
    size_t n = java_double_chars(d, dst);
    if (n > 2 && dst[n - 2] == '.' && dst[n - 1] == '0') n -= 2;
    return n;
}
inline std::string json_double_text(double d)
{
    if (std::isnan(d) || std::isinf(d)) throw JsonError("Infinite and NaN values not permitted in JSON");
    char buf[48];
    return std::string(buf, json_double_chars(d, buf));
}

// writers for messages that are formatted directly (no object tree): the same text as Json::of(v).toString()
inline void append_int(std::string &out, int64_t v)
{
    char buf[24];
    const auto r = std::to_chars(buf, buf + sizeof(buf), v);
    out.append(buf, r.ptr);
}
inline void append_double(std::string &out, double d)
{
    // whole numbers below 1e7 in magnitude -- every idle node's -100 -- are "<integer>.0" in Java, the ".0" cut off
    if (d == std::floor(d) && std::fabs(d) < 1e7 && !(d == 0.0 && std::signbit(d))) {
        append_int(out, int64_t(d));
        return;
    }
    if (std::isnan(d) || std::isinf(d)) throw JsonError("Infinite and NaN values not permitted in JSON");
    char buf[48];
    out.append(buf, json_double_chars(d, buf));
}

class Json {
public:
    enum Type { NUL, BOOL, NUMBER, STRING, ARRAY, OBJECT };
    Json() : type_(NUL) {}
    static Json null() { return Json(); }
    static Json boolean(bool b) { Json j; j.type_ = BOOL; j.text_ = b ? "true" : "false"; return j; }
    static Json number_text(const std::string &literal) { Json j; j.type_ = NUMBER; j.text_ = literal; return j; }
    static Json of(int64_t v) { return number_text(std::to_string(v)); }
    static Json of(int v) { return number_text(std::to_string(v)); }
    static Json of(double v) { return number_text(json_double_text(v)); }
    static Json of(const std::string &s) { Json j; j.type_ = STRING; j.text_ = s; return j; }
    static Json of(const char *s) { return of(std::string(s)); }
    static Json array() { Json j; j.type_ = ARRAY; return j; }
    static Json object() { Json j; j.type_ = OBJECT; return j; }

    Type type() const { return type_; }
    bool isNumber() const { return type_ == NUMBER; }
    bool isString() const { return type_ == STRING; }
    bool isArray() const { return type_ == ARRAY; }
    bool isObject() const { return type_ == OBJECT; }

    // typed accessors; the wrong type throws, as minimal-json's UnsupportedOperationException
    const std::string &asString() const { need(STRING, "string"); return text_; }
    int64_t asLong() const
    {
        need(NUMBER, "number");
        return parse_integer(text_, "long"); // Long.parseLong(string, 10)
    }
    int asInt() const
    {
        need(NUMBER, "number");
        const int64_t v = parse_integer(text_, "int");
        if (v < INT32_MIN || v > INT32_MAX) throw JsonError("For input string: \"" + text_ + "\"");
        return int(v);
    }
    double asDouble() const { need(NUMBER, "number"); return std::strtod(text_.c_str(), nullptr); }
    const Json &asObject() const { need(OBJECT, "object"); return *this; }
    const Json &asArray() const { need(ARRAY, "array"); return *this; }

    // object
    const Json *get(const std::string &name) const
    {
        need(OBJECT, "object");
        for (const auto &m : members_)
            if (m.first == name) return &m.second;
        return nullptr;
    }
    const Json &at(const std::string &name) const // json.get(name).xxx(): a missing member is a NullPointerException there
    {
        const Json *v = get(name);
        if (!v) throw JsonError("missing member \"" + name + "\"");
        return *v;
    }
    std::string getString(const std::string &name, const std::string &def) const
    {
        const Json *v = get(name);
        return v ? v->asString() : def;
    }
    bool hasString(const std::string &name) const { return get(name) != nullptr; }
    int64_t getLong(const std::string &name, int64_t def) const
    {
        const Json *v = get(name);
        return v ? v->asLong() : def;
    }
    Json &add(const std::string &name, Json v) // appends, as JsonObject.add
    {
        need(OBJECT, "object");
        members_.emplace_back(name, std::move(v));
        return *this;
    }
    Json &set(const std::string &name, Json v) // replaces the member in place or appends, as JsonObject.set
    {
        need(OBJECT, "object");
        for (auto &m : members_)
            if (m.first == name) {
                m.second = std::move(v);
                return *this;
            }
        members_.emplace_back(name, std::move(v));
        return *this;
    }
    // array
    Json &push(Json v) { need(ARRAY, "array"); items_.push_back(std::move(v)); return *this; }
    size_t size() const { return type_ == ARRAY ? items_.size() : members_.size(); }
    const Json &operator[](size_t i) const { need(ARRAY, "array"); return items_.at(i); }

    // JsonValue.toString(): minimal text
    std::string toString() const
    {
        std::string out;
        write(out);
        return out;
    }

    // the text of a JSON string / of a whole value appended to `out` (for writers that format a message directly)
    static void quote(const std::string &s, std::string &out) { write_string(s, out); }
    void append_to(std::string &out) const { write(out); }

    static Json parse(const std::string &text)
    {
        size_t p = 0;
        skip_ws(text, p);
        Json v = parse_value(text, p, 0);
        skip_ws(text, p);
        if (p != text.size()) throw JsonError("Unexpected character at " + std::to_string(p));
        return v;
    }
    // JsonObject.readFrom
    static Json parse_object(const std::string &text)
    {
        Json v = parse(text);
        if (!v.isObject()) throw JsonError("Not an object");
        return v;
    }

private:
    Type type_;
    std::string text_; // number literal, string value, "true" / "false"
    std::vector<std::pair<std::string, Json>> members_;
    std::vector<Json> items_;

    void need(Type t, const char *what) const
    {
        if (type_ != t) throw JsonError(std::string("Not a") + (what[0] == 'o' || what[0] == 'a' ? "n " : " ") + what + ": " + brief());
    }
    std::string brief() const { return type_ == OBJECT || type_ == ARRAY ? (type_ == OBJECT ? "{...}" : "[...]") : toString(); }
    static int64_t parse_integer(const std::string &s, const char *)
    {
        size_t i = 0;
        if (i < s.size() && (s[i] == '-' || s[i] == '+')) ++i;
        if (i == s.size()) throw JsonError("For input string: \"" + s + "\"");
        for (size_t k = i; k < s.size(); ++k)
            if (s[k] < '0' || s[k] > '9') throw JsonError("For input string: \"" + s + "\"");
        errno = 0;
        const long long v = std::strtoll(s.c_str(), nullptr, 10);
        if (errno == ERANGE) throw JsonError("For input string: \"" + s + "\"");
        return int64_t(v);
    }
    static void write_string(const std::string &s, std::string &out)
    {
        out += '"';
        for (unsigned char c : s) {
            switch (c) {
            case '"': out += "\\\""; break;
            case '\\': out += "\\\\"; break;
            case '\n': out += "\\n"; break;
            case '\r': out += "\\r"; break;
            case '\t': out += "\\t"; break;
            default:
                if (c < 0x20) {
                    static const char *hex = "0123456789abcdef";
                    out += "\\u00";
                    out += hex[c >> 4];
                    out += hex[c & 15];
                } else {
                    out += char(c);
                }
            }
        }
        out += '"';
    }
    void write(std::string &out) const
    {
        switch (type_) {
        case NUL: out += "null"; break;
        case BOOL:
        case NUMBER: out += text_; break;
        case STRING: write_string(text_, out); break;
        case ARRAY:
            out += '[';
            for (size_t i = 0; i < items_.size(); ++i) {
                if (i) out += ',';
                items_[i].write(out);
            }
            out += ']';
            break;
        case OBJECT:
            out += '{';
            for (size_t i = 0; i < members_.size(); ++i) {
                if (i) out += ',';
                write_string(members_[i].first, out);
                out += ':';
                members_[i].second.write(out);
            }
            out += '}';
            break;
        }
    }
    static void skip_ws(const std::string &t, size_t &p)
    {
        while (p < t.size() && (t[p] == ' ' || t[p] == '\t' || t[p] == '\n' || t[p] == '\r')) ++p;
    }
    static void append_utf8(std::string &out, unsigned cp)
    {
        if (cp < 0x80) out += char(cp);
        else if (cp < 0x800) { out += char(0xC0 | (cp >> 6)); out += char(0x80 | (cp & 0x3F)); }
        else { out += char(0xE0 | (cp >> 12)); out += char(0x80 | ((cp >> 6) & 0x3F)); out += char(0x80 | (cp & 0x3F)); }
    }
    static std::string parse_string(const std::string &t, size_t &p)
    {
        std::string out;
        ++p; // opening quote
        for (;;) {
            if (p >= t.size()) throw JsonError("Unexpected end of input");
            const char c = t[p++];
            if (c == '"') return out;
            if (static_cast<unsigned char>(c) < 0x20) throw JsonError("Expected valid string character at " + std::to_string(p - 1));
            if (c != '\\') { out += c; continue; }
            if (p >= t.size()) throw JsonError("Unexpected end of input");
            const char e = t[p++];
            switch (e) {
            case '"': case '/': case '\\': out += e; break;
            case 'b': out += '\b'; break;
            case 'f': out += '\f'; break;
            case 'n': out += '\n'; break;
            case 'r': out += '\r'; break;
            case 't': out += '\t'; break;
            case 'u': {
                if (p + 4 > t.size()) throw JsonError("Unexpected end of input");
                unsigned cp = 0;
                for (int k = 0; k < 4; ++k) {
                    const char h = t[p++];
                    cp <<= 4;
                    if (h >= '0' && h <= '9') cp |= unsigned(h - '0');
                    else if (h >= 'a' && h <= 'f') cp |= unsigned(h - 'a' + 10);
                    else if (h >= 'A' && h <= 'F') cp |= unsigned(h - 'A' + 10);
                    else throw JsonError("Expected hexadecimal digit at " + std::to_string(p - 1));
                }
                append_utf8(out, cp);
                break;
            }
            default: throw JsonError("Expected valid escape sequence at " + std::to_string(p - 1));
            }
        }
    }
    static Json parse_number(const std::string &t, size_t &p)
    {
        const size_t b = p;
        if (p < t.size() && t[p] == '-') ++p;
        if (p >= t.size() || t[p] < '0' || t[p] > '9') throw JsonError("Expected digit at " + std::to_string(p));
        if (t[p] == '0') ++p;
        else while (p < t.size() && t[p] >= '0' && t[p] <= '9') ++p;
        if (p < t.size() && t[p] == '.') {
            ++p;
            if (p >= t.size() || t[p] < '0' || t[p] > '9') throw JsonError("Expected digit at " + std::to_string(p));
            while (p < t.size() && t[p] >= '0' && t[p] <= '9') ++p;
        }
        if (p < t.size() && (t[p] == 'e' || t[p] == 'E')) {
            ++p;
            if (p < t.size() && (t[p] == '+' || t[p] == '-')) ++p;
            if (p >= t.size() || t[p] < '0' || t[p] > '9') throw JsonError("Expected digit at " + std::to_string(p));
            while (p < t.size() && t[p] >= '0' && t[p] <= '9') ++p;
        }
        return number_text(t.substr(b, p - b));
    }
    static Json parse_value(const std::string &t, size_t &p, int depth)
    {
        if (depth > 200) throw JsonError("Nesting too deep");
        if (p >= t.size()) throw JsonError("Unexpected end of input");
        const char c = t[p];
        if (c == '{') {
            Json o = object();
            ++p;
            skip_ws(t, p);
            if (p < t.size() && t[p] == '}') { ++p; return o; }
            for (;;) {
                skip_ws(t, p);
                if (p >= t.size() || t[p] != '"') throw JsonError("Expected name at " + std::to_string(p));
                std::string name = parse_string(t, p);
                skip_ws(t, p);
                if (p >= t.size() || t[p] != ':') throw JsonError("Expected ':' at " + std::to_string(p));
                ++p;
                skip_ws(t, p);
                o.members_.emplace_back(std::move(name), parse_value(t, p, depth + 1));
                skip_ws(t, p);
                if (p < t.size() && t[p] == ',') { ++p; continue; }
                if (p < t.size() && t[p] == '}') { ++p; return o; }
                throw JsonError("Expected ',' or '}' at " + std::to_string(p));
            }
        }
        if (c == '[') {
            Json a = array();
            ++p;
            skip_ws(t, p);
            if (p < t.size() && t[p] == ']') { ++p; return a; }
            for (;;) {
                skip_ws(t, p);
                a.items_.push_back(parse_value(t, p, depth + 1));
                skip_ws(t, p);
                if (p < t.size() && t[p] == ',') { ++p; continue; }
                if (p < t.size() && t[p] == ']') { ++p; return a; }
                throw JsonError("Expected ',' or ']' at " + std::to_string(p));
            }
        }
        if (c == '"') return of(parse_string(t, p));
        if (c == '-' || (c >= '0' && c <= '9')) return parse_number(t, p);
        if (t.compare(p, 4, "true") == 0) { p += 4; return boolean(true); }
        if (t.compare(p, 5, "false") == 0) { p += 5; return boolean(false); }
        if (t.compare(p, 4, "null") == 0) { p += 4; return null(); }
        throw JsonError("Expected value at " + std::to_string(p));
    }
};

} // namespace rsim
