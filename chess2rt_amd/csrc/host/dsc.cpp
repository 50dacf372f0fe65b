// dsc.cpp — SDLang / JSON readers for the scene loader (see dsc.hpp).
//
// SDLang subset (what sdlang-d accepts and the shipped scenes use,
// SURVEY.md 8(f) rank 1): `//`, `#`, `--` line comments and `/* */` block
// comments; tags `name values attributes { children }` terminated by newline
// or `;`; anonymous tags (name "content"); values: "string" with escapes,
// `raw string`, integers (optional L suffix), reals (optional f/d/BD suffix),
// true/false/on/off, null; `\` line continuation; `ns:name` identifiers.
#include "dsc.hpp"

#include <cctype>
#include <cmath>
#include <cstdlib>

#include "../../../include/c2rt.h"

namespace c2rt {
namespace host {

namespace {

[[noreturn]] void parse_fail(const std::string &what, int line)
{
    throw SceneError(C2RT_ERR_PARSE, what + " (line " + std::to_string(line) + ")");
}

// ------------------------------------------------------------------ SDLang
struct SdlParser {
    const std::string &t;
    size_t i = 0;
    int line = 1;
    explicit SdlParser(const std::string &text) : t(text) {}

    bool eof() const { return i >= t.size(); }
    char cur() const { return eof() ? '\0' : t[i]; }
    char peek(size_t k = 1) const { return i + k < t.size() ? t[i + k] : '\0'; }
    void adv() { if (cur() == '\n') ++line; ++i; }

    static bool ident_start(char c) { return std::isalpha((unsigned char)c) || c == '_'; }
    static bool ident_char(char c) { return std::isalnum((unsigned char)c) || c == '_' || c == '-' || c == '.' || c == '$'; }

    // skips blanks and comments; newlines too when `newlines`
    void skip(bool newlines)
    {
        for (;;) {
            const char c = cur();
            if (c == ' ' || c == '\t' || c == '\r') { adv(); continue; }
            if (c == '\n' && newlines) { adv(); continue; }
            if (c == '\\' && (peek() == '\n' || (peek() == '\r' && peek(2) == '\n'))) { // line continuation
                adv();
                while (cur() != '\n') adv();
                adv();
                continue;
            }
            if ((c == '/' && peek() == '/') || c == '#' || (c == '-' && peek() == '-')) {
                while (!eof() && cur() != '\n') adv();
                continue;
            }
            if (c == '/' && peek() == '*') {
                adv(); adv();
                while (!eof() && !(cur() == '*' && peek() == '/')) adv();
                if (eof()) parse_fail("unterminated /* comment", line);
                adv(); adv();
                continue;
            }
            break;
        }
    }

    std::string ident()
    {
        std::string s;
        while (ident_char(cur()) || cur() == ':') { s += cur(); adv(); }
        return s;
    }

    bool at_value() const
    {
        const char c = cur();
        return c == '"' || c == '`' || c == '\'' || std::isdigit((unsigned char)c) ||
               ((c == '-' || c == '+' || c == '.') && (std::isdigit((unsigned char)peek()) || peek() == '.'));
    }

    DscValue string_value()
    {
        DscValue v;
        v.kind = DscValue::String;
        const char q = cur();
        adv();
        if (q == '`') {
            while (!eof() && cur() != '`') { v.s += cur(); adv(); }
        } else {
            while (!eof() && cur() != q) {
                if (cur() == '\\') {
                    adv();
                    switch (cur()) {
                    case 'n': v.s += '\n'; break;
                    case 't': v.s += '\t'; break;
                    case 'r': v.s += '\r'; break;
                    case '0': v.s += '\0'; break;
                    case '\n': { // escaped newline inside a string: skip leading blanks of the next line
                        adv();
                        while (cur() == ' ' || cur() == '\t') adv();
                        continue;
                    }
                    default: v.s += cur(); break;
                    }
                    adv();
                } else {
                    if (cur() == '\n' && q == '"') parse_fail("newline in string", line);
                    v.s += cur();
                    adv();
                }
            }
        }
        if (eof()) parse_fail("unterminated string", line);
        adv();
        return v;
    }

    DscValue number_value()
    {
        const size_t start = i;
        if (cur() == '-' || cur() == '+') adv();
        bool real = false;
        while (std::isdigit((unsigned char)cur())) adv();
        if (cur() == '.' && std::isdigit((unsigned char)peek())) {
            real = true;
            adv();
            while (std::isdigit((unsigned char)cur())) adv();
        }
        if ((cur() == 'e' || cur() == 'E') &&
            (std::isdigit((unsigned char)peek()) || ((peek() == '-' || peek() == '+') && std::isdigit((unsigned char)peek(2))))) {
            real = true;
            adv();
            if (cur() == '-' || cur() == '+') adv();
            while (std::isdigit((unsigned char)cur())) adv();
        }
        const std::string text = t.substr(start, i - start);
        DscValue v;
        // suffixes: L (long), f/F (float), d/D (double), BD (decimal)
        if (cur() == 'L' || cur() == 'l') { adv(); }
        else if (cur() == 'f' || cur() == 'F') { real = true; adv(); v.kind = DscValue::Float; v.f = (double)std::strtof(text.c_str(), nullptr); return v; }
        else if (cur() == 'd' || cur() == 'D') { real = true; adv(); }
        else if ((cur() == 'B' || cur() == 'b') && (peek() == 'D' || peek() == 'd')) { real = true; adv(); adv(); }
        if (ident_char(cur())) parse_fail("malformed number '" + text + "'", line);
        if (real) {
            v.kind = DscValue::Float;
            v.f = std::strtod(text.c_str(), nullptr);
        } else {
            v.kind = DscValue::Int;
            v.i = std::strtoll(text.c_str(), nullptr, 10);
        }
        return v;
    }

    bool keyword_value(const std::string &w, DscValue &v)
    {
        if (w == "true" || w == "on") { v.kind = DscValue::Bool; v.b = true; return true; }
        if (w == "false" || w == "off") { v.kind = DscValue::Bool; v.b = false; return true; }
        if (w == "null") { v.kind = DscValue::Null; return true; }
        return false;
    }

    DscValue value()
    {
        if (cur() == '"' || cur() == '`') return string_value();
        if (cur() == '\'') { DscValue v = string_value(); return v; }
        return number_value();
    }

    // parses tags until `}` (when nested) or EOF
    void tags(std::vector<SdlTag> &out, bool nested)
    {
        for (;;) {
            skip(true);
            while (cur() == ';') { adv(); skip(true); }
            if (eof()) {
                if (nested) parse_fail("missing '}'", line);
                return;
            }
            if (cur() == '}') {
                if (!nested) parse_fail("unexpected '}'", line);
                adv();
                return;
            }
            SdlTag tag;
            tag.line = line;
            if (ident_start(cur())) {
                const size_t save = i;
                const int save_line = line;
                const std::string w = ident();
                DscValue kv;
                if (keyword_value(w, kv) && cur() != '=') { // anonymous tag starting with a keyword value
                    tag.name = "content";
                    tag.values.push_back(kv);
                } else {
                    (void)save; (void)save_line;
                    tag.name = w;
                }
            } else if (at_value()) {
                tag.name = "content";
            } else if (cur() == '{') {
                tag.name = "content";
            } else {
                parse_fail(std::string("unexpected character '") + cur() + "'", line);
            }
            // values and attributes
            for (;;) {
                skip(false);
                if (at_value()) {
                    if (!tag.attributes.empty()) parse_fail("value after attribute", line);
                    tag.values.push_back(value());
                } else if (ident_start(cur())) {
                    const std::string w = ident();
                    if (cur() == '=') {
                        adv();
                        DscValue av;
                        if (at_value()) av = value();
                        else {
                            const std::string kw = ident();
                            if (!keyword_value(kw, av)) parse_fail("bad attribute value", line);
                        }
                        tag.attributes.emplace_back(w, av);
                    } else {
                        DscValue kv;
                        if (!keyword_value(w, kv)) parse_fail("unexpected identifier '" + w + "'", line);
                        tag.values.push_back(kv);
                    }
                } else
                    break;
            }
            if (cur() == '{') {
                adv();
                tags(tag.tags, true);
                skip(false);
            }
            if (!(eof() || cur() == '\n' || cur() == ';' || cur() == '}'))
                parse_fail(std::string("unexpected character '") + cur() + "' after tag '" + tag.name + "'", line);
            out.push_back(std::move(tag));
        }
    }
};

// ------------------------------------------------------------------ JSON
struct JsonParser {
    const std::string &t;
    size_t i = 0;
    int line = 1;
    explicit JsonParser(const std::string &text) : t(text) {}
    char cur() const { return i < t.size() ? t[i] : '\0'; }
    void adv() { if (cur() == '\n') ++line; ++i; }
    void ws() { while (cur() == ' ' || cur() == '\t' || cur() == '\n' || cur() == '\r') adv(); }
    void expect(char c)
    {
        if (cur() != c) parse_fail(std::string("JSON: expected '") + c + "'", line);
        adv();
    }
    std::string string()
    {
        expect('"');
        std::string s;
        while (i < t.size() && cur() != '"') {
            if (cur() == '\\') {
                adv();
                switch (cur()) {
                case 'n': s += '\n'; break;
                case 't': s += '\t'; break;
                case 'r': s += '\r'; break;
                case 'b': s += '\b'; break;
                case 'f': s += '\f'; break;
                case 'u': {
                    unsigned cp = 0;
                    for (int k = 0; k < 4; ++k) { adv(); cp = cp * 16 + (unsigned)std::strtol(std::string(1, cur()).c_str(), nullptr, 16); }
                    if (cp < 0x80) s += (char)cp;
                    else if (cp < 0x800) { s += (char)(0xC0 | (cp >> 6)); s += (char)(0x80 | (cp & 0x3F)); }
                    else { s += (char)(0xE0 | (cp >> 12)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
                    break;
                }
                default: s += cur(); break;
                }
                adv();
            } else {
                s += cur();
                adv();
            }
        }
        expect('"');
        return s;
    }
    JsonValue value()
    {
        ws();
        JsonValue v;
        const char c = cur();
        if (c == '{') {
            v.type = JsonValue::Object;
            adv();
            ws();
            if (cur() == '}') { adv(); return v; }
            for (;;) {
                ws();
                std::string k = string();
                ws();
                expect(':');
                JsonValue m = value();
                v.object.emplace_back(std::move(k), std::move(m));
                ws();
                if (cur() == ',') { adv(); continue; }
                expect('}');
                break;
            }
        } else if (c == '[') {
            v.type = JsonValue::Array;
            adv();
            ws();
            if (cur() == ']') { adv(); return v; }
            for (;;) {
                v.array.push_back(value());
                ws();
                if (cur() == ',') { adv(); continue; }
                expect(']');
                break;
            }
        } else if (c == '"') {
            v.type = JsonValue::String;
            v.str = string();
        } else if (t.compare(i, 4, "true") == 0) { v.type = JsonValue::True; i += 4; }
        else if (t.compare(i, 5, "false") == 0) { v.type = JsonValue::False; i += 5; }
        else if (t.compare(i, 4, "null") == 0) { v.type = JsonValue::Null; i += 4; }
        else if (c == '-' || std::isdigit((unsigned char)c)) {
            const size_t start = i;
            bool real = false;
            if (cur() == '-') adv();
            while (std::isdigit((unsigned char)cur())) adv();
            if (cur() == '.') { real = true; adv(); while (std::isdigit((unsigned char)cur())) adv(); }
            if (cur() == 'e' || cur() == 'E') {
                real = true;
                adv();
                if (cur() == '-' || cur() == '+') adv();
                while (std::isdigit((unsigned char)cur())) adv();
            }
            const std::string text = t.substr(start, i - start);
            if (real) { v.type = JsonValue::Float; v.floating = std::strtod(text.c_str(), nullptr); }
            else { v.type = JsonValue::Integer; v.integer = std::strtoll(text.c_str(), nullptr, 10); }
        } else
            parse_fail("JSON: unexpected character", line);
        return v;
    }
};

// --------------------------------------------------- SdlValueWrapper :342-403
class SdlValueWrapper final : public SceneDscNode {
    const SdlTag *tag;
    const DscValue &first() const
    {
        if (tag->values.empty()) throw SceneError(C2RT_ERR_PARSE, "tag '" + tag->name + "' has no value (line " + std::to_string(tag->line) + ")");
        return tag->values[0];
    }
public:
    explicit SdlValueWrapper(const SdlTag *t) : tag(t) {}
    std::string getType() const override { return tag->name; }
    bool getName(std::string &out) const override
    {
        if (!tag->values.empty() && tag->values[0].kind == DscValue::String) { out = tag->values[0].s; return true; }
        if (isSpecified("name")) { out = getChild("name")->getString(); return true; }
        return false;
    }
    bool isSpecified(const std::string &p) const override
    {
        for (const SdlTag &c : tag->tags) if (c.name == p) return true;
        return false;
    }
    std::unique_ptr<SceneDscNode> getChild(const std::string &p) const override
    {
        for (const SdlTag &c : tag->tags) if (c.name == p) return makeVal(&c);
        throw SceneError(C2RT_ERR_PARSE, "missing tag '" + p + "'");
    }
    std::vector<std::unique_ptr<SceneDscNode>> getChildren() const override
    {
        std::vector<std::unique_ptr<SceneDscNode>> r;
        for (const SdlTag &c : tag->tags) r.push_back(makeVal(&c));
        return r;
    }
    std::vector<DscValue> getValues() const override { return tag->values; }
    // Variant.get!T: exact type or an implicit conversion (int -> long -> double)
    bool getBool() const override
    {
        if (first().kind != DscValue::Bool) throw SceneError(C2RT_ERR_PARSE, "tag '" + tag->name + "': expected a boolean");
        return first().b;
    }
    long long getInt() const override
    {
        if (first().kind != DscValue::Int) throw SceneError(C2RT_ERR_PARSE, "tag '" + tag->name + "': expected an integer");
        return first().i;
    }
    double getFloat() const override
    {
        if (first().kind == DscValue::Int) return (double)first().i;
        if (first().kind != DscValue::Float) throw SceneError(C2RT_ERR_PARSE, "tag '" + tag->name + "': expected a number");
        return first().f;
    }
    std::string getString() const override
    {
        if (first().kind != DscValue::String) throw SceneError(C2RT_ERR_PARSE, "tag '" + tag->name + "': expected a string");
        return first().s;
    }
};

// -------------------------------------------------- JsonValueWrapper :243-340
class JsonValueWrapper final : public SceneDscNode {
    const JsonValue *json;
    static double number(const JsonValue &j)
    {
        if (j.type == JsonValue::Float) return j.floating;
        if (j.type == JsonValue::Integer) return (double)j.integer;
        throw SceneError(C2RT_ERR_PARSE, "JSON: expected a number");
    }
public:
    explicit JsonValueWrapper(const JsonValue *j) : json(j) {}
    std::string getType() const override
    {
        const JsonValue *t = json->find("type");
        if (!t || t->type != JsonValue::String) throw SceneError(C2RT_ERR_PARSE, "JSON: object without \"type\"");
        return t->str;
    }
    bool getName(std::string &out) const override
    {
        const JsonValue *n = json->find("name");
        if (!n) return false;
        if (n->type != JsonValue::String) throw SceneError(C2RT_ERR_PARSE, "JSON: \"name\" must be a string");
        out = n->str;
        return true;
    }
    bool isSpecified(const std::string &p) const override
    {
        if (json->type != JsonValue::Object) throw SceneError(C2RT_ERR_PARSE, "JSON: expected an object");
        return json->find(p) != nullptr;
    }
    std::unique_ptr<SceneDscNode> getChild(const std::string &p) const override
    {
        const JsonValue *c = json->find(p);
        if (!c) throw SceneError(C2RT_ERR_PARSE, "JSON: missing key '" + p + "'");
        return makeVal(c);
    }
    std::vector<std::unique_ptr<SceneDscNode>> getChildren() const override
    {
        if (json->type != JsonValue::Array) throw SceneError(C2RT_ERR_PARSE, "JSON: expected an array");
        std::vector<std::unique_ptr<SceneDscNode>> r;
        for (const JsonValue &c : json->array) r.push_back(makeVal(&c));
        return r;
    }
    std::vector<DscValue> getValues() const override
    {
        if (json->type != JsonValue::Array) throw SceneError(C2RT_ERR_PARSE, "JSON: expected an array");
        std::vector<DscValue> r;
        for (const JsonValue &c : json->array) {
            DscValue v;
            v.kind = DscValue::Float;
            v.f = number(c);
            r.push_back(v);
        }
        return r;
    }
    bool getBool() const override
    {
        if (json->type == JsonValue::True) return true;
        if (json->type == JsonValue::False) return false;
        throw SceneError(C2RT_ERR_PARSE, "JSON: expected a boolean");
    }
    long long getInt() const override { return (long long)number(*json); }
    double getFloat() const override { return number(*json); }
    std::string getString() const override
    {
        if (json->type != JsonValue::String) throw SceneError(C2RT_ERR_PARSE, "JSON: expected a string");
        return json->str;
    }
};

} // namespace

const JsonValue *JsonValue::find(const std::string &key) const
{
    for (const auto &kv : object) if (kv.first == key) return &kv.second;
    return nullptr;
}

std::vector<SdlTag> parseSdlSource(const std::string &text)
{
    SdlParser p(text);
    // UTF-8 BOM
    if (text.size() >= 3 && (unsigned char)text[0] == 0xEF && (unsigned char)text[1] == 0xBB && (unsigned char)text[2] == 0xBF) p.i = 3;
    std::vector<SdlTag> root;
    p.tags(root, false);
    return root;
}

JsonValue parseJson(const std::string &text)
{
    JsonParser p(text);
    JsonValue v = p.value();
    p.ws();
    if (p.i < text.size()) parse_fail("JSON: trailing characters", p.line);
    return v;
}

std::unique_ptr<SceneDscNode> makeVal(const SdlTag *tag) { return std::unique_ptr<SceneDscNode>(new SdlValueWrapper(tag)); }
std::unique_ptr<SceneDscNode> makeVal(const JsonValue *json) { return std::unique_ptr<SceneDscNode>(new JsonValueWrapper(json)); }

} // namespace host
} // namespace c2rt
