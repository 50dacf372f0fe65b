// dsc.hpp — scene-description DOM: the subset of SDLang (sdlang-d 0.10.6) and
// JSON (std.json) the reference's loader consumes, behind the SceneDscNode
// interface of rt/scene_loader.d:214-241.
#pragma once

#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace c2rt {
namespace host {

// rt/exception.d: InvalidSceneException & friends -> one exception type with
// the C2RT status it maps to.
struct SceneError : std::runtime_error {
    int status;
    SceneError(int status_, const std::string &msg) : std::runtime_error(msg), status(status_) {}
};

struct DscValue {
    enum Kind { Null, Bool, Int, Float, String } kind = Null;
    bool b = false;
    long long i = 0;
    double f = 0;
    std::string s;
};

// One SDLang tag: `name value... attr=value... { children }`.
struct SdlTag {
    std::string name;
    std::vector<DscValue> values;
    std::vector<std::pair<std::string, DscValue>> attributes;
    std::vector<SdlTag> tags;
    int line = 0;
};

struct JsonValue {
    enum Type { Null, True, False, Integer, Float, String, Array, Object } type = Null;
    long long integer = 0;
    double floating = 0;
    std::string str;
    std::vector<JsonValue> array;
    std::vector<std::pair<std::string, JsonValue>> object; // insertion order
    const JsonValue *find(const std::string &key) const;
};

// sdlang.parseSource: returns the root's child tags (rt/scene_loader.d:55 uses tags[0]).
std::vector<SdlTag> parseSdlSource(const std::string &text);
JsonValue parseJson(const std::string &text);

// rt/scene_loader.d:214-241
class SceneDscNode {
public:
    virtual ~SceneDscNode() = default;
    virtual std::string getType() const = 0;
    virtual bool getName(std::string &out) const = 0;
    virtual bool isSpecified(const std::string &propertyName) const = 0;
    virtual std::unique_ptr<SceneDscNode> getChild(const std::string &propertyName) const = 0;
    virtual std::vector<std::unique_ptr<SceneDscNode>> getChildren() const = 0;
    virtual std::vector<DscValue> getValues() const = 0;
    virtual bool getBool() const = 0;
    virtual long long getInt() const = 0;
    virtual double getFloat() const = 0;
    virtual std::string getString() const = 0;
};

std::unique_ptr<SceneDscNode> makeVal(const SdlTag *tag);     // SdlValueWrapper, :342-403
std::unique_ptr<SceneDscNode> makeVal(const JsonValue *json); // JsonValueWrapper, :243-340

} // namespace host
} // namespace c2rt
