// Reader for the subset of YAML the reference's scene files use (the reference links yaml-cpp,
// src/scene/scene_loader.h:8): block mappings, block sequences (of mappings or scalars), flow sequences
// and flow mappings that may span lines and nest, plain / single-quoted / double-quoted scalars, `#`
// comments, blank lines, trailing spaces.  Anchors, tags, block scalars and multi-document streams are
// not supported and are reported as errors.
#ifndef BDPT_MINI_YAML_H
#define BDPT_MINI_YAML_H

#include <string>
#include <utility>
#include <vector>

namespace bdpt_host {

struct YamlNode {
    enum Kind { Undefined, Null, Scalar, Sequence, Map };
    Kind kind = Undefined;
    std::string scalar;
    std::vector<YamlNode> items;
    std::vector<std::pair<std::string, YamlNode>> entries;      // in file order

    bool defined() const { return kind != Undefined; }           // yaml-cpp: static_cast<bool>(node)
    bool is_scalar() const { return kind == Scalar; }
    bool is_sequence() const { return kind == Sequence; }
    bool is_map() const { return kind == Map; }
    // node["key"] of yaml-cpp on a const node: the value, or an undefined node
    const YamlNode& operator[](const std::string& key) const;
    size_t size() const { return kind == Sequence ? items.size() : (kind == Map ? entries.size() : 0); }
};

// Conversions with yaml-cpp's rules (yaml-cpp/node/convert.h): the whole scalar must be consumed;
// `.inf` / `.nan` spellings for floating point; integers accept 0x / 0 prefixes.
bool yaml_to_double(const std::string& s, double& out);
bool yaml_to_int(const std::string& s, int& out);

// Parses `text`; on failure returns false and sets `error` ("line N: ...").
bool yaml_parse(const std::string& text, YamlNode& root, std::string& error);

}  // namespace bdpt_host
#endif
