#include "mini_yaml.h"

#include <cerrno>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <stdexcept>

namespace bdpt_host {

namespace {

const YamlNode kUndefined;

struct ParseError : std::runtime_error {
    explicit ParseError(size_t line, const std::string& what)
        : std::runtime_error("line " + std::to_string(line) + ": " + what) {}
};

struct Line {
    size_t number;      // 1-based, for messages
    int indent;
    std::string text;   // without indentation, comment and trailing blanks
};

bool is_blank(char c) { return c == ' ' || c == '\t'; }

// Removes a trailing comment (a '#' at the start or after a blank, outside quotes) and trailing blanks.
std::string strip_comment(const std::string& s) {
    char quote = 0;
    size_t end = s.size();
    for (size_t i = 0; i < s.size(); i++) {
        const char c = s[i];
        if (quote) {
            if (quote == '"' && c == '\\') { i++; continue; }
            if (c == quote) {
                if (quote == '\'' && i + 1 < s.size() && s[i + 1] == '\'') { i++; continue; }
                quote = 0;
            }
        } else if (c == '"' || c == '\'') {
            // a quote only opens a quoted scalar where a scalar can start
            size_t p = i;
            while (p > 0 && is_blank(s[p - 1])) p--;
            if (p == 0 || s[p - 1] == ':' || s[p - 1] == '-' || s[p - 1] == '[' || s[p - 1] == ',' || s[p - 1] == '{') quote = c;
        } else if (c == '#' && (i == 0 || is_blank(s[i - 1]))) {
            end = i;
            break;
        }
    }
    while (end > 0 && (is_blank(s[end - 1]) || s[end - 1] == '\r')) end--;
    return s.substr(0, end);
}

std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && (is_blank(s[a]) || s[a] == '\n')) a++;            // flow collections are joined with line feeds
    while (b > a && (is_blank(s[b - 1]) || s[b - 1] == '\n')) b--;
    return s.substr(a, b - a);
}

YamlNode make_scalar(const std::string& raw, size_t line) {
    YamlNode n;
    const std::string s = trim(raw);
    if (s.empty() || s == "~" || s == "null" || s == "Null" || s == "NULL") { n.kind = YamlNode::Null; return n; }
    n.kind = YamlNode::Scalar;
    if (s[0] == '"') {
        if (s.size() < 2 || s.back() != '"') throw ParseError(line, "unterminated double-quoted scalar");
        for (size_t i = 1; i + 1 < s.size(); i++) {
            char c = s[i];
            if (c == '\\' && i + 2 < s.size()) {
                const char e = s[++i];
                switch (e) {
                    case 'n': c = '\n'; break;
                    case 't': c = '\t'; break;
                    case 'r': c = '\r'; break;
                    case '0': c = '\0'; break;
                    case '"': c = '"'; break;
                    case '\\': c = '\\'; break;
                    case '/': c = '/'; break;
                    default: throw ParseError(line, std::string("unsupported escape \\") + e);
                }
            }
            n.scalar.push_back(c);
        }
    } else if (s[0] == '\'') {
        if (s.size() < 2 || s.back() != '\'') throw ParseError(line, "unterminated single-quoted scalar");
        for (size_t i = 1; i + 1 < s.size(); i++) {
            if (s[i] == '\'' && i + 2 < s.size() && s[i + 1] == '\'') i++;
            n.scalar.push_back(s[i]);
        }
    } else {
        if (s[0] == '&' || s[0] == '*' || s[0] == '!' || s[0] == '|' || s[0] == '>')
            throw ParseError(line, "anchors, aliases, tags and block scalars are not supported");
        n.scalar = s;
    }
    return n;
}

// ---- flow collections: "[a, [b, c], {k: v}]" with the text already joined over lines
struct FlowParser {
    const std::string& s;
    size_t pos;
    size_t line;
    void skip() { while (pos < s.size() && (is_blank(s[pos]) || s[pos] == '\n')) pos++; }
    YamlNode value() {
        skip();
        if (pos >= s.size()) throw ParseError(line, "unexpected end of flow collection");
        if (s[pos] == '[') return sequence();
        if (s[pos] == '{') return mapping();
        return make_scalar(token(false), line);
    }
    // scalar text up to the next structural character
    std::string token(bool stop_at_colon) {
        skip();
        const size_t start = pos;
        if (pos < s.size() && (s[pos] == '"' || s[pos] == '\'')) {
            const char q = s[pos++];
            while (pos < s.size()) {
                if (q == '"' && s[pos] == '\\') { pos += 2; continue; }
                if (s[pos] == q) {
                    if (q == '\'' && pos + 1 < s.size() && s[pos + 1] == '\'') { pos += 2; continue; }
                    break;
                }
                pos++;
            }
            if (pos >= s.size()) throw ParseError(line, "unterminated quoted scalar in flow collection");
            pos++;
            return s.substr(start, pos - start);
        }
        while (pos < s.size()) {
            const char c = s[pos];
            if (c == ',' || c == ']' || c == '}' || c == '[' || c == '{') break;
            if (stop_at_colon && c == ':' && (pos + 1 >= s.size() || is_blank(s[pos + 1]) || s[pos + 1] == '\n')) break;
            pos++;
        }
        return s.substr(start, pos - start);
    }
    YamlNode sequence() {
        YamlNode n;
        n.kind = YamlNode::Sequence;
        pos++;   // [
        for (;;) {
            skip();
            if (pos >= s.size()) throw ParseError(line, "unterminated flow sequence");
            if (s[pos] == ']') { pos++; break; }
            n.items.push_back(value());
            skip();
            if (pos < s.size() && s[pos] == ',') { pos++; continue; }
            if (pos < s.size() && s[pos] == ']') { pos++; break; }
            throw ParseError(line, "expected ',' or ']' in flow sequence");
        }
        return n;
    }
    YamlNode mapping() {
        YamlNode n;
        n.kind = YamlNode::Map;
        pos++;   // {
        for (;;) {
            skip();
            if (pos >= s.size()) throw ParseError(line, "unterminated flow mapping");
            if (s[pos] == '}') { pos++; break; }
            const YamlNode key = make_scalar(token(true), line);
            skip();
            if (pos >= s.size() || s[pos] != ':') throw ParseError(line, "expected ':' in flow mapping");
            pos++;
            n.entries.emplace_back(key.scalar, value());
            skip();
            if (pos < s.size() && s[pos] == ',') { pos++; continue; }
            if (pos < s.size() && s[pos] == '}') { pos++; break; }
            throw ParseError(line, "expected ',' or '}' in flow mapping");
        }
        return n;
    }
};

struct BlockParser {
    std::vector<Line> lines;
    size_t at = 0;

    bool done() const { return at >= lines.size(); }

    // net count of open flow brackets in `s` (outside quotes)
    static int bracket_balance(const std::string& s) {
        int depth = 0;
        char quote = 0;
        for (size_t i = 0; i < s.size(); i++) {
            const char c = s[i];
            if (quote) {
                if (quote == '"' && c == '\\') { i++; continue; }
                if (c == quote) quote = 0;
            } else if (c == '"' || c == '\'') quote = c;
            else if (c == '[' || c == '{') depth++;
            else if (c == ']' || c == '}') depth--;
        }
        return depth;
    }

    // A value written on the rest of a line: a flow collection (possibly continued on the following
    // lines, which are consumed) or a scalar.  `at` points to the line after the one `text` came from.
    YamlNode inline_value(const std::string& text, size_t line) {
        const std::string t = trim(text);
        if (!t.empty() && (t[0] == '[' || t[0] == '{')) {
            std::string joined = t;
            int depth = bracket_balance(joined);
            while (depth > 0) {
                if (done()) throw ParseError(line, "unterminated flow collection");
                joined += "\n" + lines[at].text;
                depth = bracket_balance(joined);
                at++;
            }
            FlowParser fp{joined, 0, line};
            YamlNode n = fp.value();
            fp.skip();
            if (fp.pos != joined.size()) throw ParseError(line, "text after the end of a flow collection");
            return n;
        }
        return make_scalar(t, line);
    }

    // position of the ':' that ends a mapping key in `s`, or npos
    static size_t key_colon(const std::string& s) {
        char quote = 0;
        for (size_t i = 0; i < s.size(); i++) {
            const char c = s[i];
            if (quote) {
                if (quote == '"' && c == '\\') { i++; continue; }
                if (c == quote) quote = 0;
            } else if ((c == '"' || c == '\'') && i == 0) quote = c;
            else if (c == '[' || c == '{') return std::string::npos;     // a flow collection, not a key
            else if (c == ':' && (i + 1 == s.size() || is_blank(s[i + 1]))) return i;
        }
        return std::string::npos;
    }

    static bool is_seq_item(const std::string& s) { return s == "-" || (s.size() > 1 && s[0] == '-' && is_blank(s[1])); }

    // the block collection (or scalar) whose first line is lines[at], all of it at indentation `indent`
    YamlNode block(int indent) {
        if (done()) { YamlNode n; n.kind = YamlNode::Null; return n; }
        if (is_seq_item(lines[at].text)) return sequence(indent);
        if (key_colon(lines[at].text) != std::string::npos) return mapping(indent);
        const Line l = lines[at++];
        return inline_value(l.text, l.number);
    }

    YamlNode sequence(int indent) {
        YamlNode n;
        n.kind = YamlNode::Sequence;
        while (!done() && lines[at].indent == indent && is_seq_item(lines[at].text)) {
            Line& l = lines[at];
            // "- rest": the item starts where `rest` starts; turn the line into that
            size_t p = 1;
            while (p < l.text.size() && is_blank(l.text[p])) p++;
            if (p >= l.text.size()) {                       // "-" alone: the item is the more indented block below
                at++;
                if (!done() && lines[at].indent > indent) n.items.push_back(block(lines[at].indent));
                else { YamlNode z; z.kind = YamlNode::Null; n.items.push_back(z); }
                continue;
            }
            l.indent = indent + (int)p;
            l.text = l.text.substr(p);
            n.items.push_back(block(l.indent));
        }
        if (!done() && lines[at].indent > indent) throw ParseError(lines[at].number, "unexpected indentation");
        return n;
    }

    YamlNode mapping(int indent) {
        YamlNode n;
        n.kind = YamlNode::Map;
        while (!done() && lines[at].indent == indent) {
            const Line l = lines[at];
            if (is_seq_item(l.text)) break;
            const size_t colon = key_colon(l.text);
            if (colon == std::string::npos) throw ParseError(l.number, "expected 'key: value'");
            const YamlNode key = make_scalar(l.text.substr(0, colon), l.number);
            if (key.kind != YamlNode::Scalar) throw ParseError(l.number, "empty mapping key");
            const std::string rest = trim(l.text.substr(colon + 1));
            at++;
            YamlNode value;
            if (!rest.empty()) {
                value = inline_value(rest, l.number);
            } else if (!done() && lines[at].indent > indent) {
                value = block(lines[at].indent);
            } else if (!done() && lines[at].indent == indent && is_seq_item(lines[at].text)) {
                value = sequence(indent);                    // "key:" followed by "- item" at the same indentation
            } else {
                value.kind = YamlNode::Null;
            }
            n.entries.emplace_back(key.scalar, std::move(value));
        }
        if (!done() && lines[at].indent > indent) throw ParseError(lines[at].number, "unexpected indentation");
        return n;
    }
};

}  // namespace

const YamlNode& YamlNode::operator[](const std::string& key) const {
    if (kind == Map)
        for (const auto& e : entries)
            if (e.first == key) return e.second;
    return kUndefined;
}

bool yaml_to_double(const std::string& s, double& out) {
    if (s.empty() || is_blank(s[0])) return false;
    if (s == ".inf" || s == ".Inf" || s == ".INF" || s == "+.inf" || s == "+.Inf" || s == "+.INF") { out = std::numeric_limits<double>::infinity(); return true; }
    if (s == "-.inf" || s == "-.Inf" || s == "-.INF") { out = -std::numeric_limits<double>::infinity(); return true; }
    if (s == ".nan" || s == ".NaN" || s == ".NAN") { out = std::numeric_limits<double>::quiet_NaN(); return true; }
    // what operator>>(double) accepts: decimal floating point, no hex floats, no "inf"/"nan" words
    size_t i = 0;
    if (s[i] == '+' || s[i] == '-') i++;
    bool digits = false;
    while (i < s.size() && s[i] >= '0' && s[i] <= '9') { i++; digits = true; }
    if (i < s.size() && s[i] == '.') { i++; while (i < s.size() && s[i] >= '0' && s[i] <= '9') { i++; digits = true; } }
    if (!digits) return false;
    if (i < s.size() && (s[i] == 'e' || s[i] == 'E')) {
        size_t j = i + 1;
        if (j < s.size() && (s[j] == '+' || s[j] == '-')) j++;
        bool ed = false;
        while (j < s.size() && s[j] >= '0' && s[j] <= '9') { j++; ed = true; }
        if (!ed) return false;
        i = j;
    }
    while (i < s.size() && is_blank(s[i])) i++;              // trailing white space is allowed (stream >> std::ws)
    if (i != s.size()) return false;
    out = std::strtod(s.c_str(), nullptr);
    return true;
}

bool yaml_to_int(const std::string& s, int& out) {
    if (s.empty() || is_blank(s[0])) return false;
    errno = 0;
    char* end = nullptr;
    const long v = std::strtol(s.c_str(), &end, 0);          // stream.unsetf(std::ios::dec): 0x.. and 0.. prefixes
    if (end == s.c_str() || errno == ERANGE || v < INT_MIN || v > INT_MAX) return false;
    while (*end && is_blank(*end)) end++;
    if (*end) return false;
    out = (int)v;
    return true;
}

bool yaml_parse(const std::string& text, YamlNode& root, std::string& error) {
    try {
        BlockParser bp;
        size_t start = 0, number = 0;
        while (start <= text.size()) {
            size_t nl = text.find('\n', start);
            if (nl == std::string::npos) nl = text.size();
            number++;
            std::string raw = text.substr(start, nl - start);
            start = nl + 1;
            if (number == 1 && raw.size() >= 3 && (unsigned char)raw[0] == 0xEF && (unsigned char)raw[1] == 0xBB && (unsigned char)raw[2] == 0xBF)
                raw = raw.substr(3);                          // byte-order mark
            const std::string body = strip_comment(raw);
            size_t ind = 0;
            while (ind < body.size() && body[ind] == ' ') ind++;
            if (ind == body.size()) continue;                 // blank or comment-only
            if (body[ind] == '\t') throw ParseError(number, "tab used for indentation");
            if (body.compare(ind, 3, "---") == 0 && ind == 0) {
                if (!bp.lines.empty()) throw ParseError(number, "multiple documents are not supported");
                continue;
            }
            if (body[ind] == '%') throw ParseError(number, "directives are not supported");
            bp.lines.push_back(Line{number, (int)ind, body.substr(ind)});
        }
        if (bp.lines.empty()) { root = YamlNode(); root.kind = YamlNode::Null; return true; }
        root = bp.block(bp.lines[0].indent);
        if (!bp.done()) throw ParseError(bp.lines[bp.at].number, "unexpected content after the end of the document");
        return true;
    } catch (const std::exception& ex) {
        error = ex.what();
        return false;
    }
}

}  // namespace bdpt_host
