/* xml.cpp -- recursive-descent parser behind nori/xml.h. */
#include <nori/xml.h>

NORI_NAMESPACE_BEGIN

std::string xmlOffsetToString(const std::string &text, size_t pos) {
    int line = 1; size_t lineStart = 0;
    for (size_t i = 0; i < pos && i < text.size(); ++i)
        if (text[i] == '\n') { ++line; lineStart = i + 1; }
    return format("line %i, col %i", line, (int) (pos - lineStart) + 1);
}

namespace {

struct Cursor {
    const std::string &s; const std::string &file; size_t i = 0;
    [[noreturn]] void fail(const std::string &what, size_t at) const {
        throw NoriException("Error while parsing \"%s\": %s (at %s)", file, what, xmlOffsetToString(s, at));
    }
    bool eof() const { return i >= s.size(); }
    bool startsWith(const char *lit) const { return s.compare(i, std::strlen(lit), lit) == 0; }
    void skipSpace() { while (!eof() && std::isspace((unsigned char) s[i])) ++i; }
    static bool nameChar(char c) { return std::isalnum((unsigned char) c) || c == '_' || c == '-' || c == ':' || c == '.'; }
    std::string name() {
        size_t b = i;
        while (!eof() && nameChar(s[i])) ++i;
        if (b == i) fail("expected a name", i);
        return s.substr(b, i - b);
    }
    static std::string decode(const std::string &raw) {
        std::string out; out.reserve(raw.size());
        for (size_t k = 0; k < raw.size(); ++k) {
            /* attribute-value normalisation (XML 1.0 sec. 3.3.3; pugixml's default
               parse_wconv_attribute): literal tab / newline / CR become spaces */
            if (raw[k] == '\t' || raw[k] == '\n' || raw[k] == '\r') { out += ' '; continue; }
            if (raw[k] != '&') { out += raw[k]; continue; }
            static const std::pair<const char *, char> ents[] = {{"&amp;", '&'}, {"&lt;", '<'}, {"&gt;", '>'}, {"&quot;", '"'}, {"&apos;", '\''}};
            bool hit = false;
            for (auto &e : ents)
                if (raw.compare(k, std::strlen(e.first), e.first) == 0) { out += e.second; k += std::strlen(e.first) - 1; hit = true; break; }
            if (!hit) out += raw[k];
        }
        return out;
    }
    void attributes(XmlNode &n, const char *closers) {
        while (true) {
            skipSpace();
            if (eof()) fail("unexpected end of file inside a tag", n.offset);
            if (std::strchr(closers, s[i])) return;
            std::string key = name();
            skipSpace();
            if (eof() || s[i] != '=') fail("expected '=' after attribute name", i);
            ++i; skipSpace();
            if (eof() || (s[i] != '"' && s[i] != '\'')) fail("expected a quoted attribute value", i);
            char q = s[i++]; size_t b = i;
            while (!eof() && s[i] != q) ++i;
            if (eof()) fail("unterminated attribute value", b);
            n.attributes.emplace_back(key, decode(s.substr(b, i - b)));
            ++i;
        }
    }
    /* parses one node starting at '<' or text */
    std::unique_ptr<XmlNode> node() {
        std::unique_ptr<XmlNode> n(new XmlNode());
        n->offset = i;
        if (startsWith("<!--")) {
            size_t e = s.find("-->", i + 4);
            if (e == std::string::npos) fail("unterminated comment", i);
            n->type = XmlNode::Comment; i = e + 3; return n;
        }
        if (startsWith("<?")) {
            size_t e = s.find("?>", i + 2);
            if (e == std::string::npos) fail("unterminated declaration", i);
            n->type = XmlNode::Declaration; i = e + 2; return n;
        }
        if (startsWith("<!")) {                       /* DOCTYPE etc. */
            size_t e = s.find('>', i);
            if (e == std::string::npos) fail("unterminated markup declaration", i);
            n->type = XmlNode::Declaration; i = e + 1; return n;
        }
        if (s[i] != '<') {                            /* character data */
            size_t b = i;
            while (!eof() && s[i] != '<') ++i;
            n->type = XmlNode::Text; n->name = s.substr(b, i - b); return n;
        }
        ++i;
        n->name = name();
        attributes(*n, "/>");
        if (s[i] == '/') {
            if (i + 1 >= s.size() || s[i + 1] != '>') fail("expected '/>'", i);
            i += 2; return n;
        }
        ++i;                                          /* '>' */
        while (true) {
            size_t save = i; skipSpace();
            if (eof()) fail(format("missing closing tag for <%s>", n->name), n->offset);
            if (startsWith("</")) {
                i += 2; std::string close = name(); skipSpace();
                if (close != n->name) fail(format("mismatched closing tag </%s> for <%s>", close, n->name), save);
                if (eof() || s[i] != '>') fail("expected '>'", i);
                ++i; return n;
            }
            if (s[i] != '<') i = save;                 /* keep leading space of real text */
            std::unique_ptr<XmlNode> c = node();
            if (c->type == XmlNode::Text) {
                bool blank = true;
                for (char ch : c->name) if (!std::isspace((unsigned char) ch)) blank = false;
                if (blank) continue;
            }
            n->children.push_back(std::move(c));
        }
    }
};

} // namespace

std::vector<std::unique_ptr<XmlNode>> parseXml(const std::string &text, const std::string &filename) {
    Cursor c{text, filename};
    std::vector<std::unique_ptr<XmlNode>> top;
    while (true) {
        c.skipSpace();
        if (c.eof()) break;
        std::unique_ptr<XmlNode> n = c.node();
        if (n->type == XmlNode::Text) c.fail("unexpected content outside of the root element", n->offset);
        top.push_back(std::move(n));
    }
    int roots = 0;
    for (auto &n : top) if (n->type == XmlNode::Element) ++roots;
    if (roots != 1) c.fail(roots == 0 ? "No document element found" : "multiple root elements", text.size() ? text.size() - 1 : 0);
    return top;
}

NORI_NAMESPACE_END
