/*
 * nori/xml.h -- minimal XML DOM for Nori scene files (the reference uses
 * pugixml, an empty submodule in the snapshot).  Supports what scenes/*.xml
 * use: declaration, comments, nested elements, quoted attributes, entities.
 */
#pragma once
#include <memory>
#include <nori/common.h>

NORI_NAMESPACE_BEGIN

struct XmlNode {
    enum Type { Element, Comment, Declaration, Text } type = Element;
    std::string name;                                        /* element name */
    std::vector<std::pair<std::string, std::string>> attributes;
    std::vector<std::unique_ptr<XmlNode>> children;
    size_t offset = 0;                                       /* byte offset of '<' */

    const std::string *attribute(const std::string &key) const {
        for (auto &a : attributes) if (a.first == key) return &a.second;
        return nullptr;
    }
};

/* Parses `text`; throws NoriException("... (at line L, col C)") on malformed input.
   Returns the document's top-level nodes (declaration / comments / root element). */
std::vector<std::unique_ptr<XmlNode>> parseXml(const std::string &text, const std::string &filename);

/* "line L, col C" for a byte offset (error messages, src/parser.cpp:22-39) */
std::string xmlOffsetToString(const std::string &text, size_t pos);

NORI_NAMESPACE_END
