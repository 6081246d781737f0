/*
 * proplist.cpp -- PropertyList accessors.  Observable behaviour follows the
 * reference (src/proplist.cpp:11-48): a duplicate set warns on stderr and
 * overwrites; a typed get throws "Property '<n>' is missing!" or "... has the
 * wrong type! (expected <kind>)!"; get-with-default only falls back when the
 * name is absent.
 */
#include <nori/proplist.h>

NORI_NAMESPACE_BEGIN

namespace {
const char *kindName(int type) {
    static const char *names[] = {"boolean", "integer", "float", "string", "color", "point", "vector", "transform"};
    return names[type];
}
} // namespace

#define NORI_PROP_IMPL(Type, Name, kind)                                                             \
    void PropertyList::set##Name(const std::string &name, const Type &value) {                       \
        Property &p = touch(name, Property::kind##_type);                                            \
        p.kind##_value = value;                                                                      \
    }                                                                                                \
    Type PropertyList::get##Name(const std::string &name) const {                                    \
        const Property *p = lookup(name, Property::kind##_type, true);                               \
        return p->kind##_value;                                                                      \
    }                                                                                                \
    Type PropertyList::get##Name(const std::string &name, const Type &fallback) const {              \
        const Property *p = lookup(name, Property::kind##_type, false);                              \
        return p ? p->kind##_value : fallback;                                                       \
    }

PropertyList::Property &PropertyList::touch(const std::string &name, int type) {
    if (m_properties.count(name))
        cerr << "Property \"" << name << "\" was specified multiple times!" << endl;
    Property &p = m_properties[name];
    p.type = (Property::Type) type;
    return p;
}

const PropertyList::Property *PropertyList::lookup(const std::string &name, int type, bool required) const {
    auto it = m_properties.find(name);
    if (it == m_properties.end()) {
        if (required) throw NoriException("Property '%s' is missing!", name);
        return nullptr;
    }
    if ((int) it->second.type != type)
        throw NoriException("Property '%s' has the wrong type! (expected <%s>)!", name, kindName(type));
    return &it->second;
}

NORI_PROP_IMPL(bool, Boolean, boolean)
NORI_PROP_IMPL(int, Integer, integer)
NORI_PROP_IMPL(float, Float, float)
NORI_PROP_IMPL(std::string, String, string)
NORI_PROP_IMPL(Color3f, Color, color)
NORI_PROP_IMPL(Point3f, Point, point)
NORI_PROP_IMPL(Vector3f, Vector, vector)
NORI_PROP_IMPL(Transform, Transform, transform)

NORI_NAMESPACE_END
