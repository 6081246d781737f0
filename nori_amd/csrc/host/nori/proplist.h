/*
 * nori/proplist.h -- PropertyList: typed property bag handed to plugin
 * constructors.  Interface of the reference's include/nori/proplist.h:19-123
 * (set/get per type, get-with-default, "missing" / "wrong type" exceptions,
 * duplicate warning of src/proplist.cpp:13-14).
 */
#pragma once
#include <map>
#include <nori/common.h>

NORI_NAMESPACE_BEGIN

class PropertyList {
public:
    PropertyList() {}
#define NORI_PROP(Type, Name)                                                   \
    void set##Name(const std::string &name, const Type &value);                 \
    Type get##Name(const std::string &name) const;                              \
    Type get##Name(const std::string &name, const Type &defaultValue) const;
    NORI_PROP(bool, Boolean)
    NORI_PROP(int, Integer)
    NORI_PROP(float, Float)
    NORI_PROP(std::string, String)
    NORI_PROP(Color3f, Color)
    NORI_PROP(Point3f, Point)
    NORI_PROP(Transform, Transform)
#undef NORI_PROP
    /* Vector3f and Point3f are one C++ type here but distinct property kinds,
       as <vector> and <point> are in the scene format (parser.cpp:93-94) */
    void setVector(const std::string &name, const Vector3f &value);
    Vector3f getVector(const std::string &name) const;
    Vector3f getVector(const std::string &name, const Vector3f &defaultValue) const;

private:
    struct Property {
        enum Type { boolean_type, integer_type, float_type, string_type, color_type, point_type, vector_type, transform_type } type = boolean_type;
        bool boolean_value = false;
        int integer_value = 0;
        float float_value = 0;
        std::string string_value;
        Color3f color_value;
        Point3f point_value;
        Vector3f vector_value;
        Transform transform_value;
    };
    Property &touch(const std::string &name, int type);
    const Property *lookup(const std::string &name, int type, bool required) const;
    std::map<std::string, Property> m_properties;
};

NORI_NAMESPACE_END
