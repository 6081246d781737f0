/*
 * nori/object.h -- NoriObject, NoriObjectFactory, NORI_REGISTER_CLASS.
 * Same protocol as the reference's include/nori/object.h:19-149 (class types,
 * addChild / setParent / activate / toString, string -> constructor map filled
 * by static registration, createInstance throws for unknown names), so the
 * shipped scenes/*.xml load unchanged.
 */
#pragma once
#include <functional>
#include <map>
#include <nori/proplist.h>

NORI_NAMESPACE_BEGIN

class NoriObject {
public:
    enum EClassType {
        EScene = 0, EMesh, EBSDF, EPhaseFunction, EEmitter, EMedium, ECamera,
        EIntegrator, ESampler, ETest, EReconstructionFilter, EClassTypeCount
    };
    virtual ~NoriObject() {}
    virtual EClassType getClassType() const = 0;
    virtual std::string toString() const = 0;
    /* defaults of the protocol: leaf objects take no children and need no post-processing */
    virtual void addChild(NoriObject *) {
        throw NoriException("NoriObject::addChild() is not implemented for objects of type '%s'!", classTypeName(getClassType()));
    }
    virtual void setParent(NoriObject *) {}
    virtual void activate() {}

    static std::string classTypeName(EClassType type) {
        switch (type) {
        case EScene: return "scene";
        case EMesh: return "mesh";
        case EBSDF: return "bsdf";
        case EEmitter: return "emitter";
        case ECamera: return "camera";
        case EIntegrator: return "integrator";
        case ESampler: return "sampler";
        case ETest: return "test";
        default: return "<unknown>";
        }
    }
};

/* name -> constructor registry, filled by the static registrars NORI_REGISTER_CLASS plants in every
   plugin's translation unit; the table is created on first use, so registration order across
   translation units does not matter */
class NoriObjectFactory {
public:
    typedef std::function<NoriObject *(const PropertyList &)> Constructor;
    static void registerClass(const std::string &name, const Constructor &constr) { registry()[name] = constr; }
    static NoriObject *createInstance(const std::string &name, const PropertyList &propList) {
        const auto it = registry().find(name);
        if (it == registry().end()) throw NoriException("A constructor for class \"%s\" could not be found!", name);
        return it->second(propList);
    }
private:
    static std::map<std::string, Constructor> &registry() {
        static std::map<std::string, Constructor> table;
        return table;
    }
};

#define NORI_REGISTER_CLASS(cls, name)                                          \
    cls *cls##_create(const PropertyList &list) { return new cls(list); }       \
    static struct cls##_ {                                                      \
        cls##_() { NoriObjectFactory::registerClass(name, cls##_create); }      \
    } cls##__NORI_;

NORI_NAMESPACE_END
