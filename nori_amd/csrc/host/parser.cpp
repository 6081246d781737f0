/*
 * parser.cpp -- loadFromXML: builds the NoriObject graph from a scene file.
 * Grammar and checks of the reference's src/parser.cpp:16-305: the 11 object
 * tags and 13 property / transform tags (parser.cpp:79-102), exact attribute
 * sets per tag (:105-116), structural rules (:140-157), children constructed
 * before their parent, then addChild + setParent in document order, then
 * activate() (:166-199); transform operations LEFT-multiply the running
 * transform (:238-292); errors are re-thrown with file / line context.
 */
#include <nori/plugins.h>
#include <nori/xml.h>

#include <fstream>
#include <set>

NORI_NAMESPACE_BEGIN

namespace {

enum ETag {
    EBoolean = NoriObject::EClassTypeCount, EInteger, EFloat, EString, EPoint, EVector, EColor,
    ETransform, ETranslate, EMatrix, ERotate, EScale, ELookAt, EInvalid
};

const std::map<std::string, int> &tagTable() {
    static const std::map<std::string, int> tags = {
        {"scene", NoriObject::EScene}, {"mesh", NoriObject::EMesh}, {"bsdf", NoriObject::EBSDF},
        {"emitter", NoriObject::EEmitter}, {"camera", NoriObject::ECamera}, {"medium", NoriObject::EMedium},
        {"phase", NoriObject::EPhaseFunction}, {"integrator", NoriObject::EIntegrator},
        {"sampler", NoriObject::ESampler}, {"rfilter", NoriObject::EReconstructionFilter}, {"test", NoriObject::ETest},
        {"boolean", EBoolean}, {"integer", EInteger}, {"float", EFloat}, {"string", EString}, {"point", EPoint},
        {"vector", EVector}, {"color", EColor}, {"transform", ETransform}, {"translate", ETranslate},
        {"matrix", EMatrix}, {"rotate", ERotate}, {"scale", EScale}, {"lookat", ELookAt}};
    return tags;
}

Vector3f toVector3f(const std::string &str) {
    std::vector<std::string> tokens = tokenize(str);
    if (tokens.size() != 3) throw NoriException("Expected 3 values");
    return Vector3f(toFloat(tokens[0]), toFloat(tokens[1]), toFloat(tokens[2]));
}

struct Parser {
    const std::string &filename;
    const std::string &text;
    float transform[16];

    std::string where(const XmlNode &n) const { return xmlOffsetToString(text, n.offset); }

    void checkAttributes(const XmlNode &node, std::set<std::string> attrs) const {
        for (auto &a : node.attributes) {
            auto it = attrs.find(a.first);
            if (it == attrs.end())
                throw NoriException("Error while parsing \"%s\": unexpected attribute \"%s\" in \"%s\" at %s", filename, a.first, node.name, where(node));
            attrs.erase(it);
        }
        if (!attrs.empty())
            throw NoriException("Error while parsing \"%s\": missing attribute \"%s\" in \"%s\" at %s", filename, *attrs.begin(), node.name, where(node));
    }

    const std::string &attr(const XmlNode &n, const char *key) const { return *n.attribute(key); }

    void leftMultiply(const float *op) { mat4Mul(op, transform, transform); }

    NoriObject *parseTag(XmlNode &node, PropertyList &list, int parentTag) {
        if (node.type == XmlNode::Comment || node.type == XmlNode::Declaration) return nullptr;
        if (node.type != XmlNode::Element)
            throw NoriException("Error while parsing \"%s\": unexpected content at %s", filename, where(node));
        auto it = tagTable().find(node.name);
        if (it == tagTable().end())
            throw NoriException("Error while parsing \"%s\": unexpected tag \"%s\" at %s", filename, node.name, where(node));
        const int tag = it->second;

        const bool hasParent = parentTag != EInvalid;
        const bool parentIsObject = hasParent && parentTag < NoriObject::EClassTypeCount;
        const bool currentIsObject = tag < NoriObject::EClassTypeCount;
        const bool parentIsTransform = parentTag == ETransform;
        const bool currentIsTransformOp = tag == ETranslate || tag == ERotate || tag == EScale || tag == ELookAt || tag == EMatrix;

        if (!hasParent && !currentIsObject)
            throw NoriException("Error while parsing \"%s\": root element \"%s\" must be a Nori object (at %s)", filename, node.name, where(node));
        if (parentIsTransform != currentIsTransformOp)
            throw NoriException("Error while parsing \"%s\": transform nodes can only contain transform operations (at %s)", filename, where(node));
        if (hasParent && !parentIsObject && !(parentIsTransform && currentIsTransformOp))
            throw NoriException("Error while parsing \"%s\": node \"%s\" requires a Nori object as parent (at %s)", filename, node.name, where(node));

        if (tag == NoriObject::EScene && !node.attribute("type"))
            node.attributes.emplace_back("type", "scene");
        else if (tag == ETransform)
            mat4Identity(transform);

        PropertyList propList;
        std::vector<NoriObject *> children;
        for (auto &ch : node.children) {
            NoriObject *child = parseTag(*ch, propList, tag);
            if (child) children.push_back(child);
        }

        NoriObject *result = nullptr;
        try {
            if (currentIsObject) {
                checkAttributes(node, {"type"});
                result = NoriObjectFactory::createInstance(attr(node, "type"), propList);
                if (result->getClassType() != (int) tag)
                    throw NoriException("Unexpectedly constructed an object of type <%s> (expected type <%s>): %s",
                                        NoriObject::classTypeName(result->getClassType()),
                                        NoriObject::classTypeName((NoriObject::EClassType) tag), result->toString());
                for (auto ch : children) {
                    result->addChild(ch);
                    ch->setParent(result);
                }
                result->activate();
            } else {
                switch (tag) {
                case EString: checkAttributes(node, {"name", "value"}); list.setString(attr(node, "name"), attr(node, "value")); break;
                case EFloat: checkAttributes(node, {"name", "value"}); list.setFloat(attr(node, "name"), toFloat(attr(node, "value"))); break;
                case EInteger: checkAttributes(node, {"name", "value"}); list.setInteger(attr(node, "name"), toInt(attr(node, "value"))); break;
                case EBoolean: checkAttributes(node, {"name", "value"}); list.setBoolean(attr(node, "name"), toBool(attr(node, "value"))); break;
                case EPoint: checkAttributes(node, {"name", "value"}); list.setPoint(attr(node, "name"), toVector3f(attr(node, "value"))); break;
                case EVector: checkAttributes(node, {"name", "value"}); list.setVector(attr(node, "name"), toVector3f(attr(node, "value"))); break;
                case EColor: {
                    checkAttributes(node, {"name", "value"});
                    Vector3f c = toVector3f(attr(node, "value"));
                    list.setColor(attr(node, "name"), Color3f(c.x(), c.y(), c.z()));
                } break;
                case ETransform: checkAttributes(node, {"name"}); list.setTransform(attr(node, "name"), Transform(transform)); break;
                case ETranslate: {
                    checkAttributes(node, {"value"});
                    Vector3f v = toVector3f(attr(node, "value"));
                    float op[16]; mat4Identity(op);
                    op[3] = v.x(); op[7] = v.y(); op[11] = v.z();
                    leftMultiply(op);
                } break;
                case EMatrix: {
                    checkAttributes(node, {"value"});
                    std::vector<std::string> tokens = tokenize(attr(node, "value"));
                    if (tokens.size() != 16) throw NoriException("Expected 16 values");
                    float op[16];
                    for (int i = 0; i < 16; ++i) op[i] = toFloat(tokens[i]);
                    leftMultiply(op);
                } break;
                case EScale: {
                    checkAttributes(node, {"value"});
                    Vector3f v = toVector3f(attr(node, "value"));
                    float op[16]; mat4Identity(op);
                    op[0] = v.x(); op[5] = v.y(); op[10] = v.z();
                    leftMultiply(op);
                } break;
                case ERotate: {
                    checkAttributes(node, {"angle", "axis"});
                    const float angle = degToRad(toFloat(attr(node, "angle")));
                    const Vector3f a = toVector3f(attr(node, "axis")).normalized();
                    /* Eigen::AngleAxis::toRotationMatrix */
                    const float s = std::sin(angle), c = std::cos(angle);
                    const Vector3f sa = a * s, ca = a * (1.0f - c);
                    float op[16]; mat4Identity(op);
                    float t;
                    t = ca.x() * a.y(); op[1] = t - sa.z(); op[4] = t + sa.z();
                    t = ca.x() * a.z(); op[2] = t + sa.y(); op[8] = t - sa.y();
                    t = ca.y() * a.z(); op[6] = t - sa.x(); op[9] = t + sa.x();
                    op[0] = ca.x() * a.x() + c; op[5] = ca.y() * a.y() + c; op[10] = ca.z() * a.z() + c;
                    leftMultiply(op);
                } break;
                case ELookAt: {
                    checkAttributes(node, {"origin", "target", "up"});
                    const Vector3f origin = toVector3f(attr(node, "origin")), target = toVector3f(attr(node, "target")), up = toVector3f(attr(node, "up"));
                    const Vector3f dir = (target - origin).normalized();
                    const Vector3f left = up.normalized().cross(dir).normalized();
                    const Vector3f newUp = dir.cross(left).normalized();
                    float op[16] = {left.x(), newUp.x(), dir.x(), origin.x(),
                                    left.y(), newUp.y(), dir.y(), origin.y(),
                                    left.z(), newUp.z(), dir.z(), origin.z(),
                                    0, 0, 0, 1};
                    leftMultiply(op);
                } break;
                default: throw NoriException("Unhandled element \"%s\"", node.name);
                }
            }
        } catch (const NoriException &e) {
            throw NoriException("Error while parsing \"%s\": %s (at %s)", filename, e.what(), where(node));
        }
        return result;
    }
};

} // namespace

NoriObject *loadFromXML(const std::string &filename) {
    std::ifstream is(filename, std::ios::binary);
    if (is.fail()) throw NoriException("Error while parsing \"%s\": File was not found", filename);
    std::string text((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
    std::vector<std::unique_ptr<XmlNode>> doc = parseXml(text, filename);
    Parser p{filename, text, {}};
    mat4Identity(p.transform);
    PropertyList list;
    for (auto &n : doc)
        if (n->type == XmlNode::Element) return p.parseTag(*n, list, EInvalid);
    throw NoriException("Error while parsing \"%s\": no root element", filename);
}

NORI_NAMESPACE_END
