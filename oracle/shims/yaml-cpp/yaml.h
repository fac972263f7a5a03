// oracle/_ref shim for <yaml-cpp/yaml.h> (TEST INFRASTRUCTURE ONLY, see ../mini_eigen.h).
// Configuration parsing is control plane and out of scope; oracle/_ref sets option structs
// directly.  YAML::Node only has to exist so that the reference's headers and option loaders compile: every node is
// undefined, conversions throw.
#ifndef CTGN_ORACLE_YAML_SHIM_H
#define CTGN_ORACLE_YAML_SHIM_H
#include <cstddef>
#include <ostream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
namespace YAML {
    struct NodeType { enum value { Undefined, Null, Scalar, Sequence, Map }; };
    class Exception : public std::runtime_error { public: explicit Exception(const std::string &m) : std::runtime_error(m) {} };
    class Node;
    struct iterator_value;
    class Node {
    public:
        Node() {}
        template<typename T> Node(const T &) {}
        explicit operator bool() const { return false; }
        bool operator!() const { return true; }
        bool IsDefined() const { return false; }
        bool IsNull() const { return false; }
        bool IsScalar() const { return false; }
        bool IsSequence() const { return false; }
        bool IsMap() const { return false; }
        NodeType::value Type() const { return NodeType::Undefined; }
        std::size_t size() const { return 0; }
        template<typename T> T as() const { throw Exception("oracle/_ref YAML shim: no YAML support"); }
        template<typename T, typename S> T as(const S &fallback) const { return T(fallback); }
        std::string Scalar() const { return std::string(); }
        std::string Tag() const { return std::string(); }
        template<typename K> Node operator[](const K &) const { return Node(); }
        template<typename K> Node operator[](const K &) { return Node(); }
        template<typename T> Node &operator=(const T &) { return *this; }
        template<typename T> void push_back(const T &) {}
        template<typename K> bool remove(const K &) { return false; }
        void reset(const Node & = Node()) {}
        struct iterator {
            iterator_value *operator->() const;
            iterator_value &operator*() const;
            iterator &operator++() { return *this; }
            iterator operator++(int) { return *this; }
            bool operator==(const iterator &) const { return true; }
            bool operator!=(const iterator &) const { return false; }
        };
        typedef iterator const_iterator;
        iterator begin() const { return iterator(); }
        iterator end() const { return iterator(); }
    };
    struct iterator_value : public Node, public std::pair<Node, Node> {};
    inline iterator_value &Node::iterator::operator*() const { static iterator_value v; return v; }
    inline iterator_value *Node::iterator::operator->() const { static iterator_value v; return &v; }
    inline Node Load(const std::string &) { throw Exception("oracle/_ref YAML shim: no YAML support"); }
    inline Node LoadFile(const std::string &) { throw Exception("oracle/_ref YAML shim: no YAML support"); }
    inline Node Clone(const Node &n) { return n; }
    inline std::ostream &operator<<(std::ostream &os, const Node &) { return os << "<yaml>"; }
    class Emitter {
    public:
        template<typename T> Emitter &operator<<(const T &) { return *this; }
        const char *c_str() const { return ""; }
    };
}
#endif
