// oracle/_ref shim for <tinyply/tinyply.h> (TEST INFRASTRUCTURE ONLY): PLY file IO is out of scope; the types only have to
// exist because SlamCore/io.h is included by headers on the path.
#ifndef CTGN_ORACLE_TINYPLY_SHIM_H
#define CTGN_ORACLE_TINYPLY_SHIM_H
#include <cstdint>
#include <string>
#include <vector>
#include <memory>
#include <istream>
#include <ostream>
namespace tinyply {
    enum class Type : uint8_t { INVALID, INT8, UINT8, INT16, UINT16, INT32, UINT32, FLOAT32, FLOAT64 };
    struct Buffer { uint8_t *get() { return nullptr; } const uint8_t *get_const() const { return nullptr; } std::size_t size_bytes() const { return 0; } };
    struct PlyData { Type t = Type::INVALID; Buffer buffer; std::size_t count = 0; bool isList = false; };
    struct PlyProperty { std::string name; Type propertyType = Type::INVALID; bool isList = false; Type listType = Type::INVALID; std::size_t listCount = 0; };
    struct PlyElement { std::string name; std::size_t size = 0; std::vector<PlyProperty> properties; };
    struct PlyFile {
        bool parse_header(std::istream &) { return false; }
        void read(std::istream &) {}
        void write(std::ostream &, bool) {}
        std::vector<PlyElement> get_elements() const { return {}; }
        std::vector<std::string> get_info() const { return {}; }
        std::vector<std::string> &get_comments() { static std::vector<std::string> c; return c; }
        std::shared_ptr<PlyData> request_properties_from_element(const std::string &, const std::vector<std::string>, uint32_t = 0) { return nullptr; }
        void add_properties_to_element(const std::string &, const std::vector<std::string>, Type, std::size_t, const uint8_t *, Type, std::size_t) {}
        bool is_binary_file() const { return false; }
    };
}
#endif
