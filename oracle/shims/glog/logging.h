// oracle/_ref shim for <glog/logging.h> (TEST INFRASTRUCTURE ONLY, see mini_eigen.h).
// LOG(x) << ... is swallowed (FATAL throws); a failed CHECK throws ctgn_ref_shim::CheckFailure with the streamed
// message where glog would abort the process -- the extern "C" wrapper of oracle/_ref turns it into an error code.
#ifndef CTGN_ORACLE_GLOG_SHIM_H
#define CTGN_ORACLE_GLOG_SHIM_H
#include <sstream>
#include <stdexcept>
#include <string>
#include <iostream>
#include <set>
#include <map>
#include <vector>
#include <list>
#include <memory>
#include <optional>
#include <algorithm>

namespace ctgn_ref_shim {
    struct CheckFailure : public std::runtime_error {
        explicit CheckFailure(const std::string &m) : std::runtime_error(m) {}
    };
    struct NullStream {
        template<typename T> NullStream &operator<<(const T &) { return *this; }
        NullStream &operator<<(std::ostream &(*)(std::ostream &)) { return *this; }
    };
    struct FatalStream {
        std::ostringstream ss;
        FatalStream(const char *file, int line, const char *what) { ss << "Check failed: " << what << " [" << file << ":" << line << "] "; }
        template<typename T> FatalStream &operator<<(const T &v) { ss << v; return *this; }
        FatalStream &operator<<(std::ostream &(*f)(std::ostream &)) { f(ss); return *this; }
        [[noreturn]] ~FatalStream() noexcept(false) { throw CheckFailure(ss.str()); }
    };
    struct Voidify { void operator&(NullStream &) {} void operator&(FatalStream &) {} };
}
namespace google {
    inline void InitGoogleLogging(const char *) {}
    inline void InstallFailureSignalHandler() {}
}
#define CTGN_GLOG_SEVERITY_INFO ctgn_ref_shim::NullStream()
#define CTGN_GLOG_SEVERITY_WARNING ctgn_ref_shim::NullStream()
#define CTGN_GLOG_SEVERITY_ERROR ctgn_ref_shim::NullStream()
#define CTGN_GLOG_SEVERITY_FATAL ctgn_ref_shim::FatalStream(__FILE__, __LINE__, "LOG(FATAL)")
#define LOG(sev) CTGN_GLOG_SEVERITY_##sev
#define VLOG(n) ctgn_ref_shim::NullStream()
#define DLOG(sev) ctgn_ref_shim::NullStream()
#define LOG_IF(sev, cond) if (cond) CTGN_GLOG_SEVERITY_##sev
#define LOG_EVERY_N(sev, n) ctgn_ref_shim::NullStream()
#define LOG_FIRST_N(sev, n) ctgn_ref_shim::NullStream()
#define CHECK(cond) if (cond) {} else ctgn_ref_shim::FatalStream(__FILE__, __LINE__, #cond)
#define CTGN_CHECK_OP(a, b, op) if ((a) op (b)) {} else ctgn_ref_shim::FatalStream(__FILE__, __LINE__, #a " " #op " " #b)
#define CHECK_EQ(a, b) CTGN_CHECK_OP(a, b, ==)
#define CHECK_NE(a, b) CTGN_CHECK_OP(a, b, !=)
#define CHECK_LT(a, b) CTGN_CHECK_OP(a, b, <)
#define CHECK_LE(a, b) CTGN_CHECK_OP(a, b, <=)
#define CHECK_GT(a, b) CTGN_CHECK_OP(a, b, >)
#define CHECK_GE(a, b) CTGN_CHECK_OP(a, b, >=)
#define CHECK_NOTNULL(p) (p)
#define DCHECK(cond) CHECK(cond)
#endif
