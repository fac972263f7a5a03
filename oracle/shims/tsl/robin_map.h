// oracle/_ref shim for <tsl/robin_map.h> (TEST INFRASTRUCTURE ONLY, see mini_eigen.h).
// Same interface as the subset of tsl::robin_map the reference uses (find / end / operator[] / erase / reserve / size /
// iteration with .key() / .value()), backed by std::unordered_map.  Lookup results are identical; only the ITERATION
// ORDER differs from the real open-addressing table (it decides the output order of sub_sample_frame and of
// MapAsPointCloud, never which elements exist) -- comparisons against oracle/_ref on those outputs are set comparisons.
#ifndef CTGN_ORACLE_TSL_SHIM_H
#define CTGN_ORACLE_TSL_SHIM_H
#include <unordered_map>
#include <functional>
namespace tsl {
    template<class Key, class T, class Hash = std::hash<Key>, class KeyEqual = std::equal_to<Key>,
             class Allocator = std::allocator<std::pair<const Key, T>>>
    class robin_map {
        typedef std::unordered_map<Key, T, Hash, KeyEqual, Allocator> base_t;
        base_t m_;
    public:
        typedef Key key_type;
        typedef T mapped_type;
        typedef std::pair<const Key, T> value_type;
        typedef std::size_t size_type;
        template<class It> struct it_wrap : public It {
            it_wrap() {}
            it_wrap(const It &i) : It(i) {}
            const Key &key() const { return (*this)->first; }
            auto &value() const { return (*this)->second; }
        };
        typedef it_wrap<typename base_t::iterator> iterator;
        typedef it_wrap<typename base_t::const_iterator> const_iterator;
        robin_map() {}
        explicit robin_map(size_type n) : m_(n) {}
        iterator begin() { return m_.begin(); }
        iterator end() { return m_.end(); }
        const_iterator begin() const { return m_.begin(); }
        const_iterator end() const { return m_.end(); }
        const_iterator cbegin() const { return m_.cbegin(); }
        const_iterator cend() const { return m_.cend(); }
        iterator find(const Key &k) { return m_.find(k); }
        const_iterator find(const Key &k) const { return m_.find(k); }
        size_type count(const Key &k) const { return m_.count(k); }
        bool contains(const Key &k) const { return m_.find(k) != m_.end(); }
        T &operator[](const Key &k) { return m_[k]; }
        T &at(const Key &k) { return m_.at(k); }
        const T &at(const Key &k) const { return m_.at(k); }
        template<class... A> std::pair<iterator, bool> emplace(A &&... a) { auto r = m_.emplace(std::forward<A>(a)...); return {iterator(r.first), r.second}; }
        std::pair<iterator, bool> insert(const value_type &v) { auto r = m_.insert(v); return {iterator(r.first), r.second}; }
        size_type erase(const Key &k) { return m_.erase(k); }
        iterator erase(const_iterator it) { return m_.erase(static_cast<const typename base_t::const_iterator &>(it)); }
        void clear() { m_.clear(); }
        void reserve(size_type n) { m_.reserve(n); }
        void rehash(size_type n) { m_.rehash(n); }
        size_type size() const { return m_.size(); }
        bool empty() const { return m_.empty(); }
    };
}
#endif
