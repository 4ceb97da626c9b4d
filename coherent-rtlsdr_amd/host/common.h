// common.h -- the small host utilities of the reference's include/common.h that the hot path
// touches: lvector (mutex-guarded vector, include/common.h:182-246) and sync_threshold (:32).
// cbuffer / barrier / lqueue are USB-ring plumbing and stay in the reference.
#ifndef COMMONH
#define COMMONH
#include <mutex>
#include <vector>

const float sync_threshold = 0.005f; // include/common.h:32

template <class T>
class lvector {
    std::vector<T> v;
public:
    std::mutex m;
    void push_back(T x) { std::lock_guard<std::mutex> l(m); v.push_back(x); }
    size_t size() { std::lock_guard<std::mutex> l(m); return v.size(); }
    T &operator[](size_t i) { return v[i]; }
    typename std::vector<T>::iterator begin() { return v.begin(); }
    typename std::vector<T>::iterator end() { return v.end(); }
    void lock() { m.lock(); }
    void unlock() { m.unlock(); }
};
#endif
