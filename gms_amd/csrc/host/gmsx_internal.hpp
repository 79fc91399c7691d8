// Internal declarations shared by the host substrate and the device layer of libgmsx.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <ios>
#include <memory>
#include <new>
#include <stdexcept>

#include "gmsx.h"

#include <cstddef>
namespace gmsx {

// No exception crosses the C ABI: every exported function that can reach an allocation of the C++ library (containers, strings, streams,
// rocPRIM's host side) runs its body under this guard and returns a status code instead — the reference's convention is exit codes, never
// unwinding through a caller (gapbs/reader.h:45,228: std::exit(-2) on a failed open), and a C / cgo / ctypes caller cannot catch anything.
template <class F>
inline int guard(F &&body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return GMSX_ERR_NOMEM;
    } catch (const std::length_error &) {
        return GMSX_ERR_NOMEM;
    } catch (const std::ios_base::failure &) {
        return GMSX_ERR_IO;
    } catch (...) {
        return GMSX_ERR_KERNEL;
    }
}

// Host CSR with 64-bit offsets and 32-bit ids (the reference keeps row POINTERS, gapbs/graph.h:361-364;
// offsets are what the device wants and what the .sg file stores).
// delete[] for an array the loader allocated; an array that IS a read-only file mapping (the aligned ".sgx" cache, loader.cpp read_sgx) unmaps instead
struct ArrayFree {
    void *map = nullptr;   // base of the mapping the array lies in (nullptr: a new[] array)
    size_t map_bytes = 0;
    template <class T>
    void operator()(T *p) const {
        if (map)
            unmap(map, map_bytes);
        else
            delete[] p;
    }
    static void unmap(void *base, size_t bytes);  // munmap (loader.cpp)
};
struct Csr {
    int64_t n = 0;
    int64_t nnz = 0;
    std::unique_ptr<int64_t[], ArrayFree> off;    // n + 1
    std::unique_ptr<int32_t[], ArrayFree> neigh;  // nnz
    bool directed = false;
    bool mapped = false;  // off / neigh are private read-only mappings of a cache file: every process that loads it shares ONE copy in the page cache
};

bool worth_relabelling(const Csr &g);
int relabel_by_degree(const Csr &g, Csr &out);

// OPTIONS (gmsx_set_option, include/gmsx.h): the library takes its tuning limits and diagnostics from explicit calls, never from the
// environment.  opt("BK_MAXC") = the value the caller set as a C string (valid until the option is set again), or nullptr when unset —
// the call sites parse it the way they used to parse an environment variable.  Every option leaves the results unchanged.
const char *opt(const char *name);
inline long long opt_int(const char *name, long long dflt) {
    const char *e = opt(name);
    return e ? std::atoll(e) : dflt;
}
inline bool opt_on(const char *name) {
    const char *e = opt(name);
    return e && std::atoi(e) != 0;
}

// threads of the host substrate (gmsx_set_host_threads / OpenMP's current maximum): what the staged upload copies with (loader.cpp)
int host_threads();
// memcpy by the threads of the host substrate (OpenMP's pool: no thread is started per call)
void parallel_memcpy(void *dst, const void *src, size_t bytes);
}  // namespace gmsx

struct gmsx_csr {
    gmsx::Csr g;
    bool relabelled = false;
};
