// fg_tables_view.hpp -- the kernels' view of fg_tables (same arrays as include/fg_hip.h, as a plain struct
// of pointers).  No HIP dependency: the encoder emitters (fg_emit.hpp) are also compiled for the host by the
// CPU tests.
#pragma once
#include <stdint.h>

#include "../../include/fg_hip.h"

namespace fg {

// Device view of fg_tables (same arrays, device pointers).
struct DevTables {
    uint64_t n;
    uint64_t ent_cap;
    uint32_t* meta;
    double* ts;
    fg_span* span[6];  // hostname, appname, procid, msgid, msg, full_msg
    uint32_t* ent_first;
    uint32_t* ent_count;
    fg_span* ent_name;
    uint64_t* ent_val;
    uint8_t* ent_type;
    uint8_t* ent_flags;
    unsigned long long* ent_used;
    // hand-over between the two kernels of a decode (fast form -> exact general form, fg_gelf.hip / fg_rfc5424.hip): the first
    // kernel stores `epoch` here when it leaves lines for the second one, which returns at once when the word holds anything
    // else.  One word of a ctx-owned ring per launch (epoch = the ctx's launch counter), so launches in flight never share one.
    uint32_t* pending;
    uint32_t epoch;
    // entry slots a wave reserves from ent_used at a time (wv::wave_alloc): 0 = exactly what each request needs.  Set by the kernel
    // launchers from the table, the grid and the lines of THIS launch (fg::entry_chunk, fg_pipeline.hpp).
    uint32_t alloc_chunk;
    // launches that reserve entry slots from this table (host side only: the sliced host paths decode one batch as many launches;
    // fg::entry_chunk divides the table's budget by it).  0 / 1 = this launch alone.
    uint32_t shares;
};
// Dynamic chunk dispatch of the streaming decoders (fg_pipeline.hpp persistent_loop): the ticket counter of ONE launch -- a word of
// a ctx-owned ring in device memory -- and the host's copy of what it holds; the launcher hands the kernel the word + that value and
// adds the tickets the launch will draw (exactly its number of chunks), so the word is never reset between launches.
struct TicketSlot {
    uint32_t* d_word;
    uint32_t* h_val;
};
// A table pointer that did not come straight out of the kernel arguments (a kernel that keeps its DevTables in LDS, k_gelf) is a
// generic pointer to the compiler: stores through it would be flat_ instructions, which also count against the LDS wait counter.
// gstore() says what every table pointer is -- global memory.  (No effect where the compiler knows already.)
template <class T> struct SameAs { typedef T type; };  // (a non-deduced context: the pointer alone names the element type)
template <int N> struct RawOf;
template <> struct RawOf<1> { typedef uint8_t type; };
template <> struct RawOf<4> { typedef uint32_t type; };
template <> struct RawOf<8> { typedef uint64_t type; };
#if defined(__HIPCC__)
#define FG_TV_HD __host__ __device__ __forceinline__
#else
#define FG_TV_HD inline
#endif
template <class T>
FG_TV_HD void gstore(T* p, uint64_t i, const typename SameAs<T>::type& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef typename RawOf<sizeof(T)>::type U;
    U u;
    __builtin_memcpy(&u, &v, sizeof(T));
    ((U __attribute__((address_space(1)))*)p)[i] = u;
#else
    p[i] = v;
#endif
}
template <class T>
FG_TV_HD T* glb(T* p) {  // (atomics: the builtin takes a generic pointer; the cast chain is what the optimiser sees)
#if defined(__HIP_DEVICE_COMPILE__)
    return (T*)(T __attribute__((address_space(1)))*)p;
#else
    return p;
#endif
}
enum { S_HOST = 0, S_APP = 1, S_PROC = 2, S_MSGID = 3, S_MSG = 4, S_FULL = 5 };

}  // namespace fg
