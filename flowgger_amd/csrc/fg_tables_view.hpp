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
};
enum { S_HOST = 0, S_APP = 1, S_PROC = 2, S_MSGID = 3, S_MSG = 4, S_FULL = 5 };

}  // namespace fg
