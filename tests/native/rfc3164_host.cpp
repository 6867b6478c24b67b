// Host build of flowgger_amd/csrc/fg_rfc3164_parse.hpp (the per-line RFC3164 decode the kernel runs): fills table
// rows for a packed batch so that the product's own fg_tables_serialize can turn them into canonical Records for the
// comparison with the oracle.  Test infrastructure only -- the product has no CPU decode path.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fg_hip.h"
#include "../../flowgger_amd/csrc/fg_rfc3164_parse.hpp"
#include "../../flowgger_amd/csrc/fg_tz_index.hpp"

namespace {
struct HostReader {
    const uint8_t* p;
    uint32_t byte(uint32_t i) { return p[i]; }
    uint32_t load4(uint32_t i, uint32_t nb) {  // only the nb wanted bytes are read (the private copy has no padding)
        uint32_t w = 0;
        memcpy(&w, p + i, nb);
        return w;
    }
    void load16(uint32_t i, uint32_t* q) { memcpy(q, p + i, 16); }  // (only called with sixteen wanted bytes)
};
}  // namespace

extern "C" int fg3_decode_batch(const uint8_t* bytes, const uint64_t* offsets, uint64_t n, int32_t current_year, uint32_t n_zones,
                                const char* const* names, const uint32_t* zone_first, const int64_t* utc_start,
                                const int32_t* utc_offset, fg_tables* t) {
    // the same index the product uploads to the GPU (fg_tz_index.hpp), bound to host memory
    std::vector<std::string> nm(names, names + n_zones);
    const uint32_t ne = n_zones ? zone_first[n_zones] : 0u;
    fg::r3164::TzIndex idx;
    if (!idx.build(nm, std::vector<uint32_t>(zone_first, zone_first + (n_zones ? n_zones + 1 : 0)),
                   std::vector<int64_t>(utc_start, utc_start + ne), std::vector<int32_t>(utc_offset, utc_offset + ne), current_year))
        return -1;
    fg::r3164::Cfg cfg{};
    cfg.current_year = current_year;
    cfg.tz = idx.view(idx.blob.data());
    const fg_span none{0u, FG_NONE};
    for (uint64_t i = 0; i < n; ++i) {
        // an exact-size private copy of the line: reads outside [0, len) would be caught by ASan-style tooling and
        // cannot accidentally see the neighbouring line
        std::vector<uint8_t> line(bytes + offsets[i], bytes + offsets[i + 1]);
        line.push_back(0xFF);
        HostReader rd{line.data()};
        fg::r3164::Row r;
        fg::r3164::parse_line(rd, (uint32_t)(offsets[i + 1] - offsets[i]), cfg, r);
        const bool ok = r.status == fg::r3164::ST_OK;
        t->meta[i] = r.status | (ok ? r.fac : 0xFFu) << 8 | (ok ? r.sev : 0xFFu) << 16 | (ok && r.msg_join ? (uint32_t)FG_F_MSG_JOIN : 0u) << 24;
        t->ts[i] = ok ? r.ts : 0.0;
        t->hostname[i] = ok ? fg_span{r.host_off, r.host_len} : none;
        t->appname[i] = none;
        t->procid[i] = none;
        t->msgid[i] = none;
        t->msg[i] = ok ? fg_span{r.msg_off, r.msg_len} : none;
        t->full_msg[i] = ok ? fg_span{0u, r.full_len} : none;
        t->ent_first[i] = 0;
        t->ent_count[i] = 0;
    }
    return 0;
}
