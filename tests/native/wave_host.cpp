// wave_host.cpp -- the wave-cooperative tokenisers (flowgger_amd/csrc/fg_*2.hpp) compiled for the CPU over the fiber
// emulation of a wavefront (fg_wave_emu.hpp): the CPU suite runs the kernels' own code, lane for lane, against the oracle.
// The driver below stands in for the streaming pipeline (fg_pipeline.hpp): group geometry, tile staging, stage-A class
// bitmaps, one decode_tile per group, table rows.  Test infrastructure; never part of libfg_hip.so.
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../flowgger_amd/csrc/fg_gelf2.hpp"

namespace {
std::string g_err;

fg::DevTables to_dev(const fg_tables& t) {
    fg::DevTables d;
    d.n = t.n;
    d.ent_cap = t.ent_cap;
    d.meta = t.meta;
    d.ts = t.ts;
    d.span[0] = t.hostname; d.span[1] = t.appname; d.span[2] = t.procid; d.span[3] = t.msgid; d.span[4] = t.msg; d.span[5] = t.full_msg;
    d.ent_first = t.ent_first;
    d.ent_count = t.ent_count;
    d.ent_name = t.ent_name;
    d.ent_val = t.ent_val;
    d.ent_type = t.ent_type;
    d.ent_flags = t.ent_flags;
    d.ent_used = (unsigned long long*)t.ent_used;
    return d;
}
}  // namespace

extern "C" const char* fgw_last_error() { return g_err.c_str(); }

// GELF: decode n framed lines; rows of lines the fast form handled are written to `t` (host arrays), handled[i] = 1;
// other rows are left untouched (handled[i] = 0: on the GPU they take the general form).  Returns 0, or -1 (fgw_last_error).
// strip = FG_FRAME_NONE: offsets delimit bare lines.  FG_FRAME_LINE / FG_FRAME_NUL: frames of a raw stream INCLUDING their
// terminators (what fg_frame_device produces): stripped per lane as the pipeline does, and stage A is told the terminator byte.
extern "C" int fgw_gelf_decode_framed(const uint8_t* bytes, uint64_t nbytes, const uint64_t* offsets, uint64_t n, const fg_tables* tables,
                                      uint32_t lines_per_group, uint32_t tile_cap, uint8_t* handled, uint32_t strip) {
    using namespace fg;
    try {
        const uint32_t term4 = strip == FG_FRAME_LINE ? 0x0A0A0A0Au : strip == FG_FRAME_NUL ? 0u : wv::kNoTerm;
        if (lines_per_group < 1 || lines_per_group > 64 || tile_cap % 1024 != 0 || tile_cap > 57344) throw std::runtime_error("bad geometry");
        const DevTables t = to_dev(*tables);
        *t.ent_used = 0;
        const uint32_t stride16 = tile_cap / 16u + 16u;
        const size_t lds_bytes = tile_cap + 64u + (size_t)stride16 * 2u * gelf2::kClasses + gelf2::extra_bytes(tile_cap, lines_per_group);
        std::vector<uint64_t> lds64(lds_bytes / 8 + 2);
        uint8_t* smem = reinterpret_cast<uint8_t*>(lds64.data());
        uint16_t* bm16 = reinterpret_cast<uint16_t*>(smem + tile_cap + 64u);
        uint8_t* extra = smem + tile_cap + 64u + (size_t)stride16 * 2u * gelf2::kClasses;
        const uint32_t L = lines_per_group;
        uint32_t ent_state[2] = {0u, 0u};
        const uint32_t alloc_chunk = 96u;  // small and odd-sized on purpose: chunk boundaries fall inside the test corpora
        for (uint64_t g0 = 0; g0 < n; g0 += L) {
            const uint32_t nl = (uint32_t)(g0 + L <= n ? L : n - g0);
            const uint64_t a0 = offsets[g0] & ~15ull;
            const uint64_t want = offsets[g0 + nl] - a0;
            const uint32_t span = want > tile_cap ? tile_cap : (uint32_t)((want + 15ull) & ~15ull);
            // stage the tile: garbage beyond the span on purpose (the kernels must not depend on it)
            gelf2::Lds lds = gelf2::carve(smem, bm16, tile_cap, extra, lines_per_group);
            // (everything but the dirty bits is wiped: they persist from tile to tile -- decode_tile clears what it used)
            std::vector<uint8_t> keep(gelf2::dirty_bytes(tile_cap));
            if (g0 == 0) memset(keep.data(), 0, keep.size());
            else memcpy(keep.data(), lds.dirty, keep.size());
            memset(smem, 0xA5, lds_bytes);
            memcpy(lds.dirty, keep.data(), keep.size());
            for (uint32_t i = 0; i < span; ++i) smem[i] = a0 + i < nbytes ? bytes[a0 + i] : 0;
            for (uint32_t c = 0; c < span / 16u; ++c) {
                uint32_t x[4], m[gelf2::kClasses + 1];
                memcpy(x, smem + 16u * c, 16);
                gelf2::classify(x[0], x[1], x[2], x[3], m, term4);
                for (uint32_t k = 0; k < gelf2::kClasses; ++k) bm16[k * stride16 + c] = (uint16_t)m[k];
                if (m[gelf2::kClasses]) lds.dirty[c >> 7] |= 1u << ((c >> 2) & 31u);
            }
            gelf2::LineOut outs[64];
            lds.ent_state = ent_state;  // persists across tiles, like the pipeline's two LDS words
            lds.alloc_chunk = alloc_chunk;
            emu::run_wave([&]() {
                gelf2::init_lds(lds);  // (the staging above wiped the LDS; the kernel does this once per wave)
                const uint32_t lane = wv::lane();
                const bool has = lane < nl;
                const uint64_t o0 = has ? offsets[g0 + lane] : 0;
                uint64_t o1 = has ? offsets[g0 + lane + 1] : 0;
                const bool in_tile = has && (o1 - a0) <= (uint64_t)span;
                if (has && strip != FG_FRAME_NONE && o1 > o0) {  // terminator stripping (fg_pipeline.hpp stage B)
                    const uint32_t b1 = bytes[o1 - 1];
                    if (strip == FG_FRAME_LINE) {
                        if (b1 == '\n') {
                            --o1;
                            if (o1 > o0 && bytes[o1 - 1] == '\r') --o1;
                        }
                    } else if (b1 == 0u) {
                        --o1;
                    }
                }
                const gelf2::LineOut o = gelf2::decode_tile(lds, span, in_tile, (uint32_t)(o0 - a0), (uint32_t)(o1 - o0), t);
                outs[lane] = o;
            });
            for (uint32_t k = 0; k < nl; ++k) {
                const gelf2::LineOut& f = outs[k];
                const uint64_t li = g0 + k;
                handled[li] = f.handled ? 1 : 0;
                if (!f.handled) continue;
                const bool ok = f.status == gelf2::G_OK;
                const fg_span none{0, FG_NONE};
                t.meta[li] = f.status | (0xFFu << 8) | ((ok ? f.severity : 0xFFu) << 16) | ((ok ? f.flags : 0u) << 24);
                t.ts[li] = (ok && f.have_ts) ? f.ts : 0.0;
                t.span[0][li] = ok ? fg_span{f.host_off, f.host_len} : none;
                t.span[1][li] = none;
                t.span[2][li] = none;
                t.span[3][li] = none;
                t.span[4][li] = ok ? fg_span{f.msg_off, f.msg_len} : none;
                t.span[5][li] = ok ? fg_span{f.full_off, f.full_len} : none;
                t.ent_first[li] = f.first;
                t.ent_count[li] = f.n_ent;
            }
        }
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

extern "C" int fgw_gelf_decode(const uint8_t* bytes, uint64_t nbytes, const uint64_t* offsets, uint64_t n, const fg_tables* tables,
                               uint32_t lines_per_group, uint32_t tile_cap, uint8_t* handled) {
    return fgw_gelf_decode_framed(bytes, nbytes, offsets, n, tables, lines_per_group, tile_cap, handled, FG_FRAME_NONE);
}

// the register-resident number parser of the GELF fast form on ONE token (unit test hook): returns 1 = parsed (kind / bits as
// serde_json 0.8 would give), 0 = "not the everyday shape" (the kernel then runs the byte-wise json_number)
extern "C" int fgw_parse_num24(const uint8_t* tok, uint32_t n, uint32_t* kind, uint64_t* bits) {
    using namespace fg;
    if (n == 0 || n > 24) return 0;
    uint32_t w[6] = {0, 0, 0, 0, 0, 0};
    uint8_t buf[24];
    memset(buf, 0xA5, sizeof(buf));  // garbage behind the token, as in the tile
    memcpy(buf, tok, n);
    memcpy(w, buf, 24);
    static double p10[23];
    static uint32_t dw[28];
    static bool init = false;
    if (!init) {
        std::vector<uint64_t> lds(4096);
        gelf2::Lds L = gelf2::carve(reinterpret_cast<uint8_t*>(lds.data()), reinterpret_cast<uint16_t*>(lds.data() + 1024), 4096,
                                    reinterpret_cast<uint8_t*>(lds.data() + 2048), 4);
        emu::run_wave([&]() { gelf2::init_lds(L); });
        memcpy(p10, L.p10, sizeof(p10));
        memcpy(dw, L.dw, sizeof(dw));
        init = true;
    }
    return gelf2::parse_num24(w, n, p10, dw, kind, bits) ? 1 : 0;
}

