// Host build of flowgger_amd/csrc/fg_sd2.hpp (the pair-parallel structured-data walk of the RFC5424 kernel): stages groups of lines
// into a tile the way the streaming pipeline does (consecutive lines from a 16-byte boundary, garbage behind the staged bytes), runs
// classify_tile / group_walk / group_emit on the fiber emulation of a wavefront (tests/native/fg_wave_emu.hpp) and hands back what the
// kernel derives from them, next to the reference's state machine run byte by byte over the same lines.  Test infrastructure only.
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../flowgger_amd/csrc/fg_sd2.hpp"

namespace {
using namespace fg;
std::string g_err;

// the reference's state machine, byte by byte (rfc5424_decoder.rs:134-158, :174-242) -- the independent check
struct Ent { uint32_t name_s, name_len, val_s, val_len, esc, sdid; };
uint32_t byte_walk(const uint8_t* ln, uint32_t pos, uint32_t len, uint32_t* msg_at, std::vector<Ent>& out) {
    for (;;) {
        uint32_t s = pos + 1, sp = s;
        while (sp < len && ln[sp] != ' ') ++sp;
        if (sp >= len) return sd2::E_NOSD;
        out.push_back({s, sp - s, 0, 0, 0, 1});
        uint32_t st = 0, name_s = 0, name_e = 0, val_s = 0, esc = 0, after = 0;
        for (uint32_t i = sp + 1; i < len; ++i) {
            const uint32_t c = ln[i];
            if (st == 3) {
                if (c == '\\') { st = 4; esc = 1; }
                else if (c == '"') { out.push_back({name_s, name_e - name_s, val_s, i - val_s, esc, 0}); st = 0; }
            } else if (st == 4) {
                st = 3;
            } else {
                const bool is_name = sd2::is_name_char(c);
                if (st == 0) {
                    if (c == ' ' || c == '"') {}
                    else if (c == ']') { after = i + 1; break; }
                    else if (is_name) { st = 1; name_s = i; }
                    else return sd2::E_SDFMT;
                } else if (st == 1) {
                    if (is_name) {}
                    else if (c == '=') { name_e = i; st = 2; }
                    else return sd2::E_SDFMT;
                } else {
                    if (c != '"') return sd2::E_SDFMT;
                    st = 3; val_s = i + 1; esc = 0;
                }
            }
        }
        if (after == 0) return sd2::E_NOBRACKET;
        if (after >= len) return sd2::E_NOMSG;
        if (ln[after] == '[') { pos = after; continue; }
        if (ln[after] != ' ') return sd2::E_MALFORMED;
        *msg_at = after;
        return sd2::E_OK;
    }
}
}  // namespace

extern "C" const char* fgs2_last_error() { return g_err.c_str(); }

// For every line i = bytes[offsets[i], offsets[i+1]) whose structured data starts at line index sd_pos[i] (0 = the line has none):
//   handled[i], status[i], msg_at[i], n_ent[i] from group_walk; its entries (through group_emit into a real entry table) as six
//   uint32 per entry in ent[6 * (ent_first[i] + k)] = name_s, name_len, val_s, val_len, esc, sdid;
//   ref_* the same from the byte-wise state machine.
// head_cap != 0: HEAD staging -- only the first head_cap bytes of every line are put into the tile (rows packed on 16-byte boundaries).
// Returns the number of entries written, -1 on an error (fgs2_last_error).
extern "C" long fgs2_walk(const uint8_t* bytes, uint64_t nbytes, const uint64_t* offsets, uint64_t n, const uint32_t* sd_pos, uint32_t lines_per_group,
                          uint32_t tile_cap, uint32_t head_cap, uint8_t* handled, uint32_t* status, uint32_t* msg_at, uint32_t* n_ent, uint32_t* ent_first,
                          uint32_t* ent, uint64_t ent_cap, uint32_t* ref_status, uint32_t* ref_msg_at, uint32_t* ref_ent_first, uint32_t* ref_n_ent,
                          uint32_t* ref_ent, uint8_t* tile_bailed) {
    try {
        if (lines_per_group < 1 || lines_per_group > 64 || tile_cap % 1024 != 0) throw std::runtime_error("bad geometry");
        const uint32_t L = lines_per_group;
        const uint32_t stride16 = tile_cap / 16u + 16u;
        const size_t lds_bytes = tile_cap + 64u + (size_t)stride16 * 2u * 2u + sd2::extra_bytes(tile_cap);
        std::vector<uint64_t> lds64(lds_bytes / 8 + 2);
        uint8_t* smem = reinterpret_cast<uint8_t*>(lds64.data());
        uint16_t* bm16 = reinterpret_cast<uint16_t*>(smem + tile_cap + 64u);
        uint8_t* extra = smem + tile_cap + 64u + (size_t)stride16 * 2u * 2u;
        // a real entry table for group_emit
        std::vector<fg_span> e_name(ent_cap);
        std::vector<uint64_t> e_val(ent_cap);
        std::vector<uint8_t> e_type(ent_cap), e_flags(ent_cap);
        DevTables t{};
        t.ent_cap = ent_cap;
        t.ent_name = e_name.data();
        t.ent_val = e_val.data();
        t.ent_type = e_type.data();
        t.ent_flags = e_flags.data();
        uint64_t used = 0, ref_used = 0;
        uint64_t g0 = 0;
        while (g0 < n) {
            // the group: consecutive lines while they fit the tile, at most L
            uint32_t nl = 0;
            uint64_t a0 = offsets[g0] & ~15ull;
            uint32_t tb[64], tl[64], span = 0;
            if (head_cap) {
                uint32_t at = 0;
                while (nl < L && g0 + nl < n) {
                    const uint64_t o0 = offsets[g0 + nl], o1 = offsets[g0 + nl + 1];
                    const uint64_t want = (o1 - o0) + (o0 & 15ull);
                    const uint32_t st = want >= head_cap ? head_cap : (uint32_t)((want + 15ull) & ~15ull);
                    if (at + st > tile_cap) break;
                    tb[nl] = at + (uint32_t)(o0 & 15ull);
                    const uint32_t len = (uint32_t)(o1 - o0), room = st - (uint32_t)(o0 & 15ull);
                    tl[nl] = st == 0u ? 0u : (room < len ? room : len);
                    at += st;
                    ++nl;
                }
                if (nl == 0) nl = 1, tb[0] = 0, tl[0] = 0;
                span = at;
            } else {
                while (nl < L && g0 + nl < n && offsets[g0 + nl + 1] - a0 <= tile_cap) {
                    tb[nl] = (uint32_t)(offsets[g0 + nl] - a0);
                    tl[nl] = (uint32_t)(offsets[g0 + nl + 1] - offsets[g0 + nl]);
                    ++nl;
                }
                if (nl == 0) {  // a single line longer than the tile: not in the tile (the kernel parses it from global memory)
                    nl = 1;
                    tb[0] = 0;
                    tl[0] = 0;
                    span = 0;
                } else {
                    span = (uint32_t)((offsets[g0 + nl] - a0 + 15ull) & ~15ull);
                }
            }
            memset(smem, 0xA5, lds_bytes);  // garbage everywhere: the kernels must not depend on what they did not write
            if (head_cap) {
                for (uint32_t k = 0; k < nl; ++k) {
                    const uint64_t o0 = offsets[g0 + k];
                    const uint32_t row0 = tb[k] - (uint32_t)(o0 & 15ull);
                    const uint64_t b0 = o0 & ~15ull;
                    const uint32_t st = ((tb[k] - row0) + tl[k] + 15u) & ~15u;
                    for (uint32_t i = 0; i < st && row0 + i < tile_cap; ++i) smem[row0 + i] = b0 + i < nbytes ? bytes[b0 + i] : 0;
                }
            } else {
                for (uint32_t i = 0; i < span; ++i) smem[i] = a0 + i < nbytes ? bytes[a0 + i] : 0;
            }
            sd2::Lds lds = sd2::carve(smem, bm16, tile_cap, extra);
            sd2::LineOut outs[64];
            sd2::LineIn ins[64];
            uint32_t firsts[64];
            bool bailed = false;
            emu::run_wave([&]() {
                const uint32_t lane = wv::lane();
                sd2::LineIn in{false, 0u, 0u, 0u, false};
                if (lane < nl) {
                    const uint64_t li = g0 + lane;
                    const uint32_t len = (uint32_t)(offsets[li + 1] - offsets[li]);
                    in.base = tb[lane];
                    in.d0 = sd_pos[li];
                    in.wlen = tl[lane];
                    in.whole = tl[lane] == len;
                    in.sd = sd_pos[li] != 0u && tl[lane] > sd_pos[li] && (head_cap ? true : in.whole);
                }
                const bool chain = span ? sd2::classify_tile(lds, span) : true;
                wv::sync();
                sd2::LineOut o{false, 0u, false, 0u, 0u, 0u, 0u};
                if (!chain) o = sd2::group_walk(lds, span, in);
                if (lane == 0u) bailed = chain;
                // slots: a plain wave prefix sum over the handled Ok lines
                uint32_t total = 0;
                const uint32_t mine = (o.handled && o.status == sd2::E_OK) ? o.n_ent : 0u;
                const uint32_t ex = wv::excl_sum(mine, &total);
                const uint32_t first = (uint32_t)used + ex;
                if (!chain) sd2::group_emit(lds, t, in, o, used + total <= ent_cap, first);
                outs[lane] = o;
                ins[lane] = in;
                firsts[lane] = first;
                wv::sync();
                if (lane == 0u) used += total;
            });
            if (used > ent_cap) throw std::runtime_error("entry table too small");
            for (uint32_t k = 0; k < nl; ++k) {
                const uint64_t li = g0 + k;
                const sd2::LineOut& o = outs[k];
                tile_bailed[li] = bailed ? 1 : 0;
                handled[li] = o.handled ? 1 : 0;
                status[li] = o.status;
                msg_at[li] = o.msg_at;
                n_ent[li] = (o.handled && o.status == sd2::E_OK) ? o.n_ent : 0u;
                ent_first[li] = firsts[k];
                if (o.handled && o.status == sd2::E_OK)
                    for (uint32_t q = 0; q < o.n_ent; ++q) {
                        const uint32_t sl = firsts[k] + q;
                        uint32_t* d = ent + 6ull * sl;
                        const bool sdid = e_type[sl] == FG_T_SDID;
                        d[0] = e_name[sl].off;
                        d[1] = e_name[sl].len;
                        d[2] = sdid ? 0u : (uint32_t)e_val[sl];
                        d[3] = sdid ? 0u : (uint32_t)(e_val[sl] >> 32);
                        d[4] = (e_flags[sl] & FG_EF_VAL_ESC) ? 1u : 0u;
                        d[5] = sdid ? 1u : 0u;
                    }
                // the reference
                std::vector<Ent> ref;
                uint32_t rmsg = 0, rst = sd2::E_OK;
                const uint32_t len = (uint32_t)(offsets[li + 1] - offsets[li]);
                if (sd_pos[li]) rst = byte_walk(bytes + offsets[li], sd_pos[li], len, &rmsg, ref);
                ref_status[li] = rst;
                ref_msg_at[li] = rmsg;
                ref_ent_first[li] = (uint32_t)ref_used;
                ref_n_ent[li] = rst == sd2::E_OK ? (uint32_t)ref.size() : 0u;
                if (rst == sd2::E_OK)
                    for (const Ent& e2 : ref) {
                        uint32_t* d = ref_ent + 6ull * ref_used++;
                        d[0] = e2.name_s; d[1] = e2.name_len; d[2] = e2.val_s; d[3] = e2.val_len; d[4] = e2.esc; d[5] = e2.sdid;
                    }
            }
            g0 += nl;
        }
        return (long)used;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}
