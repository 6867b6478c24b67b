// Host build of flowgger_amd/csrc/fg_fuse.hpp (framing inside the decode kernels): walks a raw stream tile by tile the way the fused
// kernels do -- stage [T * S - 16, T * S + S + look), per 16-byte chunk a delimiter | UTF-8 error word, count, list, lines, the tail that
// runs past the staged range -- with the wave-level steps on the fiber emulation of a wavefront (tests/native/fg_wave_emu.hpp), and
// hands back the frames it found: start, end (terminator included), UTF-8 verdict.  tests/test_fuse_cpu.py compares them with the
// oracle's restatement of the splitters (fgo_frame).  Test infrastructure only.
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../flowgger_amd/csrc/fg_fuse.hpp"

namespace {
using namespace fg;
std::string g_err;
}  // namespace

extern "C" const char* fgf_last_error() { return g_err.c_str(); }

extern "C" uint32_t fgf_plan(uint64_t avg_len, uint32_t lines, uint32_t tile_cap, uint32_t* look) {
    const fuse::TilePlan p = fuse::plan_tile(avg_len, lines, tile_cap);
    *look = p.look;
    return p.S;
}

// Frames of bytes[0 .. nbytes) (readable up to nbytes rounded up to 16; what lies behind nbytes is garbage on purpose), delimiter
// `delim`, tiles of S bytes with `look` bytes of look-ahead, at most `lines` lines per pass.  starts / ends / bad: cap entries.
// Returns the number of frames (may exceed cap: then only cap were written), -1 on an error.  *consumed = the bytes the frames cover.
// *passes = stage-B passes the kernels would have run; *scans = tiles whose last line needed the forward scan.
extern "C" long fgf_frame(const uint8_t* bytes, uint64_t nbytes, uint32_t delim, int final, uint32_t S, uint32_t look, uint32_t tile_cap,
                          uint32_t lines, uint64_t* starts, uint64_t* ends, uint8_t* bad, uint64_t cap, uint64_t* consumed, uint64_t* passes,
                          uint64_t* scans) {
    try {
        if (S == 0 || S % 16 || look % 16 || fuse::kPre + S + look + 16u > tile_cap || lines < 1 || lines > 64) throw std::runtime_error("bad geometry");
        const uint64_t padded = (nbytes + 15u) & ~15ull;
        const uint32_t delim4 = delim * 0x01010101u;
        std::vector<uint32_t> tile32(tile_cap / 4 + 16);
        uint8_t* tile = reinterpret_cast<uint8_t*>(tile32.data());
        std::vector<uint64_t> lds64(fuse::lds_bytes(tile_cap) / 8 + 2);
        const fuse::Lds L = fuse::carve(reinterpret_cast<uint8_t*>(lds64.data()), tile_cap);
        const uint64_t ntiles = (nbytes + S - 1) / S;
        uint64_t n = 0, last_end = 0;
        *passes = 0;
        *scans = 0;
        // 16 bytes at stream position pos (a multiple of 16): zeros behind the readable range (the buffer descriptor's bounds check)
        auto ld = [&](uint64_t pos) -> fuse::U4 {
            fuse::U4 q{0, 0, 0, 0};
            if (pos + 16u <= padded) memcpy(&q, bytes + pos, 16);
            return q;
        };
        uint32_t prev_chunks = 64u * L.R;
        for (uint64_t T = 0; T < ntiles; ++T) {
            const fuse::Geo g = fuse::tile_geo(T, S, look, nbytes);
            if (g.span > tile_cap) throw std::runtime_error("span exceeds the tile");
            const uint32_t nchunk = g.span / 16u;
            // ---- stage A: bytes -> tile, masks -> dm16 (the kernels do this from the register window) ----
            uint32_t pw = 0, err_acc = 0;
            for (uint32_t c = 0; c < nchunk; ++c) {
                fuse::U4 q{0, 0, 0, 0};
                if (!(T == 0 && c == 0)) q = ld(g.base + (uint64_t)c * 16u);
                memcpy(tile + c * 16u, &q, 16);
                uint32_t m = fuse::chunk_masks(q.x, q.y, q.z, q.w, pw, delim4, fuse::chunk_rem(g, c));
                if (c == 0) m = fuse::pre_chunk_mask(m, T == 0);
                L.dm16[c] = (uint16_t)m;
                err_acc |= m >> 16;
                pw = q.w;
            }
            for (uint32_t c = nchunk; c < prev_chunks; ++c) L.dm16[c] = 0u;  // (stale masks of a longer tile before)
            prev_chunks = nchunk;
            const uint32_t pw_last = pw;
            const bool any_err = err_acc != 0u;
            // ---- the wave-level part ----
            struct Out { uint64_t s, e; bool bad, valid; };
            std::vector<Out> out;
            Out slot[64];
            uint64_t tile_passes = 0, tile_scans = 0;
            emu::run_wave([&]() {
                const uint32_t lane = wv::lane();
                wv::sync();
                fuse::Count cn = fuse::count_tile(L, g);
                uint64_t tail_end;
                bool tail_bad;
                const bool need_scan = cn.n_own != 0u && cn.n_all == cn.n_own && g.end_x == fuse::kUnresolved;
                fuse::resolve_tail(g, &cn, ld, nbytes, delim4, final != 0, pw_last, &tail_end, &tail_bad);
                if (lane == 0 && need_scan) ++tile_scans;
                for (uint32_t w0 = 0; w0 < cn.n_own; w0 += fuse::kList) {
                    wv::sync();
                    fuse::build_list(L, cn, w0);
                    wv::sync();
                    for (uint32_t p0 = w0; p0 < cn.n_own && p0 < w0 + fuse::kList; p0 += lines) {
                        const uint32_t k = p0 + lane;
                        const bool valid = lane < lines && k < cn.n_own && k < w0 + fuse::kList;
                        uint64_t o0 = 0, o1 = 0;
                        bool b = false;
                        if (valid) {
                            uint32_t s, e;
                            fuse::line_at(L, cn, w0, k, &s, &e);
                            o0 = g.base + s;
                            if (e == fuse::kUnresolved) {
                                o1 = tail_end;
                                const uint32_t lim = g.end_x != fuse::kUnresolved ? g.end_x + 1u : g.span;
                                b = tail_bad || (any_err && fuse::line_bad(tile32.data(), g, s, lim));
                            } else {
                                o1 = g.base + e;
                                b = any_err && fuse::line_bad(tile32.data(), g, s, e);
                            }
                        }
                        slot[lane] = Out{o0, o1, b, valid};
                        wv::sync();
                        if (lane == 0) {
                            ++tile_passes;
                            for (uint32_t l = 0; l < 64u; ++l)
                                if (slot[l].valid) out.push_back(slot[l]);
                        }
                        wv::sync();
                    }
                }
            });
            *passes += tile_passes;
            *scans += tile_scans;
            for (const Out& o : out) {
                if (n && o.s != last_end) throw std::runtime_error("frames are not contiguous at frame " + std::to_string(n));
                if (n == 0 && o.s != 0) throw std::runtime_error("the first frame does not start at 0");
                if (n < cap) {
                    starts[n] = o.s;
                    ends[n] = o.e;
                    bad[n] = o.bad ? 1 : 0;
                }
                last_end = o.e;
                ++n;
            }
        }
        *consumed = last_end;
        return (long)n;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}
