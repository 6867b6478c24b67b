// fg_decoder.hpp -- C++ host-side mirror of flowgger's Decoder trait, Record model and line
// splitters on top of the C ABI (include/fg_hip.h).  Header-only; link with -lfg_hip -lamdhip64.
//
// Reference interface mirrored (paths relative to the flowgger source tree):
//   trait Decoder { fn decode(&self, &str) -> Result<Record,&'static str> }   decoder/mod.rs:44-46
//   CloneBoxedDecoder::clone_boxed                                            decoder/mod.rs:23-36
//   Record / StructuredData / SDValue                                         record.rs:3-82
//   LineSplitter / NulSplitter / SyslenSplitter ::run                         splitter/*_splitter.rs
// New: decode_batch() -- N framed lines, ONE call into the gfx950 kernels.  There is no CPU decode
// path in here: construction throws std::runtime_error when no gfx950 device is usable.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <istream>
#include <map>
#include <memory>
#include <optional>
#include <ostream>
#include <stdexcept>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

#include "../../include/fg_hip.h"

namespace fg {

// The batch buffers of the framers live in PAGE-LOCKED memory (fg_alloc_pinned): with bytes and offsets there, fg_decode_batch is
// zero-copy -- one launch, the kernels read the lines over the link and write the table columns into the ctx's pinned tables
// (include/fg_hip.h) -- and the raw-stream / transcode uploads run at link speed instead of through the runtime's staging buffer.
// A std::vector with this allocator is all it takes; pinning costs milliseconds per allocation, so the framers reserve their
// capacity once.
template <class T>
struct PinnedAllocator {
    using value_type = T;
    PinnedAllocator() = default;
    template <class U>
    PinnedAllocator(const PinnedAllocator<U>&) {}
    T* allocate(size_t n) {
        void* p = nullptr;
        if (fg_alloc_pinned((uint64_t)(n * sizeof(T)), &p) != FG_OK || !p) throw std::bad_alloc();
        return static_cast<T*>(p);
    }
    void deallocate(T* p, size_t) { fg_free_pinned(p); }
    template <class U>
    bool operator==(const PinnedAllocator<U>&) const { return true; }
    template <class U>
    bool operator!=(const PinnedAllocator<U>&) const { return false; }
};
using PinnedBytes = std::vector<uint8_t, PinnedAllocator<uint8_t>>;
using PinnedOffsets = std::vector<uint64_t, PinnedAllocator<uint64_t>>;

struct SDValue {  // record.rs:3-11
    enum Kind : uint8_t { String = 0, Bool = 1, F64 = 2, I64 = 3, U64 = 4, Null = 5 } kind = Null;
    std::string s;
    uint64_t bits = 0;  // Bool 0/1, F64 IEEE bits, I64 two's complement, U64
};
struct StructuredData {  // record.rs:23-27
    std::optional<std::string> sd_id;
    std::vector<std::pair<std::string, SDValue>> pairs;
};
struct Record {  // record.rs:70-82
    double ts = 0.0;
    bool ts_now = false;  // GELF without "timestamp": caller substitutes the wall clock (gelf_decoder.rs:109)
    std::string hostname;
    std::optional<uint8_t> facility, severity;
    std::optional<std::string> appname, procid, msgid, msg, full_msg;
    std::optional<std::vector<StructuredData>> sd;
};
// Result<Record, &'static str>
struct DecodeResult {
    const char* err = nullptr;  // points into libfg_hip's static error table
    Record record;
    bool ok() const { return err == nullptr; }
};

struct LtsvConfig {  // input.ltsv_schema / input.ltsv_suffixes (ltsv_decoder.rs:24-84)
    std::vector<std::pair<std::string, uint8_t>> schema;  // name -> FG_T_STRING..FG_T_U64
    std::optional<std::string> suffix_bool, suffix_f64, suffix_i64, suffix_u64;
};

namespace detail {
struct Cursor {
    const uint8_t* p;
    uint8_t u8() { return *p++; }
    uint32_t u32() { uint32_t v; memcpy(&v, p, 4); p += 4; return v; }
    uint64_t u64() { uint64_t v; memcpy(&v, p, 8); p += 8; return v; }
    std::string str() { uint32_t n = u32(); std::string s((const char*)p, n); p += n; return s; }
    std::optional<std::string> optstr() { if (!u8()) return std::nullopt; return str(); }
};
// canonical serialisation -> Record
inline DecodeResult from_canonical(const uint8_t* b, fg_format fmt, uint8_t status) {
    DecodeResult r;
    Cursor c{b};
    if (c.u8() == 1) {
        r.err = fg_error_string(fmt, status);
        if (!r.err) r.err = "<libfg_hip: internal status>";
        return r;
    }
    r.record.ts_now = c.u8() != 0;
    uint64_t tb = c.u64();
    memcpy(&r.record.ts, &tb, 8);
    uint8_t fac = c.u8(), sev = c.u8();
    if (fac != 0xFF) r.record.facility = fac;
    if (sev != 0xFF) r.record.severity = sev;
    r.record.hostname = c.optstr().value_or("");
    r.record.appname = c.optstr();
    r.record.procid = c.optstr();
    r.record.msgid = c.optstr();
    r.record.msg = c.optstr();
    r.record.full_msg = c.optstr();
    if (c.u8()) {
        std::vector<StructuredData> v(c.u32());
        for (auto& sd : v) {
            sd.sd_id = c.optstr();
            sd.pairs.resize(c.u32());
            for (auto& kv : sd.pairs) {
                kv.first = c.str();
                kv.second.kind = (SDValue::Kind)c.u8();
                switch (kv.second.kind) {
                    case SDValue::String: kv.second.s = c.str(); break;
                    case SDValue::Bool: kv.second.bits = c.u8(); break;
                    case SDValue::Null: break;
                    default: kv.second.bits = c.u64();
                }
            }
        }
        r.record.sd = std::move(v);
    }
    return r;
}
}  // namespace detail

// Record -> canonical serialisation (used by tests to compare with the oracle byte for byte)
inline std::string to_canonical(const DecodeResult& r) {
    std::string o;
    auto u8 = [&](uint8_t v) { o.push_back((char)v); };
    auto u32 = [&](uint32_t v) { o.append((const char*)&v, 4); };
    auto u64 = [&](uint64_t v) { o.append((const char*)&v, 8); };
    auto str = [&](const std::string& s) { u32((uint32_t)s.size()); o += s; };
    auto opt = [&](const std::optional<std::string>& s) { if (!s) { u8(0); return; } u8(1); str(*s); };
    if (!r.ok()) { u8(1); str(r.err); return o; }
    const Record& x = r.record;
    u8(0); u8(x.ts_now ? 1 : 0);
    uint64_t tb = 0;
    if (!x.ts_now) memcpy(&tb, &x.ts, 8);
    u64(tb);
    u8(x.facility.value_or(0xFF)); u8(x.severity.value_or(0xFF));
    u8(1); str(x.hostname);
    opt(x.appname); opt(x.procid); opt(x.msgid); opt(x.msg); opt(x.full_msg);
    if (!x.sd) { u8(0); return o; }
    u8(1); u32((uint32_t)x.sd->size());
    for (const auto& sd : *x.sd) {
        opt(sd.sd_id);
        u32((uint32_t)sd.pairs.size());
        for (const auto& kv : sd.pairs) {
            str(kv.first); u8(kv.second.kind);
            switch (kv.second.kind) {
                case SDValue::String: str(kv.second.s); break;
                case SDValue::Bool: u8((uint8_t)kv.second.bits); break;
                case SDValue::Null: break;
                default: u64(kv.second.bits);
            }
        }
    }
    return o;
}

class Decoder {
  public:
    virtual ~Decoder() { if (ctx_) fg_destroy(ctx_); }
    Decoder(const Decoder&) = delete;
    Decoder& operator=(const Decoder&) = delete;

    // the reference's trait method
    DecodeResult decode(std::string_view line) const {
        uint64_t offs[2] = {0, line.size()};
        return std::move(decode_batch((const uint8_t*)line.data(), line.size(), offs, 1)[0]);
    }
    // N framed lines packed back to back; line i = bytes[offsets[i] .. offsets[i+1])
    std::vector<DecodeResult> decode_batch(const uint8_t* bytes, uint64_t nbytes, const uint64_t* offsets, uint64_t n) const {
        fg_tables t{};
        int rc = fg_decode_batch(ctx_, fmt_, bytes, nbytes, offsets, n, &t);
        if (rc != FG_OK) throw std::runtime_error("fg_decode_batch failed: " + std::to_string(rc));
        side_effects(fmt_, FG_FRAME_NONE, bytes, offsets, t);
        std::vector<uint64_t> offs(n + 1);
        int64_t total = fg_tables_serialize(fmt_, cfgp(), bytes, offsets, &t, 0, n, nullptr, 0, offs.data());
        if (total < 0) throw std::runtime_error("fg_tables_serialize failed");
        std::vector<uint8_t> blob((size_t)total + 1);
        fg_tables_serialize(fmt_, cfgp(), bytes, offsets, &t, 0, n, blob.data(), (uint64_t)total, offs.data());
        std::vector<DecodeResult> out;
        out.reserve(n);
        for (uint64_t i = 0; i < n; ++i) out.push_back(detail::from_canonical(blob.data() + offs[i], fmt_, FG_META_STATUS(t.meta[i])));
        return out;
    }
    // what the reference decoder writes to the process's stdout while it decodes (SURVEY 8b "Side effects": LTSV's
    // println!("Missing value for name '{}'"), ltsv_decoder.rs:99): reproduced after each batch from the rows' flags
    static void side_effects(fg_format fmt, fg_framing framing, const uint8_t* bytes, const uint64_t* offsets, const fg_tables& t) {
        if (fmt != FG_LTSV || t.n == 0) return;
        const int64_t need = fg_tables_stdout(fmt, framing, bytes, offsets, &t, 0, t.n, nullptr, 0);
        if (need <= 0) return;
        std::string text((size_t)need, '\0');
        fg_tables_stdout(fmt, framing, bytes, offsets, &t, 0, t.n, (uint8_t*)&text[0], (uint64_t)need);
        fwrite(text.data(), 1, text.size(), stdout);
    }
    virtual std::unique_ptr<Decoder> clone_boxed() const = 0;  // decoder/mod.rs:29-36
    fg_ctx* ctx() const { return ctx_; }
    const fg_cfg* cfg_ptr() const { return cfgp(); }
    fg_format format() const { return fmt_; }

  protected:
    Decoder(fg_format fmt, int device, const LtsvConfig* cfg) : fmt_(fmt) {
        if (cfg) set_cfg(*cfg);
        int rc = fg_create(device, cfgp(), &ctx_);
        if (rc != FG_OK) throw std::runtime_error(rc == FG_ERR_NO_DEVICE ? "libfg_hip: no gfx950 GPU (there is no CPU fallback)"
                                                                         : "fg_create failed: " + std::to_string(rc));
    }
    Decoder(const Decoder& o, int /*clone*/) : fmt_(o.fmt_), lcfg_(o.lcfg_) {
        if (o.has_cfg_) set_cfg(lcfg_);
        if (fg_clone(o.ctx_, &ctx_) != FG_OK) throw std::runtime_error("fg_clone failed");
    }
    const fg_cfg* cfgp() const { return has_cfg_ ? &cfg_ : nullptr; }
    void set_cfg(const LtsvConfig& c) {
        lcfg_ = c;
        names_.clear();
        types_.clear();
        for (auto& kv : lcfg_.schema) { names_.push_back(kv.first.c_str()); types_.push_back(kv.second); }
        cfg_.n_schema = (uint32_t)names_.size();
        cfg_.schema_names = names_.data();
        cfg_.schema_types = types_.data();
        cfg_.suffix_bool = lcfg_.suffix_bool ? lcfg_.suffix_bool->c_str() : nullptr;
        cfg_.suffix_f64 = lcfg_.suffix_f64 ? lcfg_.suffix_f64->c_str() : nullptr;
        cfg_.suffix_i64 = lcfg_.suffix_i64 ? lcfg_.suffix_i64->c_str() : nullptr;
        cfg_.suffix_u64 = lcfg_.suffix_u64 ? lcfg_.suffix_u64->c_str() : nullptr;
        has_cfg_ = true;
    }
    fg_format fmt_;
    fg_ctx* ctx_ = nullptr;
    LtsvConfig lcfg_;
    std::vector<const char*> names_;
    std::vector<uint8_t> types_;
    fg_cfg cfg_{};
    bool has_cfg_ = false;
};

class RFC5424Decoder : public Decoder {  // decoder/rfc5424_decoder.rs:8-50
  public:
    explicit RFC5424Decoder(int device = 0) : Decoder(FG_RFC5424, device, nullptr) {}
    std::unique_ptr<Decoder> clone_boxed() const override { return std::unique_ptr<Decoder>(new RFC5424Decoder(*this, 0)); }
  private:
    RFC5424Decoder(const RFC5424Decoder& o, int) : Decoder(o, 0) {}
};
class GelfDecoder : public Decoder {  // decoder/gelf_decoder.rs:10-125
  public:
    explicit GelfDecoder(int device = 0) : Decoder(FG_GELF, device, nullptr) {}
    std::unique_ptr<Decoder> clone_boxed() const override { return std::unique_ptr<Decoder>(new GelfDecoder(*this, 0)); }
  private:
    GelfDecoder(const GelfDecoder& o, int) : Decoder(o, 0) {}
};
class LTSVDecoder : public Decoder {  // decoder/ltsv_decoder.rs:17-221
  public:
    explicit LTSVDecoder(const LtsvConfig& cfg, int device = 0) : Decoder(FG_LTSV, device, &cfg) {}
    std::unique_ptr<Decoder> clone_boxed() const override { return std::unique_ptr<Decoder>(new LTSVDecoder(*this, 0)); }
  private:
    LTSVDecoder(const LTSVDecoder& o, int) : Decoder(o, 0) {}
};

class RFC3164Decoder : public Decoder {  // decoder/rfc3164_decoder.rs:14-213
  public:
    // current_year / tz: what the reference reads from the wall clock (:179) and from time_tz's zone database (:195)
    // become configuration; tz may be null (no zone name is recognised).  fg_clone carries both over.
    RFC3164Decoder(int32_t current_year, const fg_tz_table* tz, int device = 0) : Decoder(FG_RFC3164, device, nullptr) {
        fg_rfc3164_cfg c{current_year, tz};
        if (fg_set_rfc3164(ctx_, &c) != FG_OK) throw std::runtime_error("fg_set_rfc3164 failed");
    }
    std::unique_ptr<Decoder> clone_boxed() const override { return std::unique_ptr<Decoder>(new RFC3164Decoder(*this, 0)); }
  private:
    RFC3164Decoder(const RFC3164Decoder& o, int) : Decoder(o, 0) {}
};

// ---------------------------------------------------------------------------------------------
// Batching framers: same framing and error reporting as the reference splitters, one
// decode_batch per `max_lines` / `max_bytes` instead of one decode per line.
// ---------------------------------------------------------------------------------------------
namespace detail {
inline bool valid_utf8(const uint8_t* s, size_t n) {  // std::str::from_utf8
    size_t i = 0;
    while (i < n) {
        uint8_t c = s[i];
        if (c < 0x80) { ++i; continue; }
        size_t len; uint32_t cp;
        if (c >= 0xC2 && c <= 0xDF) { len = 2; cp = c & 0x1F; }
        else if (c >= 0xE0 && c <= 0xEF) { len = 3; cp = c & 0x0F; }
        else if (c >= 0xF0 && c <= 0xF4) { len = 4; cp = c & 0x07; }
        else return false;
        if (i + len > n) return false;
        for (size_t k = 1; k < len; ++k) { if ((s[i + k] & 0xC0) != 0x80) return false; cp = (cp << 6) | (s[i + k] & 0x3F); }
        if ((len == 3 && cp < 0x800) || (len == 4 && (cp < 0x10000 || cp > 0x10FFFF)) || (cp >= 0xD800 && cp <= 0xDFFF)) return false;
        i += len;
    }
    return true;
}
inline bool is_ws_cp(uint32_t c) {
    return (c >= 9 && c <= 13) || c == 0x20 || c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) ||
           c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}
inline std::string_view trim(std::string_view s) {  // str::trim on valid UTF-8
    auto dec = [&](size_t i, uint32_t* cp) -> size_t {
        uint8_t b = (uint8_t)s[i];
        if (b < 0x80) { *cp = b; return 1; }
        size_t len = b < 0xE0 ? 2 : b < 0xF0 ? 3 : 4;
        uint32_t v = b & (0xFF >> (len + 1));
        for (size_t k = 1; k < len && i + k < s.size(); ++k) v = (v << 6) | ((uint8_t)s[i + k] & 0x3F);
        *cp = v;
        return len;
    };
    size_t b = 0, e = s.size();
    while (b < e) { uint32_t cp; size_t n = dec(b, &cp); if (!is_ws_cp(cp)) break; b += n; }
    while (e > b) {
        size_t p = e - 1;
        while (p > b && ((uint8_t)s[p] & 0xC0) == 0x80) --p;
        uint32_t cp; dec(p, &cp);
        if (!is_ws_cp(cp)) break;
        e = p;
    }
    return s.substr(b, e - b);
}
}  // namespace detail


// ---------------------------------------------------------------------------------------------
// What the splitters read from, and WHEN a batch is handed to the GPU.
//
// The reference handles every line as soon as `buf_reader.lines()` yields it and treats a read timeout (ErrorKind::WouldBlock:
// tcp_input.rs:41 sets `input.timeout` on the socket) as "Client hasn't sent any data for a while - Closing idle connection"
// (line_splitter.rs:26-33, nul_splitter.rs:22-29).  A batching framer that waited for a full chunk would sit on a trickling
// connection's lines for hours, so a batch is flushed when ANY of these holds:
//   * size      max_bytes / max_lines reached;
//   * drained   the source has nothing more to read right now (a read would block) -- after a LINGER of at most max_latency_ms
//               since the batch's first byte, and only while the batch is still small (< linger_below bytes): a burst that is
//               still arriving becomes one GPU call instead of one per packet;
//   * end       EOF, or the idle timeout: what is complete is decoded, then the reference's message is printed and run() returns.
// ---------------------------------------------------------------------------------------------
struct ByteSource {
    enum : long { kEof = 0, kWouldBlock = -1, kError = -2 };
    virtual ~ByteSource() = default;
    // up to cap bytes into dst; waits at most wait_ms for the first of them (< 0: for as long as it takes, 0: only what is
    // readable now).  > 0 bytes read, kEof, kWouldBlock (nothing within wait_ms), kError.
    virtual long read(uint8_t* dst, size_t cap, int wait_ms) = 0;
};
// a file, a string stream: never blocks, a short read means EOF is next
struct IstreamSource : ByteSource {
    std::istream& in;
    explicit IstreamSource(std::istream& s) : in(s) {}
    long read(uint8_t* dst, size_t cap, int) override {
        in.read((char*)dst, (std::streamsize)cap);
        const long got = (long)in.gcount();
        return got > 0 ? got : (long)kEof;
    }
};
// a socket / pipe / tty file descriptor (BufReader<TcpStream>, stdin): poll() + read()
struct FdSource : ByteSource {
    int fd;
    explicit FdSource(int f) : fd(f) {}
    long read(uint8_t* dst, size_t cap, int wait_ms) override;
};
struct FlushPolicy {
    size_t max_bytes = 8u << 20;     // a batch never grows beyond this (the GPU path is at link speed long before)
    size_t max_lines = 1u << 16;     // host-framing splitters only
    int idle_timeout_ms = -1;        // `input.timeout` (tcp_input.rs:26,41); < 0: none (stdin, files)
    int max_latency_ms = 5;          // linger: how long the first line of a batch may wait for company
    size_t linger_below = 64u << 10; // ... and only while the batch is smaller than this
};

}  // namespace fg
#include <poll.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
namespace fg {
inline long FdSource::read(uint8_t* dst, size_t cap, int wait_ms) {
    for (;;) {
        pollfd p{fd, POLLIN, 0};
        const int r = ::poll(&p, 1, wait_ms);
        if (r == 0) return kWouldBlock;
        if (r < 0) {
            if (errno == EINTR) continue;  // ErrorKind::Interrupted => continue (line_splitter.rs:21)
            return kError;
        }
        const ssize_t n = ::read(fd, dst, cap);
        if (n > 0) return (long)n;
        if (n == 0) return kEof;
        if (errno == EINTR) continue;
        if (errno == EAGAIN || errno == EWOULDBLOCK) return kWouldBlock;
        return kError;
    }
}

// Buffered reader over a ByteSource with the flush policy built in: the splitter registers `pending` (bytes of framed lines it
// holds back for the next GPU call) and `flush`; before the reader BLOCKS for more input it lingers (policy) and then flushes.
class BufferedSource {
  public:
    enum End { None, Eof, Idle, Error };
    BufferedSource(ByteSource& s, const FlushPolicy& p) : src_(s), pol_(p) {}
    void on_block(std::function<size_t()> pending, std::function<void()> flush) {
        pending_ = std::move(pending);
        flush_ = std::move(flush);
    }
    End end() const { return end_; }
    // next byte, or -1 at the end of the input (end() says which end)
    int get() {
        if (pos_ == buf_.size() && !more()) return -1;
        return buf_[pos_++];
    }
    // up to n bytes appended to out; fewer only at the end of the input
    size_t read_exact(std::string& out, size_t n) {
        size_t got = 0;
        while (got < n) {
            if (pos_ == buf_.size() && !more()) break;
            const size_t k = std::min(n - got, buf_.size() - pos_);
            out.append((const char*)buf_.data() + pos_, k);
            pos_ += k;
            got += k;
        }
        return got;
    }
    // bytes up to (not including) the next `delim` appended to out; true = the delimiter was found (and consumed)
    bool read_until(uint8_t delim, std::string& out) {
        for (;;) {
            if (pos_ == buf_.size() && !more()) return false;
            const uint8_t* b = buf_.data() + pos_;
            const uint8_t* hit = (const uint8_t*)memchr(b, delim, buf_.size() - pos_);
            const size_t k = hit ? (size_t)(hit - b) : buf_.size() - pos_;
            out.append((const char*)b, k);
            pos_ += k + (hit ? 1 : 0);
            if (hit) return true;
        }
    }
    // a raw chunk for the GPU framers: appends what is there to `out` (at most room bytes) under the same policy, where the
    // batch in the making is `out` itself.  false = nothing was added and the input has ended.
    template <class Vec>
    bool read_chunk(Vec& out, size_t room) {
        using clock = std::chrono::steady_clock;
        const size_t had = out.size();
        bool started = false;
        clock::time_point t0{};
        while (out.size() - had < room) {
            const size_t want = std::min(room - (out.size() - had), (size_t)1 << 20);
            const size_t at = out.size();
            out.resize(at + want);
            int wait = 0;
            if (out.size() - want == had && !started) wait = pol_.idle_timeout_ms;  // the first bytes: block (idle timeout)
            long n = src_.read(out.data() + at, want, wait);
            if (n == ByteSource::kWouldBlock && started && at < had + pol_.linger_below) {  // drained: linger while the batch is small
                const auto left = std::chrono::milliseconds(pol_.max_latency_ms) - std::chrono::duration_cast<std::chrono::milliseconds>(clock::now() - t0);
                if (left.count() > 0) n = src_.read(out.data() + at, want, (int)left.count());
            }
            out.resize(at + (n > 0 ? (size_t)n : 0));
            if (n > 0) {
                if (!started) {
                    started = true;
                    t0 = clock::now();
                }
                continue;
            }
            if (n == ByteSource::kWouldBlock && started) break;  // drained: the batch goes out
            end_ = n == ByteSource::kEof ? Eof : n == ByteSource::kWouldBlock ? Idle : Error;
            break;
        }
        return out.size() > had;
    }

  private:
    bool more() {
        using clock = std::chrono::steady_clock;
        if (end_ != None) return false;
        buf_.resize(1 << 16);
        pos_ = 0;
        long n = src_.read(buf_.data(), buf_.size(), 0);  // whatever is readable now
        if (n == ByteSource::kWouldBlock) {
            // a read would block.  Lines held back for the next GPU call: linger while the batch is young and small, then flush
            const size_t pend = pending_ ? pending_() : 0;
            if (pend) {
                if (!batch_open_) {
                    batch_open_ = true;
                    batch_t0_ = clock::now();
                }
                if (pend < pol_.linger_below) {
                    const auto left = std::chrono::milliseconds(pol_.max_latency_ms) - std::chrono::duration_cast<std::chrono::milliseconds>(clock::now() - batch_t0_);
                    if (left.count() > 0) n = src_.read(buf_.data(), buf_.size(), (int)left.count());
                }
                if (n == ByteSource::kWouldBlock) {
                    flush_();
                    batch_open_ = false;
                }
            }
            if (n == ByteSource::kWouldBlock) n = src_.read(buf_.data(), buf_.size(), pol_.idle_timeout_ms);
        }
        if (n > 0) {
            if (!batch_open_) {
                batch_open_ = true;
                batch_t0_ = clock::now();
            }
            buf_.resize((size_t)n);
            return true;
        }
        buf_.clear();
        end_ = n == ByteSource::kEof ? Eof : n == ByteSource::kWouldBlock ? Idle : Error;
        return false;
    }
    ByteSource& src_;
    FlushPolicy pol_;
    std::vector<uint8_t> buf_;
    size_t pos_ = 0;
    End end_ = None;
    std::function<size_t()> pending_;
    std::function<void()> flush_;
    bool batch_open_ = false;
    std::chrono::steady_clock::time_point batch_t0_{};
};
inline const char* kIdleMessage = "Client hasn't sent any data for a while - Closing idle connection";  // line_splitter.rs:26-33

using RecordSink = std::function<void(Record&&)>;  // stands in for encoder.encode(record) -> tx.send(bytes)

class BatchingSplitter {
  public:
    enum Framing { Line, Nul, Syslen };
    BatchingSplitter(Framing f, size_t max_lines = 1 << 16, size_t max_bytes = 32u << 20) : f_(f), max_lines_(max_lines), max_bytes_(max_bytes) {}

    // Mirrors LineSplitter/NulSplitter/SyslenSplitter::run: frames `in`, decodes in batches, hands Ok records to
    // `sink` in input order, reports errors on `err` exactly like the reference.
    void run(std::istream& in, const Decoder& decoder, const RecordSink& sink, std::ostream& err) {
        IstreamSource src(in);
        FlushPolicy pol;
        pol.max_lines = max_lines_;
        pol.max_bytes = max_bytes_;
        run(src, pol, decoder, sink, err);
    }
    // The same over a socket / pipe (FdSource) with the flush policy above: a batch goes to the GPU when it is full, when the
    // source has nothing more to give right now (after the linger), at EOF and at the idle timeout.
    void run(ByteSource& src, const FlushPolicy& pol, const Decoder& decoder, const RecordSink& sink, std::ostream& err) {
        max_lines_ = pol.max_lines;
        max_bytes_ = pol.max_bytes;
        bytes_.clear();
        offsets_.assign(1, 0);
        // (pinned once: a batch never outgrows max_bytes + one line; a longer line re-pins, once)
        bytes_.reserve(std::min<size_t>(max_bytes_, (size_t)64 << 20) + (1u << 20));
        offsets_.reserve(std::min<size_t>(max_lines_, (size_t)1 << 20) + 2);
        BufferedSource in(src, pol);
        in.on_block([&] { return bytes_.size() + (offsets_.size() - 1); }, [&] { flush(decoder, sink, err); });
        std::string line;
        if (f_ == Syslen) {
            for (;;) {  // syslen_splitter.rs:42-57: "<len> " then exactly len bytes
                // read_msglen: read_until(b' ') -- at EOF without a space the reference still drops the LAST byte it
                // read (it assumes the delimiter), parses the rest and then fails in read_exact (:27-30)
                std::string num;
                int c;
                while ((c = in.get()) >= 0 && c != ' ') num.push_back((char)c);
                if (c < 0 && in.end() != BufferedSource::Eof) {  // read_until returned Err (WouldBlock: idle timeout): :45 `Err(_)`
                    flush(decoder, sink, err);
                    err << "Can't read message's length\n";
                    break;
                }
                const bool at_eof = c < 0;
                const size_t got = num.size() + (at_eof ? 0 : 1);  // bytes read_until returned
                if (got <= 1) { flush(decoder, sink, err); err << "Can't read message's length\n"; break; }  // :45-46, :20-25
                if (at_eof) num.pop_back();
                size_t len = 0;
                bool ok = !num.empty();
                size_t k = (!num.empty() && num[0] == '+') ? 1 : 0;  // usize::from_str accepts a leading '+'
                if (k >= num.size()) ok = false;
                for (; k < num.size() && ok; ++k) {
                    if (num[k] < '0' || num[k] > '9' || len > (SIZE_MAX - 9) / 10) ok = false;
                    else len = len * 10 + (size_t)(num[k] - '0');
                }
                if (!ok) { flush(decoder, sink, err); err << "Can't read message's length\n"; break; }
                line.clear();
                const size_t have = at_eof ? 0 : in.read_exact(line, len);
                if (at_eof ? len != 0 : have != len) {  // read_exact's error, printed with `{}` (:27-30)
                    flush(decoder, sink, err);
                    err << (in.end() == BufferedSource::Idle ? "Resource temporarily unavailable (os error 11)\n"  // io::Error Display of EAGAIN
                                                              : "failed to fill whole buffer\n");                  // UnexpectedEof
                    break;
                }
                if (at_eof) {  // "0?" + EOF: an empty message is handled, then the next read_msglen sees Ok(0)
                    push(line, decoder, sink, err);
                    flush(decoder, sink, err);
                    err << "Can't read message's length\n";
                    break;
                }
                if (!detail::valid_utf8((const uint8_t*)line.data(), line.size())) {
                    // String::from_utf8(buffer).unwrap() (:32): the reference PANICS here, i.e. this connection's thread
                    // ends; mirrored as: everything before it is delivered, the panic text goes to `err`, run() returns
                    flush(decoder, sink, err);
                    err << "thread panicked: called `Result::unwrap()` on an `Err` value: FromUtf8Error (syslen_splitter.rs:32)\n";
                    return;
                }
                push(line, decoder, sink, err);
            }
        } else {
            const uint8_t delim = f_ == Line ? '\n' : '\0';
            for (;;) {
                line.clear();
                const bool terminated = in.read_until(delim, line);
                if (!terminated) {
                    // BufRead::lines() / split(): at EOF an unterminated last piece is a line when it is not empty (and keeps its
                    // trailing '\r': that is dropped only together with the '\n' it precedes); on any other end (idle timeout, I/O
                    // error) the iterator yields Err and the partial line is gone with it
                    if (in.end() == BufferedSource::Eof && !line.empty()) push(line, decoder, sink, err);
                    break;
                }
                if (f_ == Line && !line.empty() && line.back() == '\r') line.pop_back();
                push(line, decoder, sink, err);
            }
            flush(decoder, sink, err);
            if (in.end() == BufferedSource::Idle) err << kIdleMessage << "\n";  // line_splitter.rs:26-33, nul_splitter.rs:22-29
            return;
        }
        flush(decoder, sink, err);
    }

  private:
    void push(const std::string& line, const Decoder& d, const RecordSink& sink, std::ostream& err) {
        if (!detail::valid_utf8((const uint8_t*)line.data(), line.size())) {
            flush(d, sink, err);  // keep stderr/record order
            err << "Invalid UTF-8 input\n";  // line_splitter.rs:22-25, nul_splitter.rs:35-38
            return;
        }
        bytes_.insert(bytes_.end(), line.begin(), line.end());
        offsets_.push_back(bytes_.size());
        if (offsets_.size() - 1 >= max_lines_ || bytes_.size() >= max_bytes_) flush(d, sink, err);
    }
    void flush(const Decoder& d, const RecordSink& sink, std::ostream& err) {
        const uint64_t n = offsets_.size() - 1;
        if (n == 0) return;
        bytes_.resize(bytes_.size() + 16);  // readable slack (the data itself is not touched)
        auto res = d.decode_batch(bytes_.data(), offsets_.back(), offsets_.data(), n);
        for (uint64_t i = 0; i < n; ++i) {
            if (res[i].ok()) { sink(std::move(res[i].record)); continue; }
            std::string_view ln((const char*)bytes_.data() + offsets_[i], offsets_[i + 1] - offsets_[i]);
            std::string_view t = detail::trim(ln);
            if (f_ == Nul && t.empty()) continue;  // nul_splitter.rs:41-46
            err << res[i].err << ": [" << t << "]\n";  // line_splitter.rs:37-39
        }
        bytes_.clear();
        offsets_.assign(1, 0);
    }
    Framing f_;
    size_t max_lines_, max_bytes_;
    PinnedBytes bytes_;       // page-locked: fg_decode_batch reads them in place (zero-copy)
    PinnedOffsets offsets_;
};


// ---------------------------------------------------------------------------------------------
// The per-record callers: UDP, Redis and the file tailer call `decoder.decode(record)` once per record with no splitter
// in between (input/udp_input.rs:139, input/redis_input.rs:159, input/file/worker.rs:116).  Through the drop-in trait method
// that is one GPU launch + stream synchronisation + table copy per record -- tens of microseconds against ~0.15 us for the CPU
// decoder -- so for these inputs the GPU path only pays when records are decoded in batches (tools/host_path_bench.py
// --workload latency: the crossover with one CPU thread is at a few hundred lines per call).  MicroBatcher is that adapter:
// records are parked for at most max_latency_ms (or until max_lines / max_bytes are there) and decoded in ONE call; results
// come back in arrival order.  The receive loop of udp_input.rs:78-88 becomes
//     for (;;) { n = recv(sock, buf, timeout = mb.wait_ms()); if (n > 0) mb.push({buf, n}); mb.poll(); }
// ---------------------------------------------------------------------------------------------
class MicroBatcher {
  public:
    using ErrorSink = std::function<void(const char* err, std::string_view record)>;  // udp_input.rs:84-86: writeln!(stderr(), "{}", e)
    MicroBatcher(const Decoder& d, RecordSink sink, ErrorSink on_error, size_t max_lines = 4096, int max_latency_ms = 5,
                 size_t max_bytes = 4u << 20)
        : d_(d), sink_(std::move(sink)), on_error_(std::move(on_error)), max_lines_(max_lines), max_bytes_(max_bytes), latency_(max_latency_ms) {
        offsets_.assign(1, 0);
    }
    ~MicroBatcher() { flush(); }
    // one record (handle_record, udp_input.rs:125-143): the UTF-8 check happens here, in arrival order with the decode errors
    void push(std::string_view record) {
        if (offsets_.size() == 1) t0_ = std::chrono::steady_clock::now();
        bytes_.insert(bytes_.end(), record.begin(), record.end());
        offsets_.push_back(bytes_.size());
        if (offsets_.size() - 1 >= max_lines_ || bytes_.size() >= max_bytes_) flush();
    }
    // how long the caller may block in its receive call before poll() is due: -1 = nothing is parked
    int wait_ms() const {
        if (offsets_.size() == 1) return -1;
        const auto waited = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0_).count();
        return waited >= latency_ ? 0 : (int)(latency_ - waited);
    }
    void poll() {
        if (offsets_.size() > 1 && wait_ms() == 0) flush();
    }
    void flush() {
        const uint64_t n = offsets_.size() - 1;
        if (n == 0) return;
        // invalid UTF-8 never reaches decode() ("Invalid UTF-8 input", udp_input.rs:135-138): such records are cut out of the
        // batch but reported at their place
        std::vector<uint8_t> ok_bytes;
        std::vector<uint64_t> ok_offs(1, 0);
        std::vector<uint8_t> valid(n);
        for (uint64_t i = 0; i < n; ++i) {
            valid[i] = detail::valid_utf8(bytes_.data() + offsets_[i], offsets_[i + 1] - offsets_[i]);
            if (valid[i]) {
                ok_bytes.insert(ok_bytes.end(), bytes_.begin() + (std::ptrdiff_t)offsets_[i], bytes_.begin() + (std::ptrdiff_t)offsets_[i + 1]);
                ok_offs.push_back(ok_bytes.size());
            }
        }
        ok_bytes.resize(ok_bytes.size() + 16);  // readable slack
        auto res = d_.decode_batch(ok_bytes.data(), ok_offs.back(), ok_offs.data(), ok_offs.size() - 1);
        uint64_t j = 0;
        for (uint64_t i = 0; i < n; ++i) {
            std::string_view rec((const char*)bytes_.data() + offsets_[i], offsets_[i + 1] - offsets_[i]);
            if (!valid[i]) {
                on_error_("Invalid UTF-8 input", rec);
                continue;
            }
            if (res[j].ok()) sink_(std::move(res[j].record));
            else on_error_(res[j].err, rec);
            ++j;
        }
        bytes_.clear();
        offsets_.assign(1, 0);
    }
    size_t pending() const { return offsets_.size() - 1; }

  private:
    const Decoder& d_;
    RecordSink sink_;
    ErrorSink on_error_;
    size_t max_lines_, max_bytes_;
    long latency_;
    std::vector<uint8_t> bytes_;
    std::vector<uint64_t> offsets_;
    std::chrono::steady_clock::time_point t0_{};
};


// LineSplitter / NulSplitter with the framing itself on the GPU: raw chunks of the stream go to
// fg_frame_decode_batch (framing + UTF-8 validation + decode in one call); the host only carries an
// unterminated tail over to the next chunk and reports like the reference.
class GpuFramingSplitter {
  public:
    enum Framing { Line, Nul };
    explicit GpuFramingSplitter(Framing f, size_t chunk_bytes = 8u << 20) : f_(f), chunk_(chunk_bytes) {}

    void run(std::istream& in, const Decoder& d, const RecordSink& sink, std::ostream& err) {
        IstreamSource src(in);
        FlushPolicy pol;
        pol.max_bytes = chunk_;
        run(src, pol, d, sink, err);
    }
    // over a socket / pipe with the flush policy (see FlushPolicy): whatever has arrived goes to the GPU when the source runs dry
    void run(ByteSource& src, const FlushPolicy& pol, const Decoder& d, const RecordSink& sink, std::ostream& err) {
        chunk_ = pol.max_bytes;
        BufferedSource in(src, pol);
        PinnedBytes buf;  // the unterminated tail of the last chunk + what arrived since (page-locked: uploads at link speed)
        buf.reserve(std::min<size_t>(chunk_, (size_t)64 << 20) + (1u << 20) + 64);
        for (;;) {
            const bool got = in.read_chunk(buf, chunk_ > buf.size() ? chunk_ - buf.size() : chunk_);
            const bool eof = in.end() == BufferedSource::Eof;
            if (!got && in.end() != BufferedSource::None && !(eof && !buf.empty())) break;  // (at EOF the tail is a frame)
            if (buf.empty()) break;
            if (!eof && in.end() == BufferedSource::None && !memchr(buf.data(), f_ == Line ? '\n' : 0, buf.size())) {
                if (buf.size() >= chunk_) chunk_ *= 2;  // one frame longer than the chunk: read more
                continue;                                // no frame is complete yet
            }
            fg_tables t{};
            const uint64_t* off = nullptr;
            uint64_t n = 0, consumed = 0;
            buf.resize(buf.size() + 16);  // readable slack
            const uint64_t nbytes = buf.size() - 16;
            int rc = fg_frame_decode_batch(d.ctx(), d.format(), f_ == Line ? FG_FRAME_LINE : FG_FRAME_NUL, buf.data(), nbytes,
                                           eof ? 1 : 0, &t, &off, &n, &consumed);
            if (rc != FG_OK) throw std::runtime_error("fg_frame_decode_batch failed: " + std::to_string(rc));
            if (n) {
                Decoder::side_effects(d.format(), f_ == Line ? FG_FRAME_LINE : FG_FRAME_NUL, buf.data(), off, t);
                emit(d, buf.data(), off, n, t, sink, err);
            }
            buf.resize(nbytes);
            if (consumed == 0 && !eof && n == 0 && buf.size() >= chunk_) chunk_ *= 2;  // one frame longer than the chunk: read more
            buf.erase(buf.begin(), buf.begin() + (std::ptrdiff_t)consumed);
            if (eof || in.end() != BufferedSource::None) break;
        }
        if (in.end() == BufferedSource::Idle) err << kIdleMessage << "\n";  // (the unterminated tail is dropped, as lines() drops it)
    }

  private:
    void emit(const Decoder& d, const uint8_t* bytes, const uint64_t* off, uint64_t n, const fg_tables& t, const RecordSink& sink,
              std::ostream& err) {
        std::vector<uint64_t> so(n + 1);
        int64_t total = fg_tables_serialize(d.format(), d.cfg_ptr(), bytes, off, &t, 0, n, nullptr, 0, so.data());
        if (total < 0) throw std::runtime_error("fg_tables_serialize failed");
        std::vector<uint8_t> blob((size_t)total + 1);
        fg_tables_serialize(d.format(), d.cfg_ptr(), bytes, off, &t, 0, n, blob.data(), (uint64_t)total, so.data());
        for (uint64_t i = 0; i < n; ++i) {
            const uint8_t st = FG_META_STATUS(t.meta[i]);
            if (st == FG_ST_BAD_UTF8) {
                err << "Invalid UTF-8 input\n";  // line_splitter.rs:22-25, nul_splitter.rs:35-38
                continue;
            }
            DecodeResult r = detail::from_canonical(blob.data() + so[i], d.format(), st);
            if (r.ok()) {
                sink(std::move(r.record));
                continue;
            }
            // the line as the reference saw it: without its terminator
            uint64_t b = off[i], e = off[i + 1];
            if (f_ == Line) {
                if (e > b && bytes[e - 1] == '\n') { --e; if (e > b && bytes[e - 1] == '\r') --e; }
            } else if (e > b && bytes[e - 1] == 0) {
                --e;
            }
            std::string_view tl = detail::trim(std::string_view((const char*)bytes + b, e - b));
            if (f_ == Nul && tl.empty()) continue;  // nul_splitter.rs:41-46
            err << r.err << ": [" << tl << "]\n";   // line_splitter.rs:37-39
        }
    }
    Framing f_;
    size_t chunk_;
};

// ---------------------------------------------------------------------------------------------
// Encoder / Merger configuration (encoder/mod.rs:54-56, merger/mod.rs:30-32) and the splitter that runs the WHOLE
// handle_line on the GPU (fg_transcode_batch): framing, UTF-8 check, decode, encode, merger.  What comes back is the
// byte stream the output writes, plus one verdict per line for the reference's stderr messages.
// ---------------------------------------------------------------------------------------------
struct EncoderConfig {
    fg_encoder encoder = FG_ENC_GELF;
    fg_merger merger = FG_MERGE_LINE;
    std::map<std::string, std::string> extra;  // output.gelf_extra / output.ltsv_extra (a sorted table, like toml's)
    std::optional<std::string> prepend;        // the formatted output.syslog_prepend_timestamp header
    double now_ts = 0.0;                       // ts of GELF records decoded without "timestamp" (gelf_decoder.rs:109)
};

class TranscodingSplitter {
  public:
    enum Framing { Line, Nul, Syslen };
    TranscodingSplitter(Framing f, EncoderConfig enc, size_t chunk_bytes = 8u << 20) : f_(f), enc_(std::move(enc)), chunk_(chunk_bytes) {}

    // LineSplitter / NulSplitter / SyslenSplitter::run (splitter/*_splitter.rs) + the output thread's write: `out`
    // receives the encoded, framed messages in input order; `err` what the reference prints for dropped lines.
    // Line / Nul: the GPU frames the raw chunks.  Syslen ("<len> " + len bytes, syslen_splitter.rs:42-57): the length
    // prefixes are a sequential chain -- the host hops from prefix to prefix (it never touches the message bytes
    // otherwise) and hands the GPU the packed messages with their offsets.
    void run(std::istream& in, const Decoder& d, std::ostream& out, std::ostream& err) {
        IstreamSource src(in);
        FlushPolicy pol;
        pol.max_bytes = chunk_;
        run(src, pol, d, out, err);
    }
    // over a socket / pipe with the flush policy (see FlushPolicy)
    void run(ByteSource& src, const FlushPolicy& pol, const Decoder& d, std::ostream& out, std::ostream& err) {
        chunk_ = pol.max_bytes;
        std::vector<const char*> ks, vs;
        for (auto& kv : enc_.extra) { ks.push_back(kv.first.c_str()); vs.push_back(kv.second.c_str()); }
        fg_encode_cfg ec{};
        ec.encoder = enc_.encoder;
        ec.merger = enc_.merger;
        ec.n_extra = (uint32_t)ks.size();
        ec.extra_keys = ks.data();
        ec.extra_values = vs.data();
        ec.prepend = enc_.prepend ? enc_.prepend->c_str() : nullptr;
        ec.now_ts = enc_.now_ts;
        BufferedSource bin(src, pol);
        if (f_ == Syslen) {
            run_syslen(bin, d, ec, out, err);
            return;
        }
        std::vector<uint8_t> buf;  // the unterminated tail of the last chunk + what arrived since
        for (;;) {
            const bool got = bin.read_chunk(buf, chunk_ > buf.size() ? chunk_ - buf.size() : chunk_);
            const bool eof = bin.end() == BufferedSource::Eof;
            if (!got && bin.end() != BufferedSource::None && !(eof && !buf.empty())) break;  // (at EOF the tail is a frame)
            if (buf.empty()) break;
            if (!eof && bin.end() == BufferedSource::None && !memchr(buf.data(), f_ == Line ? '\n' : 0, buf.size())) {
                if (buf.size() >= chunk_) chunk_ *= 2;  // one frame longer than the chunk: read more
                continue;                                // no frame is complete yet
            }
            fg_transcoded r{};
            int rc = fg_transcode_batch(d.ctx(), d.format(), f_ == Line ? FG_FRAME_LINE : FG_FRAME_NUL, &ec, buf.data(), buf.size(), nullptr,
                                        0, eof ? 1 : 0, &r);
            if (rc != FG_OK) throw std::runtime_error("fg_transcode_batch failed: " + std::to_string(rc));
            if (r.out_bytes) {
                out.write((const char*)r.out, (std::streamsize)r.out_bytes);
                out.flush();  // (the batch is what has arrived: it goes out now, not when the stream's buffer fills)
            }
            report(d, r, buf.data(), r.frame_offsets, err);
            if (r.consumed == 0 && !eof && r.n == 0 && buf.size() >= chunk_) chunk_ *= 2;
            buf.erase(buf.begin(), buf.begin() + (std::ptrdiff_t)r.consumed);
            if (eof || bin.end() != BufferedSource::None) break;
        }
        if (bin.end() == BufferedSource::Idle) err << kIdleMessage << "\n";  // line_splitter.rs:26-33
    }

  private:
    // stderr for the lines the pipeline dropped, as the reference prints them
    void report(const Decoder& d, const fg_transcoded& r, const uint8_t* bytes, const uint64_t* offs, std::ostream& err) const {
        if (r.n) {  // the decoder's stdout side effects, from the meta column alone
            fg_tables only_meta{};
            only_meta.n = r.n;
            only_meta.meta = const_cast<uint32_t*>(r.meta);
            Decoder::side_effects(d.format(), f_ == Line ? FG_FRAME_LINE : f_ == Nul ? FG_FRAME_NUL : FG_FRAME_NONE, bytes, offs, only_meta);
        }
        for (uint64_t i = 0; i < r.n; ++i) {
            const uint8_t st = FG_META_STATUS(r.meta[i]), es = r.enc_status[i];
            if (st == 0 && es == 0) continue;
            if (st == FG_ST_BAD_UTF8) {
                err << "Invalid UTF-8 input\n";  // line_splitter.rs:22-25, nul_splitter.rs:35-38
                continue;
            }
            uint64_t b = offs[i], e = offs[i + 1];  // the line without its terminator
            if (f_ == Line) {
                if (e > b && bytes[e - 1] == '\n') { --e; if (e > b && bytes[e - 1] == '\r') --e; }
            } else if (f_ == Nul && e > b && bytes[e - 1] == 0) {
                --e;
            }
            std::string_view tl = detail::trim(std::string_view((const char*)bytes + b, e - b));
            if (f_ == Nul && tl.empty()) continue;  // nul_splitter.rs:41-46
            const char* msg = st ? fg_error_string(d.format(), st) : fg_encode_error_string(es);
            err << (msg ? msg : "?") << ": [" << tl << "]\n";  // line_splitter.rs:37-39, syslen_splitter.rs:35-37
        }
    }
    void flush_syslen(const Decoder& d, const fg_encode_cfg& ec, std::vector<uint8_t>& bytes, std::vector<uint64_t>& offs,
                      std::ostream& out, std::ostream& err) const {
        const uint64_t n = offs.size() - 1;
        if (n == 0) return;
        const uint64_t nbytes = bytes.size();
        bytes.resize(nbytes + 16);  // readable slack
        fg_transcoded r{};
        int rc = fg_transcode_batch(d.ctx(), d.format(), FG_FRAME_NONE, &ec, bytes.data(), nbytes, offs.data(), n, 1, &r);
        if (rc != FG_OK) throw std::runtime_error("fg_transcode_batch failed: " + std::to_string(rc));
        if (r.out_bytes) {
            out.write((const char*)r.out, (std::streamsize)r.out_bytes);
            out.flush();
        }
        report(d, r, bytes.data(), offs.data(), err);
        bytes.clear();
        offs.assign(1, 0);
    }
    void run_syslen(BufferedSource& in, const Decoder& d, const fg_encode_cfg& ec, std::ostream& out, std::ostream& err) {
        std::vector<uint8_t> bytes;
        std::vector<uint64_t> offs(1, 0);
        in.on_block([&] { return bytes.size() + (offs.size() - 1); }, [&] { flush_syslen(d, ec, bytes, offs, out, err); });
        std::string num, msg;
        for (;;) {
            num.clear();
            int c;
            while ((c = in.get()) >= 0 && c != ' ') num.push_back((char)c);
            // read_msglen (:42-57): EOF / idle timeout / nothing before the space / not a usize -> "Can't read message's length" (:20-25)
            bool ok = c >= 0 && !num.empty();
            size_t len = 0, k = (ok && num[0] == '+') ? 1 : 0;
            if (ok && k >= num.size()) ok = false;
            for (; ok && k < num.size(); ++k) {
                if (num[k] < '0' || num[k] > '9') ok = false;
                else len = len * 10 + (size_t)(num[k] - '0');
            }
            if (!ok) {
                flush_syslen(d, ec, bytes, offs, out, err);
                err << "Can't read message's length\n";
                return;
            }
            msg.clear();
            if (in.read_exact(msg, len) != len) {  // read_exact fails (:27-30): the partial frame is dropped
                flush_syslen(d, ec, bytes, offs, out, err);
                err << (in.end() == BufferedSource::Idle ? "Resource temporarily unavailable (os error 11)\n" : "failed to fill whole buffer\n");
                return;
            }
            if (!detail::valid_utf8((const uint8_t*)msg.data(), len)) {  // String::from_utf8(..).unwrap() panics (:33): the thread ends
                flush_syslen(d, ec, bytes, offs, out, err);
                err << "Invalid UTF-8 input\n";
                return;
            }
            bytes.insert(bytes.end(), msg.begin(), msg.end());
            offs.push_back(bytes.size());
            if (bytes.size() >= chunk_) flush_syslen(d, ec, bytes, offs, out, err);
        }
    }
    Framing f_;
    EncoderConfig enc_;
    size_t chunk_;
};

}  // namespace fg
