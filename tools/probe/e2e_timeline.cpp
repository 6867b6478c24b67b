// Per-slice timeline of the sliced host path, outside the library: why do the GELF and LTSV corpora run at the SUM of their upload and
// download times through fg_decode_batch when the RFC5424 corpora overlap the two directions?
// Replays the library's own shape -- uploads on one stream, fg_decode_batch_device per slice on a second, rows + entry ranges back on a
// third, chained by events -- with a pair of TIMING events around every operation, and prints where each operation of each slice
// started and ended (ms from the first upload), the busy time of the three roles and how long an upload and a download were in
// flight at the same moment.
// usage: e2e_timeline <rfc5424|gelf|ltsv> <file of '\n'-separated lines> [slice MiB = 32] [mode: all | collect]
//   all      everything is queued at once, downloads wait for the kernels through events, entry ranges are sized from a dry run
//   collect  like fg_decode_batch: the host waits for a slice's entry counter, then queues its downloads (sixteen slices issued ahead)
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/probe/e2e_timeline.cpp -o tools/probe/e2e_timeline -Iinclude -Lflowgger_amd -lfg_hip -Wl,-rpath,'$ORIGIN/../../flowgger_amd'
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../flowgger_amd/host/fg_decoder.hpp"

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

struct Span {
    hipEvent_t a, b;
    float t0 = 0, t1 = 0;
};
static int mk(Span& s) { return hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess; }

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const std::string fmt = argv[1];
    const uint64_t slice_bytes = (uint64_t)(argc > 3 ? atof(argv[3]) : 32.0) << 20;
    const bool collect = argc > 4 && std::string(argv[4]) == "collect";
    std::unique_ptr<fg::Decoder> dec;
    if (fmt == "rfc5424") dec.reset(new fg::RFC5424Decoder());
    else if (fmt == "gelf") dec.reset(new fg::GelfDecoder());
    else {
        fg::LtsvConfig c;  // flowgger_amd.synth.LTSV_CONFIG
        c.schema = {{"counter", FG_T_U64}, {"score", FG_T_I64}, {"mean", FG_T_F64}, {"done", FG_T_BOOL}};
        c.suffix_u64 = "_u64";
        c.suffix_f64 = "_f64";
        dec.reset(new fg::LTSVDecoder(c));
    }
    // ---- the corpus: packed bytes + offsets in pinned memory
    std::ifstream in(argv[2], std::ios::binary);
    std::string raw((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    std::vector<uint64_t> offs{0};
    uint8_t* h_bytes = nullptr;
    CK(hipHostMalloc((void**)&h_bytes, raw.size() + 64, hipHostMallocDefault));
    uint64_t w = 0;
    for (size_t i = 0, s = 0; i <= raw.size(); ++i)
        if (i == raw.size() || raw[i] == '\n') {
            if (i > s) {
                memcpy(h_bytes + w, raw.data() + s, i - s);
                w += i - s;
                offs.push_back(w);
            }
            s = i + 1;
        }
    const uint64_t n = offs.size() - 1, nbytes = w;
    const uint32_t slices = (uint32_t)std::max<uint64_t>(1, (nbytes + slice_bytes - 1) / slice_bytes);
    std::vector<uint64_t> cut(slices + 1);
    if (fg_shard_plan(offs.data(), n, slices, cut.data()) != FG_OK) return 3;
    // The device entry point plans its launch from nbytes / n, so every slice is handed over as a batch of its own: bytes from the
    // slice's first 16-byte boundary, offsets relative to it (slice k's offsets live at [cut[k] + k, cut[k + 1] + k] of one array).
    uint64_t* h_offs = nullptr;
    CK(hipHostMalloc((void**)&h_offs, (n + slices + 1) * 8, hipHostMallocDefault));
    std::vector<uint64_t> b0s(slices), b1s(slices);
    for (uint32_t k = 0; k < slices; ++k) {
        b0s[k] = offs[cut[k]] & ~15ull;
        b1s[k] = offs[cut[k + 1]];
        for (uint64_t i = cut[k]; i <= cut[k + 1]; ++i) h_offs[i + k] = offs[i] - b0s[k];
    }
    // ---- device + host tables
    const uint64_t ent_cap = fmt == "rfc5424" ? nbytes / 16 + 1024 : nbytes / 8 + 1024;
    uint64_t sizes[FG_TABLE_ARRAYS];
    fg_tables_layout(n, ent_cap, sizes);
    uint64_t total = 0, at[FG_TABLE_ARRAYS];
    for (int k = 0; k < FG_TABLE_ARRAYS; ++k) {
        at[k] = total;
        total += (sizes[k] + 255) / 256 * 256;
    }
    uint8_t *d_tab = nullptr, *h_tab = nullptr, *d_bytes = nullptr;
    uint64_t* d_offs = nullptr;
    CK(hipMalloc((void**)&d_tab, total));
    CK(hipHostMalloc((void**)&h_tab, total, hipHostMallocDefault));
    CK(hipMalloc((void**)&d_bytes, nbytes + 64));
    CK(hipMalloc((void**)&d_offs, (n + slices + 1) * 8));
    memset(h_tab, 0, total);
    auto carve = [&](uint8_t* b) {
        fg_tables t{};
        t.n = n;
        t.ent_cap = ent_cap;
        void** f[] = {(void**)&t.meta, (void**)&t.ts, (void**)&t.hostname, (void**)&t.appname, (void**)&t.procid, (void**)&t.msgid,
                      (void**)&t.msg, (void**)&t.full_msg, (void**)&t.ent_first, (void**)&t.ent_count, (void**)&t.ent_name,
                      (void**)&t.ent_val, (void**)&t.ent_type, (void**)&t.ent_flags, (void**)&t.ent_used};
        for (int k = 0; k < FG_TABLE_ARRAYS; ++k) *f[k] = b + at[k];
        return t;
    };
    const fg_tables dt = carve(d_tab), ht = carve(h_tab);
    hipStream_t s_up, s_run, s_down;
    CK(hipStreamCreateWithFlags(&s_run, hipStreamNonBlocking));  // (created and used first, like the ctx's own stream)
    CK(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_down, hipStreamNonBlocking));
    {
        uint64_t probe = 0;
        CK(hipMemcpyAsync(d_offs, &probe, 8, hipMemcpyHostToDevice, s_run));
        CK(hipMemcpyAsync(&probe, d_offs, 8, hipMemcpyDeviceToHost, s_run));
        CK(hipStreamSynchronize(s_run));
    }
    auto slice_tables = [&](uint64_t l0, uint64_t rows) {
        fg_tables sl = dt;
        sl.n = rows;
        sl.meta += l0; sl.ts += l0; sl.hostname += l0; sl.appname += l0; sl.procid += l0; sl.msgid += l0; sl.msg += l0; sl.full_msg += l0;
        sl.ent_first += l0; sl.ent_count += l0;
        return sl;
    };
    // ---- dry run: entries per slice (the device entry point restarts the entry counter per call, so every slice's entries start at
    //      0 here -- the bytes that cross the link are the same)
    std::vector<uint64_t> ent_of(slices, 0);
    CK(hipMemcpy(d_bytes, h_bytes, nbytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_offs, h_offs, (n + slices + 1) * 8, hipMemcpyHostToDevice));
    for (uint32_t k = 0; k < slices; ++k) {
        const uint64_t l0 = cut[k], rows = cut[k + 1] - l0;
        if (!rows) continue;
        const fg_tables sl = slice_tables(l0, rows);
        if (fg_decode_batch_device(dec->ctx(), dec->format(), d_bytes + b0s[k], b1s[k] - b0s[k], d_offs + l0 + k, rows, &sl, (void*)s_run) != FG_OK) return 4;
        CK(hipMemcpyAsync(&ent_of[k], dt.ent_used, 8, hipMemcpyDeviceToHost, s_run));
        CK(hipStreamSynchronize(s_run));
    }
    uint64_t ent_total = 0;
    for (auto e : ent_of) ent_total += e;
    printf("%s: %llu lines, %.1f MB in, %u slices of <= %.0f MiB, %.1f MB of rows + %.1f MB of entries out, mode %s\n", fmt.c_str(),
           (unsigned long long)n, nbytes / 1e6, slices, slice_bytes / 1048576.0, n * 76 / 1e6, ent_total * 18 / 1e6, collect ? "collect" : "all");
    // ---- the timed run
    std::vector<Span> up(slices), run(slices), down(slices);
    std::vector<hipEvent_t> e_up(slices), e_run(slices);
    for (uint32_t k = 0; k < slices; ++k) {
        if (mk(up[k]) || mk(run[k]) || mk(down[k])) return 5;
        CK(hipEventCreateWithFlags(&e_up[k], hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&e_run[k], hipEventDisableTiming));
    }
    uint64_t* h_cnt = nullptr;
    CK(hipHostMalloc((void**)&h_cnt, slices * 8 + 8, hipHostMallocDefault));
    for (int rep = 0; rep < 2; ++rep) {  // (the second repetition is reported)
        CK(hipDeviceSynchronize());
        uint32_t issued = 0;
        auto issue = [&](uint32_t k) -> int {
            const uint64_t l0 = cut[k], l1 = cut[k + 1], rows = l1 - l0;
            CK(hipEventRecord(up[k].a, s_up));
            CK(hipMemcpyAsync(d_offs + l0 + k, h_offs + l0 + k, (l1 - l0 + 1) * 8, hipMemcpyHostToDevice, s_up));
            const uint64_t b0 = b0s[k], b1 = b1s[k];
            if (b1 > b0) CK(hipMemcpyAsync(d_bytes + b0, h_bytes + b0, b1 - b0, hipMemcpyHostToDevice, s_up));
            CK(hipEventRecord(up[k].b, s_up));
            CK(hipEventRecord(e_up[k], s_up));
            CK(hipStreamWaitEvent(s_run, e_up[k], 0));
            CK(hipEventRecord(run[k].a, s_run));
            if (rows) {
                const fg_tables sl = slice_tables(l0, rows);
                if (fg_decode_batch_device(dec->ctx(), dec->format(), d_bytes + b0, b1 - b0, d_offs + l0 + k, rows, &sl, (void*)s_run) != FG_OK) return 4;
            }
            CK(hipMemcpyAsync(h_cnt + k, dt.ent_used, 8, hipMemcpyDeviceToHost, s_run));
            CK(hipEventRecord(run[k].b, s_run));
            CK(hipEventRecord(e_run[k], s_run));
            return 0;
        };
        auto collect_slice = [&](uint32_t k, bool wait_host) -> int {
            const uint64_t l0 = cut[k], rows = cut[k + 1] - l0;
            if (wait_host) CK(hipEventSynchronize(e_run[k]));
            else CK(hipStreamWaitEvent(s_down, e_run[k], 0));
            CK(hipEventRecord(down[k].a, s_down));
            if (rows) {
                CK(hipMemcpyAsync(ht.meta + l0, dt.meta + l0, rows * 4, hipMemcpyDeviceToHost, s_down));
                CK(hipMemcpyAsync(ht.ts + l0, dt.ts + l0, rows * 8, hipMemcpyDeviceToHost, s_down));
                fg_span* hs[6] = {ht.hostname, ht.appname, ht.procid, ht.msgid, ht.msg, ht.full_msg};
                fg_span* ds[6] = {dt.hostname, dt.appname, dt.procid, dt.msgid, dt.msg, dt.full_msg};
                for (int j = 0; j < 6; ++j) CK(hipMemcpyAsync(hs[j] + l0, ds[j] + l0, rows * 8, hipMemcpyDeviceToHost, s_down));
                CK(hipMemcpyAsync(ht.ent_first + l0, dt.ent_first + l0, rows * 4, hipMemcpyDeviceToHost, s_down));
                CK(hipMemcpyAsync(ht.ent_count + l0, dt.ent_count + l0, rows * 4, hipMemcpyDeviceToHost, s_down));
            }
            const uint64_t e = ent_of[k];  // (from the dry run: the same volume the library brings back for this slice)
            uint64_t e0 = 0;
            for (uint32_t j = 0; j < k; ++j) e0 += ent_of[j];
            if (e) {
                CK(hipMemcpyAsync(ht.ent_name + e0, dt.ent_name + e0, e * 8, hipMemcpyDeviceToHost, s_down));
                CK(hipMemcpyAsync(ht.ent_val + e0, dt.ent_val + e0, e * 8, hipMemcpyDeviceToHost, s_down));
                CK(hipMemcpyAsync(ht.ent_type + e0, dt.ent_type + e0, e, hipMemcpyDeviceToHost, s_down));
                CK(hipMemcpyAsync(ht.ent_flags + e0, dt.ent_flags + e0, e, hipMemcpyDeviceToHost, s_down));
            }
            CK(hipEventRecord(down[k].b, s_down));
            return 0;
        };
        if (!collect) {
            for (uint32_t k = 0; k < slices; ++k)
                if (issue(k)) return 6;
            for (uint32_t k = 0; k < slices; ++k)
                if (collect_slice(k, false)) return 6;
        } else {
            for (uint32_t k = 0; k < slices; ++k) {
                while (issued < slices && issued < k + 16)
                    if (issue(issued++)) return 6;
                if (collect_slice(k, true)) return 6;
            }
        }
        CK(hipDeviceSynchronize());
    }
    for (uint32_t k = 0; k < slices; ++k) {
        CK(hipEventElapsedTime(&up[k].t0, up[0].a, up[k].a));
        CK(hipEventElapsedTime(&up[k].t1, up[0].a, up[k].b));
        CK(hipEventElapsedTime(&run[k].t0, up[0].a, run[k].a));
        CK(hipEventElapsedTime(&run[k].t1, up[0].a, run[k].b));
        CK(hipEventElapsedTime(&down[k].t0, up[0].a, down[k].a));
        CK(hipEventElapsedTime(&down[k].t1, up[0].a, down[k].b));
    }
    printf("slice   upload [ms]        kernels [ms]       download [ms]\n");
    for (uint32_t k = 0; k < slices; ++k)
        printf("%5u  %7.2f-%7.2f   %7.2f-%7.2f (%5.3f)   %7.2f-%7.2f\n", k, up[k].t0, up[k].t1, run[k].t0, run[k].t1, run[k].t1 - run[k].t0,
               down[k].t0, down[k].t1);
    double busy_up = 0, busy_run = 0, busy_down = 0, both = 0;
    for (uint32_t k = 0; k < slices; ++k) {
        busy_up += up[k].t1 - up[k].t0;
        busy_run += run[k].t1 - run[k].t0;
        busy_down += down[k].t1 - down[k].t0;
        for (uint32_t j = 0; j < slices; ++j) both += std::max(0.f, std::min(up[k].t1, down[j].t1) - std::max(up[k].t0, down[j].t0));
    }
    const float end = std::max(down[slices - 1].t1, run[slices - 1].t1);
    printf("total %.2f ms = %.1f M lines/s; busy: uploads %.2f ms, kernels %.2f ms (from the moment the slice's upload is done), "
           "downloads %.2f ms; an upload and a download in flight together for %.2f ms\n",
           end, n / (end * 1e-3) / 1e6, busy_up, busy_run, busy_down, both);
    return 0;
}
