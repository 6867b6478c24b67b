// CPU test of the framers' flush policy (flowgger_amd/host/fg_decoder.hpp: ByteSource / FlushPolicy / BufferedSource) without a GPU:
// frames STDIN on '\n' exactly as fg::BatchingSplitter does, with a counting stand-in for the GPU call.  Prints
//   FLUSH <lines> <ms since start>   whenever the policy hands a batch over
//   CHUNK <bytes> <ms>               (mode "chunk": what GpuFramingSplitter / TranscodingSplitter would send to the GPU)
//   END eof|idle|error
// usage: framer_policy_test <line|chunk> <idle_timeout_ms> <max_latency_ms> [linger_below]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../flowgger_amd/host/fg_decoder.hpp"

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const std::string mode = argv[1];
    fg::FdSource src(0);
    fg::FlushPolicy pol;
    pol.idle_timeout_ms = atoi(argv[2]);
    pol.max_latency_ms = atoi(argv[3]);
    if (argc > 4) pol.linger_below = (size_t)atoi(argv[4]);
    const auto t0 = std::chrono::steady_clock::now();
    auto ms = [&] { return (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count(); };
    fg::BufferedSource in(src, pol);
    if (mode == "chunk") {
        std::vector<uint8_t> buf;
        for (;;) {
            const bool got = in.read_chunk(buf, 1 << 20);
            if (got) {
                printf("CHUNK %zu %ld\n", buf.size(), ms());
                fflush(stdout);
                buf.clear();
            }
            if (in.end() != fg::BufferedSource::None) break;
        }
    } else {
        size_t pending_lines = 0, pending_bytes = 0;
        auto flush = [&] {
            if (!pending_lines) return;
            printf("FLUSH %zu %ld\n", pending_lines, ms());
            fflush(stdout);
            pending_lines = pending_bytes = 0;
        };
        in.on_block([&] { return pending_bytes + pending_lines; }, flush);
        std::string line;
        for (;;) {
            line.clear();
            if (!in.read_until('\n', line)) {
                if (in.end() == fg::BufferedSource::Eof && !line.empty()) {
                    ++pending_lines;
                    pending_bytes += line.size();
                }
                break;
            }
            ++pending_lines;
            pending_bytes += line.size();
            if (pending_lines >= pol.max_lines) flush();
        }
        flush();
    }
    printf("END %s\n", in.end() == fg::BufferedSource::Eof ? "eof" : in.end() == fg::BufferedSource::Idle ? "idle" : "error");
    return 0;
}
