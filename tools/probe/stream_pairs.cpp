// Which pairs of HIP streams carry an H2D and a D2H copy AT THE SAME TIME on this box?  (The host paths of libfg_hip put uploads and
// downloads on different streams; measured in round 3: one pair gave 97 GB/s for both directions together, another 57 GB/s -- the
// copies serialised.)  Creates N non-blocking streams (+ the null stream) and times H2D on stream a with D2H on stream b.
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/stream_pairs.cpp -o gpurun_out/stream_pairs
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

int main() {
    const size_t n = 512u << 20;
    void *h0, *h1, *d0, *d1;
    hipHostMalloc(&h0, n, hipHostMallocDefault);
    hipHostMalloc(&h1, n, hipHostMallocDefault);
    hipMalloc(&d0, n);
    hipMalloc(&d1, n);
    memset(h0, 1, n);
    memset(h1, 0, n);
    const int N = 6;
    std::vector<hipStream_t> s(N);
    for (int i = 0; i < N; ++i) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
    auto run = [&](int a, int b) {
        double best = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            hipMemcpyAsync(d0, h0, n, hipMemcpyHostToDevice, s[a]);
            if (b >= 0) hipMemcpyAsync(h1, d1, n, hipMemcpyDeviceToHost, s[b]);
            hipStreamSynchronize(s[a]);
            if (b >= 0) hipStreamSynchronize(s[b]);
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            double g = (b >= 0 ? 2.0 : 1.0) * n / dt / 1e9;
            if (g > best) best = g;
        }
        return best;
    };
    printf("H2D alone on s0: %.1f GB/s\n", run(0, -1));
    for (int a = 0; a < N; ++a) {
        for (int b = 0; b < N; ++b)
            if (a != b) printf("up s%d + down s%d: %6.1f   ", a, b, run(a, b));
        printf("\n");
    }
    // the same pair again after a kernel-free warm-up of every stream (is the mapping fixed at creation or at first use?)
    printf("again: s0+s1 %.1f, s1+s2 %.1f, s2+s0 %.1f, s4+s5 %.1f\n", run(0, 1), run(1, 2), run(2, 0), run(4, 5));
    return 0;
}
