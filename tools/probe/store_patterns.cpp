// tools/probe/store_patterns.cpp -- what the encoder's WRITE pass can hope for (one-pass encoder).
// Every wave owns 64 consecutive "messages" (sizes ~ U[500, 700] bytes, packed back to back like the encoders' output) and writes
// them with one of the store patterns below; nothing is read, so the figure is the store path alone, at the occupancy the real
// kernel has (LDS-limited: `lds` dynamic bytes per wave).  GB/s of output per pattern:
//   0  lane-per-message, aligned dword stores          (emit::PackSink today: 64 cache lines touched per store instruction)
//   1  lane-per-message, aligned 16-byte stores
//   2  lane-per-message through a 64-byte LDS slot per lane, flushed by the wave: 4 lanes per slot, 16 slots per instruction
//   3  lane-per-message through a 256-byte LDS slot per lane, flushed by the wave: 16 lanes per slot
//   4  fully coalesced (the wave's whole range, 1 KiB per instruction): the ceiling
//   5  as 1, but every store instruction executes with a QUARTER of the lanes (lane & 3 == step & 3) -- four times the instructions
//      for the same bytes: what a sink does whose lanes complete their 16-byte blocks at different put_word calls
//   6  as 5 with an EIGHTH of the lanes
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/store_patterns tools/probe/store_patterns.cpp
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(64) void k_store(const uint64_t* __restrict__ off, uint64_t n, uint8_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t lane = threadIdx.x;
    const uint64_t g0 = (uint64_t)blockIdx.x * 64u;
    const uint64_t li = g0 + lane < n ? g0 + lane : n - 1u;
    const uint64_t o0 = off[li], o1 = off[li + 1];
    const uint64_t w0 = off[g0], w1 = off[g0 + 64u < n ? g0 + 64u : n];
    if (MODE == 0) {
        uint64_t a = (o0 + 3u) & ~3ull, e = o1 & ~3ull;
        uint32_t v = (uint32_t)o0;
        for (; a < e; a += 4u) {
            *reinterpret_cast<uint32_t*>(out + a) = v;
            v = v * 1664525u + 1013904223u;
        }
    } else if (MODE == 1) {
        uint64_t a = (o0 + 15u) & ~15ull, e = o1 & ~15ull;
        uint32_t v = (uint32_t)o0;
        for (; a < e; a += 16u) {
            u32x4 q = {v, v + 1u, v + 2u, v + 3u};
            *reinterpret_cast<u32x4*>(out + a) = q;
            v = v * 1664525u + 1013904223u;
        }
    } else if (MODE == 5 || MODE == 6) {
        constexpr uint32_t PARTS = MODE == 5 ? 4u : 8u;
        uint64_t a = (o0 + 15u) & ~15ull;
        const uint64_t e = o1 & ~15ull;
        uint32_t v = (uint32_t)o0;
        for (uint32_t step = 0; __any(a < e); ++step) {
            if ((lane % PARTS) == (step % PARTS) && a < e) {
                u32x4 q = {v, v + 1u, v + 2u, v + 3u};
                *reinterpret_cast<u32x4*>(out + a) = q;
                v = v * 1664525u + 1013904223u;
                a += 16u;
            }
        }
    } else if (MODE == 2 || MODE == 3) {
        constexpr uint32_t SLOT = MODE == 2 ? 64u : 256u;   // bytes per lane and round
        constexpr uint32_t LPS = SLOT / 16u;                // lanes that flush one slot
        constexpr uint32_t SPI = 64u / LPS;                 // slots per flush instruction
        u32x4* slots = reinterpret_cast<u32x4*>(lds);
        uint64_t a = (o0 + 15u) & ~15ull;
        const uint64_t e = o1 & ~15ull;
        uint32_t v = (uint32_t)o0;
        // rounds: every lane fills its slot (SLOT / 16 iterations of 16 bytes), then the wave flushes all 64 slots
        for (;;) {
            const bool any = __any(a < e);
            if (!any) break;
            const uint64_t a_round = a;
            uint32_t filled = 0;
#pragma unroll
            for (uint32_t k = 0; k < LPS; ++k) {
                if (a < e) {
                    u32x4 q = {v, v + 1u, v + 2u, v + 3u};
                    slots[lane * LPS + k] = q;
                    v = v * 1664525u + 1013904223u;
                    a += 16u;
                    filled += 16u;
                }
            }
            __syncthreads();
            // flush: slot s is written by lanes (s % SPI) * LPS .. + LPS - 1 of instruction s / SPI
#pragma unroll
            for (uint32_t i = 0; i < 64u / SPI; ++i) {
                const uint32_t s = i * SPI + lane / LPS, part = lane % LPS;
                const uint64_t sa = __shfl(a_round, s, 64);
                const uint32_t sf = __shfl(filled, s, 64);
                if (part * 16u < sf) *reinterpret_cast<u32x4*>(out + sa + part * 16u) = slots[s * LPS + part];
            }
            __syncthreads();
        }
    } else {
        uint64_t a = ((w0 + 15u) & ~15ull) + lane * 16u;
        const uint64_t e = w1 & ~15ull;
        uint32_t v = (uint32_t)o0;
        for (; a < e; a += 1024u) {
            u32x4 q = {v, v + 1u, v + 2u, v + 3u};
            *reinterpret_cast<u32x4*>(out + a) = q;
            v = v * 1664525u + 1013904223u;
        }
    }
}

int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4000000ull;
    const uint32_t lds = argc > 2 ? (uint32_t)atoi(argv[2]) : 22 * 1024u;  // 7 waves per CU, like the encoders
    std::vector<uint64_t> off(n + 1);
    uint64_t s = 0;
    uint32_t r = 12345u;
    for (uint64_t i = 0; i < n; ++i) {
        off[i] = s;
        r = r * 1664525u + 1013904223u;
        s += 500u + (r >> 8) % 201u;
    }
    off[n] = s;
    uint64_t* d_off;
    uint8_t* d_out;
    CHECK(hipMalloc(&d_off, (n + 1) * 8));
    CHECK(hipMalloc(&d_out, s + 4096));
    CHECK(hipMemcpy(d_off, off.data(), (n + 1) * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const uint32_t blocks = (uint32_t)((n + 63) / 64);
    printf("%llu messages, %.1f MB of output, %u B of LDS per wave\n", (unsigned long long)n, s / 1e6, lds);
    for (int mode = 0; mode < 7; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0));
            switch (mode) {
                case 0: hipLaunchKernelGGL(k_store<0>, dim3(blocks), dim3(64), lds, 0, d_off, n, d_out); break;
                case 1: hipLaunchKernelGGL(k_store<1>, dim3(blocks), dim3(64), lds, 0, d_off, n, d_out); break;
                case 2: hipLaunchKernelGGL(k_store<2>, dim3(blocks), dim3(64), lds, 0, d_off, n, d_out); break;
                case 3: hipLaunchKernelGGL(k_store<3>, dim3(blocks), dim3(64), lds < 16384u ? 16384u : lds, 0, d_off, n, d_out); break;
                case 4: hipLaunchKernelGGL(k_store<4>, dim3(blocks), dim3(64), lds, 0, d_off, n, d_out); break;
                case 5: hipLaunchKernelGGL(k_store<5>, dim3(blocks), dim3(64), lds, 0, d_off, n, d_out); break;
                default: hipLaunchKernelGGL(k_store<6>, dim3(blocks), dim3(64), lds, 0, d_off, n, d_out); break;
            }
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("mode %d: %8.3f ms  %8.1f GB/s\n", mode, best, s / (best * 1e-3) / 1e9);
    }
    return 0;
}
