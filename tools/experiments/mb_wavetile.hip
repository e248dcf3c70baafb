// Experiment (not part of the product): what does a pass cost when a tile is owned by ONE wavefront?
// 64 lanes x 64 amplitudes (complex64) = a 12-bit tile in 128 VGPRs: gates act on the 6 register-slot bits, layout
// changes go through a small wave-private LDS staging buffer (sub-tiles of 2^k registers x 64 lanes), so a pass has
// no workgroup barrier at all and every wave streams on its own.  Measures the skeleton (load / store), the skeleton +
// VALU work equivalent to G Hadamard-like gates, and + T staged LDS trips, at 2 / 3 waves per SIMD.
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/mb_wavetile tools/experiments/mb_wavetile.hip && /tmp/mb_wavetile
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct Args {
    const float4* in;
    float4* out;
    uint64_t wr_slot_off[6];   // (in float4 units) what slots 1..5 add to the write offset; [0] unused
    uint64_t wr_lane_off[6];   // what lane bits 0..5 add
    uint8_t wr_blk_pos[24];    // tile-number bit j -> output index bit (amplitudes)
    int nblk;
    int tiles_per_wave;
    float c0, c1;
};

template <int Q> __device__ __forceinline__ void hgate(v2f (&a)[64], const v2f m2) {
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        if ((j >> Q) & 1) continue;
        v2f& A = a[j];
        v2f& B = a[j | (1 << Q)];
        A = A + B;
        B = __builtin_elementwise_fma(B, m2, A);
    }
}

// staged trip: the top K slots trade places with K lane bits (here: lane bits 6-K .. 5), group by group through a
// wave-private buffer of 2^K x (64 + pad) eight-byte slots
template <int K> __device__ __forceinline__ void trip(v2f (&a)[64], unsigned lds_base, unsigned lane) {
    constexpr int G = 64 >> K, NR = 1 << K, A0 = 6 - K;
    constexpr unsigned STRIDE = 64 + (1u << A0);
    constexpr unsigned NBUF = K <= 3 ? 2 : 1;
    // write: slot = j * STRIDE + lane ; read: lane' = (lane & low) | (j' << A0), j = lane >> A0
    const unsigned wbase = lds_base + 8u * lane;
    const unsigned rbase = lds_base + 8u * ((lane >> A0) * STRIDE + (lane & ((1u << A0) - 1u)));
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int j = 0; j < NR; ++j)
            *(__attribute__((address_space(3))) v2f*)(uintptr_t)(wbase + 8u * (j * STRIDE) + (g % NBUF) * 8u * NR * STRIDE) = a[g + G * j];
#pragma unroll
        for (int j = 0; j < NR; ++j)
            a[g + G * j] = *(__attribute__((address_space(3))) v2f*)(uintptr_t)(rbase + 8u * (j << A0) + (g % NBUF) * 8u * NR * STRIDE);
    }
}

__host__ __device__ constexpr unsigned wave_lds_bytes(int k) { return (k <= 3 ? 2u : 1u) * 8u * (1u << k) * (64u + (1u << (6 - k))); }

template <int WPS, int GATES, int TRIPS, int K, int NT = 0>
__global__ __launch_bounds__(256, WPS) void pass_kernel(const Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned lds_base = wave * wave_lds_bytes(K);
    uint64_t wave_id = (uint64_t)blockIdx.x * 4u + wave;
    // load layout: slot 0 = index bit 0 (the float4), lanes = index bits 1..6, slots 1..5 = index bits 7..11
    uint64_t wlane = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
        if ((lane >> i) & 1u) wlane += p.wr_lane_off[i];
    v2f a[64];
    const v2f m2 = {p.c0, p.c0};
    for (int t = 0; t < p.tiles_per_wave; ++t) {
        const uint64_t tile = wave_id * (uint64_t)p.tiles_per_wave + t;
        const float4* src = p.in + tile * 2048u + lane;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            float4 v;
            if (NT & 1) { const v4f t_ = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src + 64 * j)); v = float4{t_.x, t_.y, t_.z, t_.w}; }
            else v = src[64 * j];
            a[2 * j] = v2f{v.x, v.y};
            a[2 * j + 1] = v2f{v.z, v.w};
        }
        if constexpr (GATES > 0) {
#pragma unroll 1
            for (int g = 0; g < GATES / 6; ++g) {
                hgate<0>(a, m2); hgate<1>(a, m2); hgate<2>(a, m2); hgate<3>(a, m2); hgate<4>(a, m2); hgate<5>(a, m2);
                if constexpr (TRIPS > 0) {
                    if (g < TRIPS) trip<K>(a, lds_base, lane);
                }
            }
        }
        uint64_t wt = 0;
        for (int b = 0; b < p.nblk; ++b) wt |= ((tile >> b) & 1ull) << p.wr_blk_pos[b];
        float4* dst = p.out + (wt >> 1) + wlane;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            uint64_t o = 0;
#pragma unroll
            for (int s = 1; s < 6; ++s)
                if ((j >> (s - 1)) & 1) o += p.wr_slot_off[s];
            const float4 w_ = float4{a[2 * j].x * p.c1, a[2 * j].y * p.c1, a[2 * j + 1].x * p.c1, a[2 * j + 1].y * p.c1};
            if (NT & 2) __builtin_nontemporal_store(v4f{w_.x, w_.y, w_.z, w_.w}, reinterpret_cast<v4f*>(dst + o)); else dst[o] = w_;
        }
    }
    (void)smem;
}

template <int WPS, int GATES, int TRIPS, int K, int NT = 0>
static void run(const char* name, Args a, int nbits, int tiles_per_wave) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    a.tiles_per_wave = tiles_per_wave;
    const uint64_t tiles = 1ull << (nbits - 12);
    const unsigned grid = (unsigned)(tiles / tiles_per_wave / 4);
    const size_t lds = WPS == 2 ? 60u * 1024u : 4u * wave_lds_bytes(K);   // 2 waves per SIMD: LDS padded so that two workgroups fit a CU
    hipFuncSetAttribute(reinterpret_cast<const void*>(&pass_kernel<WPS, GATES, TRIPS, K, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float best = 1e9;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((pass_kernel<WPS, GATES, TRIPS, K, NT>), dim3(grid), dim3(256), lds, 0, a);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    const double bytes = 2.0 * 8.0 * (double)(1ull << nbits);
    hipError_t err = hipGetLastError();
    printf("%-44s wps=%d gates=%2d trips=%d k=%d tpw=%2d  %7.3f ms  %6.0f GB/s  %.3f of 8 TB/s  %s\n", name, WPS, GATES, TRIPS, K,
           tiles_per_wave, best, bytes / best / 1e6, bytes / best / 1e6 / 8000.0, err == hipSuccess ? "" : hipGetErrorString(err));
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int nbits = argc > 1 ? atoi(argv[1]) : 32;      // 2^32 amplitudes = the headline state (n = 28, batch 16)
    const size_t bytes = 8ull << nbits;
    float4 *in, *out;
    if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, bytes) != hipSuccess) {
        printf("alloc failed\n");
        return 1;
    }
    hipMemset(in, 0, bytes);
    hipMemset(out, 0, bytes);
    Args a{};
    a.in = in;
    a.out = out;
    a.c0 = -2.0f;
    a.c1 = 0.5f;
    // write side: tile bits 1..6 (lanes) -> output bits 1..6 (1 KiB contiguous per store instruction), the five slot bits
    // -> scattered output bits, as a permuted store leaves them; tile-number bits fill the remaining positions in order
    struct Pat { const char* name; int lane_pos[6]; int slot_pos[5]; };
    const Pat pats[] = {{"write: lanes 1..6, slots 7..11 = plain copy", {1, 2, 3, 4, 5, 6}, {7, 8, 9, 10, 11}},
                        {"write: lanes 1..6 (1 KiB runs), slots 12,15,18,21,24", {1, 2, 3, 4, 5, 6}, {12, 15, 18, 21, 24}},
                        {"write: lanes 1..4,13,14 (256 B runs), slots 12,16,17,18,19", {1, 2, 3, 4, 13, 14}, {12, 16, 17, 18, 19}},
                        {"write: lanes 1..3,13,14,15 (128 B runs), slots 12,16,17,18,19", {1, 2, 3, 13, 14, 15}, {12, 16, 17, 18, 19}},
                        {"write: lanes 1..3,13,17,18 (128 B runs), slots 12,14,15,16,19", {1, 2, 3, 13, 17, 18}, {12, 14, 15, 16, 19}},
                        {"write: lanes 1..5,13 (512 B runs), slots 12,16,17,18,19", {1, 2, 3, 4, 5, 13}, {12, 16, 17, 18, 19}}};
    for (const Pat& pt : pats) {
        uint64_t used = 1;
        for (int i = 0; i < 6; ++i) {
            a.wr_lane_off[i] = (1ull << pt.lane_pos[i]) >> 1;       // float4 units
            used |= 1ull << pt.lane_pos[i];
        }
        for (int s = 1; s < 6; ++s) {
            a.wr_slot_off[s] = (1ull << pt.slot_pos[s - 1]) >> 1;
            used |= 1ull << pt.slot_pos[s - 1];
        }
        a.nblk = nbits - 12;
        for (int b = 0, q = 0; b < a.nblk; ++b, ++q) {
            while ((used >> q) & 1ull) ++q;
            a.wr_blk_pos[b] = (uint8_t)q;
        }
        printf("# %s\n", pt.name);
        run<3, 0, 0, 3>("skeleton", a, nbits, 1);
        run<3, 48, 3, 3>("skeleton + 48 gates + 3 staged trips", a, nbits, 1);
        run<3, 72, 4, 3>("skeleton + 72 gates + 4 staged trips", a, nbits, 1);
        run<3, 96, 4, 3>("skeleton + 96 gates + 4 staged trips", a, nbits, 1);
        run<2, 72, 4, 3>("skeleton + 72 gates + 4 staged trips", a, nbits, 1);
        run<3, 0, 0, 3, 1>("skeleton, nt loads", a, nbits, 1);
        run<3, 0, 0, 3, 2>("skeleton, nt stores", a, nbits, 1);
        run<3, 0, 0, 3, 3>("skeleton, nt loads + stores", a, nbits, 1);
        run<3, 72, 4, 3, 3>("72 gates + 4 trips, nt loads + stores", a, nbits, 1);
    }
    return 0;
}
