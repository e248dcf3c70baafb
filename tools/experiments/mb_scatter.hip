// Memory-side experiment (not part of the product): a pass of the fused kernel reads a tile whose high bits are
// GATHERED from far-apart addresses and writes it back to the same places.  If a pass could write its tile to other
// index bits than it read it from (a bit-permuted, out-of-place store), every pass could gather only near bits and
// leave the far-apart traffic to the write side.  Is a scattered write cheaper than a scattered read?
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/mb_scatter tools/experiments/mb_scatter.hip && /tmp/mb_scatter
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int M = 13, L = 4, H = M - L;   // tile bits, contiguous low bits, gathered bits (as the product's c64 tile)

struct Map {
    uint64_t off[M];      // amplitude offset of tile bit i (i < L: 1 << i)
    uint8_t sorted[H];    // gathered positions ascending (for the block base)
};

__device__ inline uint64_t insert_zero(uint64_t x, int pos) {
    const uint64_t lo = x & ((1ull << pos) - 1);
    return ((x >> pos) << (pos + 1)) | lo;
}

__global__ __launch_bounds__(512) void copy_tiles(const float2* in, float2* out, Map rd, Map wr, int n) {
    const unsigned tid = threadIdx.x;
    uint64_t tr = (uint64_t)blockIdx.x << L, tw = (uint64_t)blockIdx.x << L;
    for (int i = 0; i < H; ++i) { tr = insert_zero(tr, rd.sorted[i]); tw = insert_zero(tw, wr.sorted[i]); }
    const uint64_t sample = (uint64_t)blockIdx.y << n;
    // thread bits = tile bits 1..9, register slots = tile bit 0 (the 16-byte vector) and tile bits 10..12
    uint64_t orr = 0, ow = 0;
    for (int i = 0; i < 9; ++i) if ((tid >> i) & 1u) { orr += rd.off[1 + i]; ow += wr.off[1 + i]; }
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint64_t o = orr;
        for (int s = 0; s < 3; ++s) if ((j >> s) & 1) o += rd.off[10 + s];
        v[j] = *reinterpret_cast<const float4*>(in + sample + tr + o);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint64_t o = ow;
        for (int s = 0; s < 3; ++s) if ((j >> s) & 1) o += wr.off[10 + s];
        *reinterpret_cast<float4*>(out + sample + tw + o) = v[j];
    }
}

static Map make_map(std::vector<int> gathered) {
    Map m{};
    for (int i = 0; i < L; ++i) m.off[i] = 1ull << i;
    for (int i = 0; i < H; ++i) m.off[L + i] = 1ull << gathered[i];
    std::sort(gathered.begin(), gathered.end());
    for (int i = 0; i < H; ++i) m.sorted[i] = (uint8_t)gathered[i];
    return m;
}

int main() {
    const int n = 28, batch = 4;
    const size_t bytes = sizeof(float2) * ((size_t)batch << n);
    float2 *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    const std::vector<int> near = {4, 5, 6, 7, 8, 9, 10, 11, 12};
    const std::vector<int> mid = {10, 11, 12, 13, 14, 15, 16, 17, 18};
    const std::vector<int> far4 = {6, 9, 12, 15, 18, 20, 22, 24, 26};
    const std::vector<int> far9 = {19, 20, 21, 22, 23, 24, 25, 26, 27};
    struct Case { const char* name; std::vector<int> r, w; bool inplace; };
    const std::vector<Case> cases = {
        {"read near  / write near  (in place)", near, near, true},  {"read mid   / write mid   (in place)", mid, mid, true},
        {"read far4  / write far4  (in place)", far4, far4, true},  {"read far9  / write far9  (in place)", far9, far9, true},
        {"read far4  / write far4  (out of place)", far4, far4, false},
        {"read near  / write far4  (out of place)", near, far4, false}, {"read far4  / write near  (out of place)", far4, near, false},
        {"read mid   / write far4  (out of place)", mid, far4, false},  {"read far4  / write mid   (out of place)", far4, mid, false},
        {"read near  / write far9  (out of place)", near, far9, false}, {"read far9  / write near  (out of place)", far9, near, false},
        {"read mid   / write mid   (out of place)", mid, mid, false},
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("# n=%d batch=%d complex64, tile 2^%d: %.2f GB read + %.2f GB written per pass\n", n, batch, M, bytes / 1e9, bytes / 1e9);
    for (const auto& c : cases) {
        const Map r = make_map(c.r), w = make_map(c.w);
        dim3 grid(1u << (n - M), batch);
        float2* dst = c.inplace ? a : b;
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(copy_tiles, grid, dim3(512), 0, 0, a, dst, r, w, n);
        hipEventRecord(e0);
        const int reps = 5;
        for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(copy_tiles, grid, dim3(512), 0, 0, a, dst, r, w, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        printf("%-44s %7.3f ms  %6.0f GB/s\n", c.name, ms, 2.0 * bytes / ms / 1e6);
    }
    return 0;
}
