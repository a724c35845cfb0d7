// Standalone A/B of the bf16-storage GEMM kernels (no torch): gemm_bf16_256_kernel (rounds 2-4) against the round-5
// ring kernel on the shapes of BASELINE configs[2], bit-for-bit comparison of the outputs + timing with HIP events.
// Operand data matters: the chip lowers its clock under the bf16 MFMA + DMA load, by an amount that depends on the data
// (f2: 593 us on dense random operands, 453 us on zeros) -- the default A operand is post-ReLU (half zeros) like the real
// activations; DENSE=1: dense random, ZERO=1: zeros.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/hw_probe/ring_probe.hip -o tools/hw_probe/ring_probe
//   ./tools/hw_probe/ring_probe [reps]
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <string>
#define GRL_RING_NO_CABI 1
#include "../../grl_amd/csrc/gemm_bf16.hip"
#include "../../grl_amd/csrc/gemm_bf16_ring.hip"

int grl_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fprintf(stderr, "\n");
    return code;
}
int grl_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return GRL_ELAUNCH; }
    return 0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_bf16(__bf16* p, size_t n, unsigned seed, float scale, float offset) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
        p[i] = (__bf16)(((h & 0xffff) / 32768.f - 1.f) * scale + offset);
    }
}
// post-ReLU activations: max(0, x) of a centred value -- half of the elements are exact zeros, as on the real path
__global__ void relu_bf16(__bf16* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = (float)p[i];
        p[i] = (__bf16)(v > 0.f ? v : 0.f);
    }
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float scale, float offset) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
        p[i] = ((h & 0xffff) / 32768.f - 1.f) * scale + offset;
    }
}

struct Shape {
    const char* name;
    int M, N, K;
    bool res, sqd, gbias;
    int conv_c;          // 3x3 conv over [n][16][8][C] (stride 1 pad 1) when > 0
    int groups;          // > 1: grouped launch of that many problems (new kernel only; old runs them one by one)
};

template <class F>
float time_ms(F&& f, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return ms / reps;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    const float dscale = getenv("ZERO") ? 0.f : 1.f;      // ZERO=1: all-zero operands (what the clock does without the data-dependent power)
    const char* only = argc > 2 ? argv[2] : nullptr;
    std::vector<Shape> shapes = {
        {"f2 65536x2048x2048", 65536, 2048, 2048, false, false, false, 0, 1},
        {"l4c2 65536x512x4608 3x3", 65536, 512, 4608, false, false, false, 512, 1},
        {"l3c2 65536x256x2304 3x3", 65536, 256, 2304, false, false, false, 256, 1},
        {"l4c3 65536x2048x512 +res", 65536, 2048, 512, true, false, false, 0, 1},
        {"l3c3 65536x1024x256 +res", 65536, 1024, 256, true, false, false, 0, 1},
        {"l3c1 65536x256x1024", 65536, 256, 1024, false, false, false, 0, 1},
        {"l4c1 65536x512x2048", 65536, 512, 2048, false, false, false, 0, 1},
        {"gce0 65536x1024x2048 gbias", 65536, 1024, 2048, false, false, true, 0, 1},
        {"l4dn 65536x2048x1024", 65536, 2048, 1024, false, false, false, 0, 1},
        {"f1 8192x2048x2048 sqd", 8192, 2048, 2048, true, true, false, 0, 1},
        {"f1x2 8192x2048x2048 sqd", 8192, 2048, 2048, true, true, false, 0, 2},
        {"mc1 8192x512x2048", 8192, 512, 2048, false, false, false, 0, 1},
        {"mc1x2 8192x512x2048", 8192, 512, 2048, false, false, false, 0, 2},
        {"mc2 8192x512x512", 8192, 512, 512, false, false, false, 0, 1},
        {"mc2x2 8192x512x512", 8192, 512, 512, false, false, false, 0, 2},
        {"mc3 8192x2048x512 +res", 8192, 2048, 512, true, false, false, 0, 1},
        {"mc3x2 8192x2048x512 +res", 8192, 2048, 512, true, false, false, 0, 2},
        {"edge 1000x520x192 +res", 1000, 520, 192, true, false, false, 0, 1},
    };
    printf("%-30s %10s %10s %10s %10s %10s %10s   (TFLOP/s; us)\n", "shape", "old256", "r256n4", "r256n5", "r256x128", "r128x256", "best");
    for (const Shape& sh : shapes) {
        if (only && !strstr(sh.name, only)) continue;
        const int G = sh.groups;
        const size_t M = sh.M, N = sh.N, K = sh.K;
        const int nimg = sh.conv_c ? sh.M / 128 : 0;
        const size_t a_elems = sh.conv_c ? (size_t)nimg * 128 * sh.conv_c : M * K;
        __bf16 *a[4], *w[4], *res[4], *y_old[4], *y_new[4];
        float *scale, *shift, *gb = nullptr;
        const size_t y_elems = sh.sqd ? (M / 32) * N * 2 : M * N;       // (fp32 partials: twice the bf16 elements)
        for (int g = 0; g < G; ++g) {
            CK(hipMalloc(&a[g], a_elems * 2));
            CK(hipMalloc(&w[g], N * K * 2));
            CK(hipMalloc(&res[g], M * N * 2));
            CK(hipMalloc(&y_old[g], y_elems * 2));
            CK(hipMalloc(&y_new[g], y_elems * 2));
            fill_bf16<<<1024, 256>>>(a[g], a_elems, 11u + g, dscale, getenv("DENSE") ? 0.3f * dscale : 0.f);
            if (!getenv("DENSE")) relu_bf16<<<1024, 256>>>(a[g], a_elems);
            fill_bf16<<<1024, 256>>>(w[g], N * K, 23u + g, dscale * 1.5f / sqrtf((float)K), 0.f);
            fill_bf16<<<1024, 256>>>(res[g], M * N, 37u + g, 1.f, 0.f);
        }
        CK(hipMalloc(&scale, N * 4));
        CK(hipMalloc(&shift, N * 4));
        fill_f32<<<64, 256>>>(scale, N, 5u, 0.5f, 1.f);
        fill_f32<<<64, 256>>>(shift, N, 7u, 0.2f, 0.f);
        if (sh.gbias) {
            CK(hipMalloc(&gb, (M / 1024) * N * 4));
            fill_f32<<<64, 256>>>(gb, (M / 1024) * N, 9u, 0.3f, 0.f);
        }
        CK(hipDeviceSynchronize());
        auto desc = [&](int g, __bf16* y) {
            GrlGemm d;
            memset(&d, 0, sizeof(d));
            d.a = (const float*)a[g];
            d.w = (const float*)w[g];
            d.y = (float*)y;
            d.scale = scale;
            d.shift = shift;
            d.res = sh.res ? (const float*)res[g] : nullptr;
            d.gbias = gb;
            d.rows_per_group = 1024;
            d.M = sh.M; d.N = sh.N; d.K = sh.K;
            d.lda = sh.K; d.ldw = sh.K; d.ldy = sh.N; d.ldres = sh.N;
            d.relu = 1;
            d.epilogue = sh.sqd ? GRL_EPI_SQDIFF : GRL_EPI_AFFINE;
            d.math = GRL_MATH_BF16S;
            if (sh.sqd) { d.res_rows = 128; d.res_gstride = 128; d.scale = nullptr; }
            if (sh.conv_c) {
                d.conv = 1; d.H = 16; d.W = 8; d.C = sh.conv_c; d.Ho = 16; d.Wo = 8; d.kh = d.kw = 3; d.stride = 1; d.pad = 1;
                d.lda = 0;
            }
            return d;
        };
        grl_gemm_bf16_tile_mode(1);
        bool old_ok = true;
        auto run_old = [&]() {
            for (int g = 0; g < G; ++g) {
                GrlGemm d = desc(g, y_old[g]);
                if (grl_gemm_bf16_256(d, 0) != 1) old_ok = false;
            }
        };
        auto run_new = [&](int variant) {
            GrlGemm d = desc(0, y_new[0]);
            ring::Group grp;
            memset(&grp, 0, sizeof(grp));
            grp.n = G;
            for (int g = 0; g < G; ++g) {
                grp.a[g] = a[g]; grp.w[g] = w[g]; grp.y[g] = y_new[g]; grp.scale[g] = d.scale; grp.shift[g] = d.shift;
                grp.res[g] = sh.res ? res[g] : nullptr;
            }
            return ring::launch_variant(d, grp, 0, variant);
        };
        const double flop = 2.0 * M * N * K * G;
        run_old();
        CK(hipDeviceSynchronize());
        float t_old = old_ok ? time_ms(run_old, reps) : 0.f;
        std::vector<char> h_old(y_elems * 2), h_new(y_elems * 2);
        float t_new[4] = {0, 0, 0, 0};
        char status[4] = {'-', '-', '-', '-'};
        for (int v = 0; v < 4; ++v) {
            for (int g = 0; g < G; ++g) CK(hipMemset(y_new[g], 0xff, y_elems * 2));
            const int rc = run_new(v);
            if (rc != 0) { (void)hipGetLastError(); continue; }
            if (hipDeviceSynchronize() != hipSuccess) { printf("variant %d: device error\n", v); return 1; }
            status[v] = '=';
            for (int g = 0; g < G && old_ok; ++g) {
                CK(hipMemcpy(h_old.data(), y_old[g], y_elems * 2, hipMemcpyDeviceToHost));
                CK(hipMemcpy(h_new.data(), y_new[g], y_elems * 2, hipMemcpyDeviceToHost));
                if (memcmp(h_old.data(), h_new.data(), y_elems * 2) != 0) {
                    status[v] = 'X';
                    size_t bad = 0, first = (size_t)-1;
                    for (size_t i = 0; i < y_elems; ++i)
                        if (((uint16_t*)h_old.data())[i] != ((uint16_t*)h_new.data())[i]) { if (first == (size_t)-1) first = i; ++bad; }
                    fprintf(stderr, "  %s variant %d group %d: %zu of %zu elements differ (first at %zu = row %zu col %zu)\n", sh.name, v, g,
                            bad, y_elems, first, first / (sh.sqd ? 2 * N : N), first % (sh.sqd ? 2 * N : N));
                }
            }
            t_new[v] = time_ms([&] { run_new(v); }, reps);
        }
        float best = 1e9f;
        for (int v = 0; v < 4; ++v) if (t_new[v] > 0 && t_new[v] < best) best = t_new[v];
        auto tf = [&](float ms) { return ms > 0 ? flop / (ms * 1e-3) * 1e-12 : 0.0; };
        printf("%-30s %6.0f %5.0f %5.0f%c%5.0f %5.0f%c%5.0f %5.0f%c%5.0f %5.0f%c%5.0f %6.0f %5.0f\n", sh.name, tf(t_old), t_old * 1e3,
               tf(t_new[0]), status[0], t_new[0] * 1e3, tf(t_new[1]), status[1], t_new[1] * 1e3, tf(t_new[2]), status[2], t_new[2] * 1e3,
               tf(t_new[3]), status[3], t_new[3] * 1e3, tf(best), best * 1e3);
        fflush(stdout);
        for (int g = 0; g < G; ++g) { hipFree(a[g]); hipFree(w[g]); hipFree(res[g]); hipFree(y_old[g]); hipFree(y_new[g]); }
        hipFree(scale); hipFree(shift);
        if (gb) hipFree(gb);
    }
    return 0;
}
