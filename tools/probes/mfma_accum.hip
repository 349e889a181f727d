// Probe: how does v_mfma_f32_32x32x1_2b_f32 round D = C + a * b when C != 0 -- as one fused multiply-add, or as a rounded
// product followed by a rounded sum (what the reference's SSE mul + add does)?  Also the k = 2 form (32x32x2): order and
// rounding of the two products.
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o mfma_accum mfma_accum.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <cmath>
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k1(const float *a, const float *b, const float *c, float *dm)
{   // D[blk][i][j] = c[blk][i][j] + a[blk][i] * b[blk][j]
    const int lane = threadIdx.x;
    const float av = a[blockIdx.x * 64 + lane], bv = b[blockIdx.x * 64 + lane];
    f32x32 cc;
    for (int e = 0; e < 32; ++e) {
        const int blk = e >> 4, i = (e & 3) + 8 * ((e & 15) >> 2) + 4 * (lane >> 5), j = lane & 31;
        cc[e] = c[((size_t)blockIdx.x * 2 + blk) * 1024 + i * 32 + j];
    }
    f32x32 d = __builtin_amdgcn_mfma_f32_32x32x1f32(av, bv, cc, 0, 0, 0);
    for (int e = 0; e < 32; ++e) {
        const int blk = e >> 4, i = (e & 3) + 8 * ((e & 15) >> 2) + 4 * (lane >> 5), j = lane & 31;
        dm[((size_t)blockIdx.x * 2 + blk) * 1024 + i * 32 + j] = d[e];
    }
}
__global__ void k2(const float *a, const float *b, const float *c, float *dm)
{   // 32x32x2: A[i][k] from lane i + 32 k, B[k][j] from lane j + 32 k; D[i][j] = c + a[i][0] b[0][j] + a[i][1] b[1][j]
    const int lane = threadIdx.x;
    const float av = a[blockIdx.x * 64 + lane], bv = b[blockIdx.x * 64 + lane];
    f32x16 cc;
    for (int e = 0; e < 16; ++e) {
        const int i = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), j = lane & 31;
        cc[e] = c[(size_t)blockIdx.x * 2048 + i * 32 + j];
    }
    f32x16 d = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, cc, 0, 0, 0);
    for (int e = 0; e < 16; ++e) {
        const int i = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), j = lane & 31;
        dm[(size_t)blockIdx.x * 2048 + i * 32 + j] = d[e];
    }
}
static uint64_t sm(uint64_t &s) { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
static bool same(float x, float y) { uint32_t a, b; memcpy(&a, &x, 4); memcpy(&b, &y, 4); return a == b; }
int main()
{
    const int NB = 2048;
    std::vector<float> a(NB * 64), b(NB * 64), c((size_t)NB * 2048);
    uint64_t s = 7;
    for (size_t i = 0; i < a.size(); ++i) {
        a[i] = (float)((sm(s) >> 40) * (1.0 / 16777216.0)) * 0.5f;                        // HOG feature range
        b[i] = ((int)(sm(s) >> 40) - 8388608) * (1.0f / 8388608.0f) * 0.2f;               // weight range
    }
    for (size_t i = 0; i < c.size(); ++i) c[i] = ((int)(sm(s) >> 40) - 8388608) * (1.0f / 8388608.0f) * ((i & 1) ? 0.3f : 0.01f);
    float *da, *db, *dc, *dm;
    hipMalloc(&da, a.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dc, c.size() * 4); hipMalloc(&dm, c.size() * 4);
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> hm(c.size());
    {
        hipLaunchKernelGGL(k1, dim3(NB), dim3(64), 0, 0, da, db, dc, dm);
        hipMemcpy(hm.data(), dm, hm.size() * 4, hipMemcpyDeviceToHost);
        size_t fused = 0, unfused = 0, both = 0, neither = 0;
        for (int blk2 = 0; blk2 < NB * 2; ++blk2)
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    const size_t o = (size_t)blk2 * 1024 + i * 32 + j;
                    const float ai = a[(blk2 >> 1) * 64 + (blk2 & 1) * 32 + i], bj = b[(blk2 >> 1) * 64 + (blk2 & 1) * 32 + j];
                    const float f = fmaf(ai, bj, c[o]);
                    volatile float p = ai * bj;
                    const float u = p + c[o];
                    const bool mf = same(hm[o], f), mu = same(hm[o], u);
                    if (mf && mu) ++both; else if (mf) ++fused; else if (mu) ++unfused; else ++neither;
                }
        printf("32x32x1 C != 0: %zu equal to both, %zu only to fma(a,b,c), %zu only to round(round(a*b)+c), %zu to neither\n", both, fused, unfused, neither);
    }
    {
        hipLaunchKernelGGL(k2, dim3(NB), dim3(64), 0, 0, da, db, dc, dm);
        hipMemcpy(hm.data(), dm, hm.size() * 4, hipMemcpyDeviceToHost);
        size_t cnt[6] = {0, 0, 0, 0, 0, 0};
        for (int blk = 0; blk < NB; ++blk)
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    const size_t o = (size_t)blk * 2048 + i * 32 + j;
                    const float a0 = a[blk * 64 + i], a1 = a[blk * 64 + 32 + i], b0 = b[blk * 64 + j], b1 = b[blk * 64 + 32 + j], cv = c[o];
                    volatile float p0 = a0 * b0, p1 = a1 * b1;
                    const float u01 = (cv + p0) + p1;                       // unfused, k ascending
                    const float f01 = fmaf(a1, b1, fmaf(a0, b0, cv));       // fused chain, k ascending
                    const float f10 = fmaf(a0, b0, fmaf(a1, b1, cv));
                    const float ex = (float)((double)cv + (double)a0 * b0 + (double)a1 * b1);   // (nearly) exact, rounded once
                    bool any = false;
                    if (same(hm[o], u01)) { ++cnt[0]; any = true; }
                    if (same(hm[o], f01)) { ++cnt[1]; any = true; }
                    if (same(hm[o], f10)) { ++cnt[2]; any = true; }
                    if (same(hm[o], ex)) { ++cnt[3]; any = true; }
                    if (!any) ++cnt[4];
                    ++cnt[5];
                }
        printf("32x32x2: of %zu: %zu == unfused k-ascending, %zu == fma chain k-ascending, %zu == fma chain k-descending, %zu == rounded-once sum, %zu none of these\n",
               cnt[5], cnt[0], cnt[1], cnt[2], cnt[3], cnt[4]);
    }
    return 0;
}
