// Probe: is v_mfma_f32_32x32x1_2b_f32 with C = 0 the correctly rounded fp32 product (== v_mul_f32) for every pair?
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o mfma_product mfma_product.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <cmath>
typedef float f32x32 __attribute__((ext_vector_type(32)));

__global__ void k(const float *a, const float *b, float *dm, float *dv)
{   // one wave: block k = lanes 32k..32k+31; D[i][j] = a[i] * b[j]
    const int lane = threadIdx.x;
    const float av = a[blockIdx.x * 64 + lane], bv = b[blockIdx.x * 64 + lane];
    f32x32 c;
    for (int e = 0; e < 32; ++e) c[e] = 0.0f;
    f32x32 d = __builtin_amdgcn_mfma_f32_32x32x1f32(av, bv, c, 0, 0, 0);
    // layout: reg e (0..15 block 0, 16..31 block 1): row i = (e&3) + 8*((e&15)>>2) + 4*(lane>>5), col j = lane&31
    for (int e = 0; e < 32; ++e) {
        const int blk = e >> 4, i = (e & 3) + 8 * ((e & 15) >> 2) + 4 * (lane >> 5), j = lane & 31;
        const float ai = a[blockIdx.x * 64 + blk * 32 + i], bj = b[blockIdx.x * 64 + blk * 32 + j];
        const size_t o = ((size_t)blockIdx.x * 2 + blk) * 1024 + i * 32 + j;
        dm[o] = d[e];
        dv[o] = ai * bj;
    }
}
static uint64_t sm(uint64_t &s) { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
int main()
{
    const int NB = 4096;
    std::vector<float> a(NB * 64), b(NB * 64);
    uint64_t s = 1;
    const float special[] = {0.0f, -0.0f, 1.0f, -1.0f, INFINITY, -INFINITY, NAN, 1e-38f, -1e-38f, 1e-45f, 3e-39f, 3.4e38f, -3.4e38f, 1.17549435e-38f, 0.5f, 2.0f};
    for (size_t i = 0; i < a.size(); ++i) {
        uint32_t ua = (uint32_t)sm(s), ub = (uint32_t)sm(s);
        const int mode = (i / 64) % 4;
        float fa, fb;
        if (mode == 0) { memcpy(&fa, &ua, 4); memcpy(&fb, &ub, 4); }                                   // any bit pattern
        else if (mode == 1) { fa = (float)((ua >> 8) * (1.0 / 16777216.0)) * 0.5f; fb = ((int)(ub >> 8) - 8388608) * (1.0f / 8388608.0f) * 0.2f; }   // HOG x weight range
        else if (mode == 2) { fa = special[ua % 16]; fb = special[ub % 16]; }
        else { fa = ldexpf((float)((ua >> 8) | 1) , -24 - (int)(ua & 63)); fb = ldexpf((float)((ub >> 8) | 1), -24 - (int)(ub & 127)); }   // products that underflow
        a[i] = fa; b[i] = fb;
    }
    float *da, *db, *dm, *dv;
    hipMalloc(&da, a.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dm, (size_t)NB * 2048 * 4); hipMalloc(&dv, (size_t)NB * 2048 * 4);
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(NB), dim3(64), 0, 0, da, db, dm, dv);
    std::vector<float> hm((size_t)NB * 2048), hv((size_t)NB * 2048);
    hipMemcpy(hm.data(), dm, hm.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hv.data(), dv, hv.size() * 4, hipMemcpyDeviceToHost);
    size_t diff[4] = {0, 0, 0, 0}, tot[4] = {0, 0, 0, 0}, nanonly[4] = {0,0,0,0}, zsign[4] = {0,0,0,0}, den[4] = {0,0,0,0};
    int shown = 0;
    for (size_t o = 0; o < hm.size(); ++o) {
        const int mode = (o / 2048) % 4;
        uint32_t um, uv; memcpy(&um, &hm[o], 4); memcpy(&uv, &hv[o], 4);
        ++tot[mode];
        if (um != uv) {
            if (std::isnan(hm[o]) && std::isnan(hv[o])) { ++nanonly[mode]; continue; }
            if ((um | uv) == 0x80000000u) { ++zsign[mode]; continue; }
            if (std::fabs(hv[o]) < 1.17549435e-38f || std::fabs(hm[o]) < 1.17549435e-38f) { ++den[mode]; if (shown < 6) { printf("denormal case: mfma %a vmul %a\n", hm[o], hv[o]); ++shown; } continue; }
            ++diff[mode];
            if (shown < 12) { printf("DIFF mode %d: mfma %a (%08x) vmul %a (%08x)\n", mode, hm[o], um, hv[o], uv); ++shown; }
        }
    }
    for (int m = 0; m < 4; ++m) printf("mode %d: %zu products, %zu real diffs, %zu NaN-payload diffs, %zu zero-sign diffs, %zu denormal-result diffs\n", m, tot[m], diff[m], nanonly[m], zsign[m], den[m]);
    return 0;
}
