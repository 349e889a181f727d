// Probe: does v_pk_add_f32 / v_pk_mul_f32 lose cycles when both 64-bit sources start in the same VGPR bank pair?
// hipcc --offload-arch=gfx950 -O2 -o vgpr_banks vgpr_banks.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X X X X X X X X X X X X X X X X
// accumulators v[32:63] (16 pairs); sources: SAME = pairs whose start has the same (index mod 4) as the accumulator,
// DIFF = the other bank pair
template <int MODE>
__global__ void k(float *out, int iters)
{
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {        // src1 in the same bank pair as src0/dst: v[32:33] += v[64:65] ...
            asm volatile(REP16(
                "v_pk_add_f32 v[32:33], v[32:33], v[64:65]\n v_pk_add_f32 v[34:35], v[34:35], v[66:67]\n"
                "v_pk_add_f32 v[36:37], v[36:37], v[68:69]\n v_pk_add_f32 v[38:39], v[38:39], v[70:71]\n"
                "v_pk_add_f32 v[40:41], v[40:41], v[72:73]\n v_pk_add_f32 v[42:43], v[42:43], v[74:75]\n"
                "v_pk_add_f32 v[44:45], v[44:45], v[76:77]\n v_pk_add_f32 v[46:47], v[46:47], v[78:79]\n")
                ::: "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47",
                    "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79");
        } else if (MODE == 1) { // src1 in the other bank pair: v[32:33] += v[66:67] ...
            asm volatile(REP16(
                "v_pk_add_f32 v[32:33], v[32:33], v[66:67]\n v_pk_add_f32 v[34:35], v[34:35], v[64:65]\n"
                "v_pk_add_f32 v[36:37], v[36:37], v[70:71]\n v_pk_add_f32 v[38:39], v[38:39], v[68:69]\n"
                "v_pk_add_f32 v[40:41], v[40:41], v[74:75]\n v_pk_add_f32 v[42:43], v[42:43], v[72:73]\n"
                "v_pk_add_f32 v[44:45], v[44:45], v[78:79]\n v_pk_add_f32 v[46:47], v[46:47], v[76:77]\n")
                ::: "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47",
                    "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79");
        } else if (MODE == 2) { // mul with broadcast (op_sel_hi) as in the kernel, same bank pair
            asm volatile(REP16(
                "v_pk_mul_f32 v[32:33], v[48:49], v[64:65] op_sel_hi:[0,1]\n v_pk_mul_f32 v[34:35], v[50:51], v[66:67] op_sel_hi:[0,1]\n"
                "v_pk_mul_f32 v[36:37], v[52:53], v[68:69] op_sel_hi:[0,1]\n v_pk_mul_f32 v[38:39], v[54:55], v[70:71] op_sel_hi:[0,1]\n"
                "v_pk_mul_f32 v[40:41], v[56:57], v[72:73] op_sel_hi:[0,1]\n v_pk_mul_f32 v[42:43], v[58:59], v[74:75] op_sel_hi:[0,1]\n"
                "v_pk_mul_f32 v[44:45], v[60:61], v[76:77] op_sel_hi:[0,1]\n v_pk_mul_f32 v[46:47], v[62:63], v[78:79] op_sel_hi:[0,1]\n")
                ::: "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47");
        } else if (MODE == 4) { // mul with an SGPR pair as the weight operand (k_conv3), VGPR source broadcast
            asm volatile(REP16(
                "v_pk_mul_f32 v[32:33], v[48:49], s[8:9] op_sel_hi:[0,1]\n v_pk_mul_f32 v[34:35], v[50:51], s[10:11] op_sel_hi:[0,1]\n"
                "v_pk_mul_f32 v[36:37], v[52:53], s[12:13] op_sel_hi:[0,1]\n v_pk_mul_f32 v[38:39], v[54:55], s[14:15] op_sel_hi:[0,1]\n"
                "v_pk_mul_f32 v[40:41], v[56:57], s[8:9] op_sel_hi:[0,1]\n v_pk_mul_f32 v[42:43], v[58:59], s[10:11] op_sel_hi:[0,1]\n"
                "v_pk_mul_f32 v[44:45], v[60:61], s[12:13] op_sel_hi:[0,1]\n v_pk_mul_f32 v[46:47], v[62:63], s[14:15] op_sel_hi:[0,1]\n")
                ::: "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47");
        } else if (MODE == 5) { // the k_conv3 pattern: mul by an SGPR pair, then add into an accumulator
            asm volatile(REP16(
                "v_pk_mul_f32 v[32:33], v[48:49], s[8:9] op_sel_hi:[0,1]\n v_pk_mul_f32 v[34:35], v[48:49], s[10:11] op_sel_hi:[0,1]\n"
                "v_pk_mul_f32 v[36:37], v[48:49], s[12:13] op_sel_hi:[0,1]\n v_pk_mul_f32 v[38:39], v[48:49], s[14:15] op_sel_hi:[0,1]\n"
                "v_pk_add_f32 v[64:65], v[64:65], v[32:33]\n v_pk_add_f32 v[66:67], v[66:67], v[34:35]\n"
                "v_pk_add_f32 v[68:69], v[68:69], v[36:37]\n v_pk_add_f32 v[70:71], v[70:71], v[38:39]\n")
                ::: "v32","v33","v34","v35","v36","v37","v38","v39","v64","v65","v66","v67","v68","v69","v70","v71");
        } else if (MODE == 6) { // the k_conv pattern: mul by a VGPR pair, then add
            asm volatile(REP16(
                "v_pk_mul_f32 v[32:33], v[48:49], v[72:73] op_sel_hi:[0,1]\n v_pk_mul_f32 v[34:35], v[48:49], v[74:75] op_sel_hi:[0,1]\n"
                "v_pk_mul_f32 v[36:37], v[48:49], v[76:77] op_sel_hi:[0,1]\n v_pk_mul_f32 v[38:39], v[48:49], v[78:79] op_sel_hi:[0,1]\n"
                "v_pk_add_f32 v[64:65], v[64:65], v[32:33]\n v_pk_add_f32 v[66:67], v[66:67], v[34:35]\n"
                "v_pk_add_f32 v[68:69], v[68:69], v[36:37]\n v_pk_add_f32 v[70:71], v[70:71], v[38:39]\n")
                ::: "v32","v33","v34","v35","v36","v37","v38","v39","v64","v65","v66","v67","v68","v69","v70","v71");
        } else {                // mul, other bank pair
            asm volatile(REP16(
                "v_pk_mul_f32 v[32:33], v[50:51], v[64:65] op_sel_hi:[0,1]\n v_pk_mul_f32 v[34:35], v[48:49], v[66:67] op_sel_hi:[0,1]\n"
                "v_pk_mul_f32 v[36:37], v[54:55], v[68:69] op_sel_hi:[0,1]\n v_pk_mul_f32 v[38:39], v[52:53], v[70:71] op_sel_hi:[0,1]\n"
                "v_pk_mul_f32 v[40:41], v[58:59], v[72:73] op_sel_hi:[0,1]\n v_pk_mul_f32 v[42:43], v[56:57], v[74:75] op_sel_hi:[0,1]\n"
                "v_pk_mul_f32 v[44:45], v[62:63], v[76:77] op_sel_hi:[0,1]\n v_pk_mul_f32 v[46:47], v[60:61], v[78:79] op_sel_hi:[0,1]\n")
                ::: "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47");
        }
    }
    if (iters < 0) out[threadIdx.x] = acc;
}

template <int MODE> double run(int waves_per_simd)
{
    float *d; hipMalloc(&d, 4096);
    const int iters = 20000;
    dim3 grid(256 * waves_per_simd), block(256);             // 4 waves per block (one per SIMD) -> waves_per_simd per SIMD on 256 CUs
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr = (double)iters * 128;                 // pk instructions per wave
    const double cyc = ms * 1e-3 * 2.34e9 / (instr * waves_per_simd);   // cycles per instruction per SIMD (at 2.34 GHz)
    hipFree(d);
    return cyc;
}
int main()
{
    for (int w = 1; w <= 6; ++w)
        printf("waves/SIMD %d: add same-bank %.2f  add other-bank %.2f  mul same-bank %.2f  mul other-bank %.2f  mul SGPR-pair %.2f  mul(SGPR)+add %.2f  mul(VGPR)+add %.2f  cycles per instruction\n",
               w, run<0>(w), run<1>(w), run<2>(w), run<3>(w), run<4>(w), run<5>(w), run<6>(w));
    return 0;
}
