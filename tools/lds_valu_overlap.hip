// lds_valu_overlap.hip -- microbenchmark: can the LDS stores of one wave overlap the packed-VALU stream of the
// other wave of the same SIMD on gfx950?  (K1's row time is its VALU issue time PLUS its LDS store time; this
// decides whether a wave-specialised / anti-phase schedule could turn the sum into a maximum.)
// One 512-thread workgroup per CU (128 KiB of LDS keeps a second one out): waves w and w + 4 share a SIMD.
//   mode 0: all 8 waves VALU            mode 1: all 8 waves LDS stores
//   mode 2: waves 0-3 VALU, 4-7 idle    mode 3: waves 4-7 LDS stores, 0-3 idle
//   mode 4: waves 0-3 VALU while waves 4-7 store
//   mode 5: every wave alternates VALU / store phases, both waves of a SIMD in the SAME phase (K1 today)
//   mode 6: the same phases, the two waves of a SIMD in OPPOSITE phases
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_valu_overlap lds_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP8(X) X X X X X X X X

__device__ __forceinline__ void valu_burst(v2f &p0, v2f &p1, v2f &p2, v2f &p3, v2f pb)
{
    // 32 independent-enough packed adds (4 chains x 8)
    REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                      : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));)
}
__device__ __forceinline__ void lds_burst(unsigned addr, v2f v)
{
    // 8 ds_write_b64, 64 lanes x 8 B each, conflict-free, distinct offsets
    asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:4096\n ds_write_b64 %0, %1 offset:8192\n ds_write_b64 %0, %1 offset:12288\n"
                 "ds_write_b64 %0, %1 offset:16384\n ds_write_b64 %0, %1 offset:20480\n ds_write_b64 %0, %1 offset:24576\n ds_write_b64 %0, %1 offset:28672\n"
                 : : "v"(addr), "v"(v) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters)
{
    extern __shared__ unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    const unsigned addr = (unsigned)(size_t)(lds_u8 *)smem + (unsigned)((tid & 63) * 8 + (wave & 3) * 512);   // 512 B per wave-store, 8 planes of 4 KiB
    v2f p0 = {tid * 1e-3f, 1.f}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, pb = {0.5f, 0.25f};
    const bool hi = wave >= 4;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { valu_burst(p0, p1, p2, p3, pb); valu_burst(p0, p1, p2, p3, pb); }
        else if (MODE == 1) { lds_burst(addr, p0); lds_burst(addr, p1); }
        else if (MODE == 2) { if (!hi) { valu_burst(p0, p1, p2, p3, pb); valu_burst(p0, p1, p2, p3, pb); } }
        else if (MODE == 3) { if (hi) { lds_burst(addr, p0); lds_burst(addr, p1); } }
        else if (MODE == 4) { if (!hi) { valu_burst(p0, p1, p2, p3, pb); valu_burst(p0, p1, p2, p3, pb); } else { lds_burst(addr, p0); lds_burst(addr, p1); } }
        else if (MODE == 5) {   // in phase: 8 VALU bursts then 4 store bursts, a barrier per phase like K1's passes
            for (int j = 0; j < 8; ++j) valu_burst(p0, p1, p2, p3, pb);
            for (int j = 0; j < 4; ++j) lds_burst(addr, p0);
            __syncthreads();
        } else {                // opposite phases: waves 4-7 store while waves 0-3 compute, then swap
            if (!hi) { for (int j = 0; j < 8; ++j) valu_burst(p0, p1, p2, p3, pb); } else { for (int j = 0; j < 4; ++j) lds_burst(addr, p0); }
            if (hi) { for (int j = 0; j < 8; ++j) valu_burst(p0, p1, p2, p3, pb); } else { for (int j = 0; j < 4; ++j) lds_burst(addr, p0); }
            __syncthreads();
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const float r = p0.x + p1.y + p2.x + p3.y;
    if (r == 123.456f) out[0] = r;
}

template <int MODE> float run(const char *name, int iters)
{
    float *d; hipMalloc(&d, 4);
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256, 512, 128 * 1024>>>(d, 10); hipDeviceSynchronize();
    hipEventRecord(a); k<MODE><<<256, 512, 128 * 1024>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("mode %d %-44s %.3f ms  (%.1f ns per iteration)\n", MODE, name, ms, ms * 1e6 / iters);
    hipFree(d);
    return ms;
}
int main()
{
    const int it = 20000;
    run<0>("8 waves: 64 v_pk_add each", it);
    run<1>("8 waves: 16 ds_write_b64 each", it);
    run<2>("waves 0-3: 64 v_pk_add, others idle", it);
    run<3>("waves 4-7: 16 ds_write_b64, others idle", it);
    run<4>("waves 0-3 VALU beside waves 4-7 stores", it);
    run<5>("in phase: 256 v_pk_add + 32 stores, barrier", it / 4);
    run<6>("opposite phases: the same work per wave", it / 4);
    return 0;
}
