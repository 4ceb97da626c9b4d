// cvt_round.hip -- how v_cvt_pk_u8_f32 rounds and saturates on gfx950 (phase kernel):
//   hipcc -O2 --offload-arch=gfx950 -o cvt_round tools/cvt_round.hip && ./cvt_round
// Measured (MI355X): ties go to even (0.5 -> 0, 1.5 -> 2, 2.5 -> 2, 254.5 -> 254), the result saturates to [0, 255]
// (300 -> 255, -1 -> 0).  rotq_word relies on the saturation only: its rounding is done before (rintf as one packed add of
// 1.5 * 2^23), because y + 128 ahead of the conversion would be a second rounding.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float *in, unsigned *out, int n)
{
    const int i = threadIdx.x;
    if (i < n) {
        unsigned o = 0;
        const float x = in[i];
        asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(o) : "v"(x));
        out[i] = o;
    }
}
int main()
{
    const float h[] = {0.5f, 1.5f, 2.5f, 3.5f, 0.49f, 0.51f, 1.49f, 2.51f, 254.5f, 255.5f, 255.49f, 300.f, -0.5f, -1.f, 127.5f, 128.5f, 126.5f, 0.9999f, 1.0f, 2.4999f, 2.5001f};
    const int n = sizeof(h) / sizeof(h[0]);
    float *d = nullptr;
    unsigned *o = nullptr, r[64];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess || hipMalloc(&o, 4 * n) != hipSuccess) return 1;
    if (hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
    if (hipMemcpy(r, o, 4 * n, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int i = 0; i < n; ++i) printf("%g -> %u\n", h[i], r[i] & 255u);
    return 0;
}
