// xcorr14.hpp -- geometry of the B = 16384 (L = 8192) specialisation of K0 / K1 (the configuration every BASELINE.json config
// except cfg5 runs at): sizes, the LDS image's addressing, wave-level ordering.  The kernels themselves are xcorr14p.hpp (packed
// complex arithmetic) and xcorr14q.hpp (two rows per CU); the first, scalar-fp32 formulation of the same network lives in
// tools/xcorr14_scalar.hpp for reference and is not part of the product library (it agrees to rounding, not to the bit, since
// the packed kernels fold the inverse passes' twiddles into their first radix-2 stage).
//
// One 512-thread workgroup per row, 32 points per thread in registers, the row resident in LDS
// (128 KiB + 4 KiB padding of the CU's 160 KiB):
//
//   16384 = 32 x 32 x 16      P0 radix-32 (stride 512)  P1 radix-32 (stride 16)  J radix-16 (stride 1)
//
//   K1:  int8 HBM --P0--> LDS --P1--> LDS --J: DFT16 . conj(ref) . IDFT16--> LDS --P1'--> LDS --P0'--> |.|^2, argmax
//
// i.e. 4 LDS exchanges for the forward + inverse transform pair (a Stockham radix-16 chain needs 7).
//   * P0 reads the int8 row straight from HBM (2-byte lane loads, 128 B per wave-instruction) and
//     prunes the zero half of the reference's zero-padding (src/crtlsdr.cc:205-207,215-218):
//     the first radix-2 stage of the radix-32 butterfly has one zero input.
//   * DIF forward / DIT inverse: the spectrum is never reordered; the reference spectrum K0 leaves
//     in HBM is in the same digit-reversed order, laid out [slot][group] so the J pass reads it
//     with coalesced 16-byte loads.
//   * inter-pass twiddles: five table values w^1,w^2,w^4,w^8,w^16 per thread and pass (tables are
//     laid out so the loads coalesce) and 26 complex products; product depth <= 4, so every
//     twiddle is within ~3 ulp of the correctly rounded value.
//   * LDS layout: one 16-element (128 B) group pad after every 512 elements plus an XOR swizzle
//     keyed on the group index G = e >> 4: 16-byte slot s of group G lives at slot s ^ (G & 7) of
//     group G ^ ((G >> 3) & 1) -- conflict-free for the stride-512 (b64), stride-16 (b64) and
//     stride-1 (b128 reads in 16-lane groups, b128 writes in 8-lane groups) access patterns.
//   * only P0 -> P1 and P1' -> P0' are workgroup-wide exchanges (2 barriers per row): P1, J and
//     P1' of a 512-element sub-block are produced and consumed by the same wave (a wave's 64
//     threads own 4 sub-blocks = 128 J groups), so that section needs only wave-level ordering
//     and the 8 waves drift apart, overlapping one wave's LDS traffic with another's VALU work.
#pragma once
#include "arith.hpp"
#include "fft_lds.hpp"
#include "plan_args.hpp"
#include <stdint.h>

// Diagnostic builds (tools/k1_phases.hip) define CRSDR_STAMP(i) to record s_memtime at phase
// boundaries; in the product build it expands to nothing and no stamp executes.
#ifndef CRSDR_STAMP
#define CRSDR_STAMP(i)
#endif

namespace crsdr {
namespace x14 {

constexpr int N = 16384, L = 8192, THREADS = 512;
constexpr int LDS_ELEMS = (1024 + 32) * 16; // 1024 groups of 16 + one pad group per 32
constexpr int LDS_BYTES = LDS_ELEMS * 8 + 512;
constexpr int TWA_STRIDE = 512, TWB_STRIDE = 16; // [5][512] W_16384^(t 2^j), [5][16] W_512^(n 2^j)
// |ifft|^2 of unscaled int8 data is (127^2)^2 times the reference's value (signal and ref spectra
// each carry one factor 127): mag = sqrt(peak / L) / 127^2
constexpr float kInvScale2 = 1.0f / (127.0f * 127.0f);

// cos / sin (pi k / 16), k = 0..16
__device__ constexpr float kCos16[17] = {1.0f,
                                         0.98078528040323044913f,
                                         0.92387953251128675613f,
                                         0.83146961230254523708f,
                                         0.70710678118654752440f,
                                         0.55557023301960222474f,
                                         0.38268343236508977173f,
                                         0.19509032201612826785f,
                                         0.0f,
                                         -0.19509032201612826785f,
                                         -0.38268343236508977173f,
                                         -0.55557023301960222474f,
                                         -0.70710678118654752440f,
                                         -0.83146961230254523708f,
                                         -0.92387953251128675613f,
                                         -0.98078528040323044913f,
                                         -1.0f};

// output X[k] of a 32-point transform is left in register xpos(k) (two 16-point halves, even k first)
__device__ __forceinline__ constexpr int xpos(int k) { return (k & 1) * 16 + (k >> 1); }

// ---- LDS addressing (float2 element index) ------------------------------------------------------
// generic: element e -> G = e>>4, s = e&15; group' = G ^ ((G>>3)&1); slot' = (s>>1) ^ (G&7);
//          phys = (group' + (group'>>5))*16 + 2*slot' + (s&1)
// P0 / P0': element k*512 + t  ->  k*528 + p0_base(t)      (G = 32k + (t>>4): G&7 = (t>>4)&7, (G>>3)&1 = (t>>7)&1)
__device__ __forceinline__ int p0_base(int t)
{
    return ((((t >> 4) ^ ((t >> 7) & 1))) << 4) + ((t & 15) ^ (((t >> 4) & 7) << 1));
}
// P1 / P1': element blk*512 + i*16 + n2  ->  blk*528 + p1_off(i) + (n2 ^ p1_swz(i))   (G = 32 blk + i)
__device__ __forceinline__ constexpr int p1_off(int i) { return (i ^ ((i >> 3) & 1)) << 4; }
__device__ __forceinline__ constexpr int p1_swz(int i) { return (i & 7) << 1; }
// J: group g (16 elements): 16-byte slot j lives at float4 index j_base(g) + (j ^ (g & 7))
__device__ __forceinline__ int j_base(int g)
{
    const int gp = g ^ ((g >> 3) & 1);
    return (gp + (gp >> 5)) * 8;
}
// wave-level ordering of LDS traffic inside the wave-local section: LDS operations of one wave
// execute in issue order, so waiting for this wave's own outstanding LDS ops is sufficient.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

} // namespace x14
} // namespace crsdr
