// crsdr.hip -- C ABI (include/crsdr.h) over the gfx950 kernels in kernels.hpp.
// Host side is plain C++: plan = device buffers + twiddle table + two HIP streams + events.
// No torch types, no CPU fallback: every compute entry point needs a HIP device.
#include "../../include/crsdr.h"
#include "kernels.hpp"

#include <hip/hip_runtime.h>

#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <vector>
#include <sched.h>

using namespace crsdr;

// ---- error plumbing ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(e_ == hipErrorOutOfMemory ? CRSDR_ENOMEM                                    \
                        : (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? CRSDR_ENODEV    \
                                                                                  : CRSDR_EHIP,     \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" int crsdr_abi_version(void) { return CRSDR_ABI_VERSION; }
extern "C" const char *crsdr_last_error(void) { return g_err; }

extern "C" int crsdr_device_count(int *count)
{
    if (!count) return fail(CRSDR_EINVAL, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(CRSDR_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return CRSDR_OK;
}

extern "C" int crsdr_device_info(int device, char *name, int name_cap, int *compute_units, int *clock_khz, int *memory_clock_khz,
                                 size_t *memory_bytes)
{
    int n = 0;
    int rc = crsdr_device_count(&n);
    if (rc) return rc;
    if (device < 0 || device >= n) return fail(CRSDR_ENODEV, "device_info: device %d of %d", device, n);
    hipDeviceProp_t pr;
    HIP_TRY(hipGetDeviceProperties(&pr, device));
    if (name && name_cap > 0) { snprintf(name, (size_t)name_cap, "%s (%s)", pr.name, pr.gcnArchName); }
    if (compute_units) *compute_units = pr.multiProcessorCount;
    if (clock_khz) *clock_khz = pr.clockRate;
    if (memory_clock_khz) *memory_clock_khz = pr.memoryClockRate;
    if (memory_bytes) *memory_bytes = pr.totalGlobalMem;
    return CRSDR_OK;
}

static int require_device();
// Page-locked memory should sit on the NUMA node the GPU hangs off: a transfer that crosses the socket interconnect runs at
// roughly half the PCIe rate.  The pages are placed where the allocating thread runs, so the thread is moved to that node's
// CPUs for the duration of the allocation (sysfs: /sys/bus/pci/devices/<bdf>/numa_node, /sys/devices/system/node/nodeN/cpulist).
// Best effort: any failure leaves the affinity alone.  CRSDR_HOST_NUMA=0 switches it off.
static bool near_device_cpus(cpu_set_t *set)
{
    static const bool enabled = [] { const char *e = getenv("CRSDR_HOST_NUMA"); return !e || atoi(e) != 0; }();
    if (!enabled) return false;
    int dev = 0;
    char bdf[64] = {0}, path[256], buf[4096];
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(bdf, sizeof(bdf), dev) != hipSuccess) return false;
    for (char *c = bdf; *c; ++c) *c = (char)tolower(*c);
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    int node = -1;
    const int got = fscanf(f, "%d", &node);
    fclose(f);
    if (got != 1 || node < 0) return false;
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f) return false;
    const bool ok = fgets(buf, sizeof(buf), f) != nullptr;
    fclose(f);
    if (!ok) return false;
    CPU_ZERO(set);
    int n = 0;
    for (char *tok = strtok(buf, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
        int lo = 0, hi = 0;
        const int k = sscanf(tok, "%d-%d", &lo, &hi);
        if (k < 1) continue;
        if (k == 1) hi = lo;
        for (int c = lo; c <= hi && c < CPU_SETSIZE; ++c) { CPU_SET(c, set); ++n; }
    }
    return n > 0;
}

extern "C" int crsdr_host_alloc(void **ptr, size_t bytes)
{
    if (!ptr || !bytes) return fail(CRSDR_EINVAL, "host_alloc: NULL pointer or zero size");
    *ptr = nullptr;
    { int rc_ = require_device(); if (rc_) return rc_; }
    cpu_set_t old_set, near_set;
    const bool moved = sched_getaffinity(0, sizeof(old_set), &old_set) == 0 && near_device_cpus(&near_set) &&
                       sched_setaffinity(0, sizeof(near_set), &near_set) == 0;
    hipError_t e = hipHostMalloc(ptr, bytes, hipHostMallocDefault);
    if (e == hipSuccess && moved) memset(*ptr, 0, bytes);      // first touch on the near node, if the runtime left any page untouched
    if (moved) (void)sched_setaffinity(0, sizeof(old_set), &old_set);
    if (e != hipSuccess) { *ptr = nullptr; return fail(e == hipErrorOutOfMemory ? CRSDR_ENOMEM : CRSDR_EHIP, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
    return CRSDR_OK;
}
extern "C" int crsdr_host_free(void *ptr)
{
    if (!ptr) return CRSDR_OK;
    HIP_TRY(hipHostFree(ptr));
    return CRSDR_OK;
}

static int require_device()
{
    int n = 0;
    int rc = crsdr_device_count(&n);
    if (rc) return rc;
    if (n < 1) return fail(CRSDR_ENODEV, "no HIP device: libcrsdr has no CPU fallback");
    return CRSDR_OK;
}

static int ilog2_exact(int n)
{
    if (n < 1 || (n & (n - 1))) return -1;
    int l = 0;
    while ((1 << l) < n) ++l;
    return l;
}

// forward twiddle table W_n^k = (cos, -sin)(2 pi k / n), generated in double, rounded once
static int make_twiddles(int n, float2 **d_tw)
{
    std::vector<float2> h((size_t)n);
    for (int k = 0; k < n; ++k) {
        double a = 2.0 * M_PI * (double)k / (double)n;
        h[k] = make_float2((float)std::cos(a), (float)(-std::sin(a)));
    }
    HIP_TRY(hipMalloc((void **)d_tw, sizeof(float2) * (size_t)n));
    HIP_TRY(hipMemcpy(*d_tw, h.data(), sizeof(float2) * (size_t)n, hipMemcpyHostToDevice));
    return CRSDR_OK;
}

// x14 tables: twA[j][t] = W_16384^(t 2^j), t < 512; twB[j][n] = W_512^(n 2^j), n < 16; j < 5
static int make_twiddles14(float2 **d_twA, float2 **d_twB)
{
    std::vector<float2> a(5 * 512), b(5 * 16);
    for (int j = 0; j < 5; ++j) {
        for (int t = 0; t < 512; ++t) {
            double ang = 2.0 * M_PI * (double)(t << j) / 16384.0;
            a[j * 512 + t] = make_float2((float)std::cos(ang), (float)(-std::sin(ang)));
        }
        for (int n = 0; n < 16; ++n) {
            double ang = 2.0 * M_PI * (double)(n << j) / 512.0;
            b[j * 16 + n] = make_float2((float)std::cos(ang), (float)(-std::sin(ang)));
        }
    }
    HIP_TRY(hipMalloc((void **)d_twA, sizeof(float2) * a.size()));
    HIP_TRY(hipMalloc((void **)d_twB, sizeof(float2) * b.size()));
    HIP_TRY(hipMemcpy(*d_twA, a.data(), sizeof(float2) * a.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(*d_twB, b.data(), sizeof(float2) * b.size(), hipMemcpyHostToDevice));
    return CRSDR_OK;
}

// forward table W_n^(k * step), k < count
static int make_twiddle_table(double n, size_t count, size_t step, float2 **d)
{
    std::vector<float2> h(count);
    for (size_t k = 0; k < count; ++k) {
        double a = 2.0 * M_PI * (double)(k * step) / n;
        h[k] = make_float2((float)std::cos(a), (float)(-std::sin(a)));
    }
    HIP_TRY(hipMalloc((void **)d, sizeof(float2) * count));
    HIP_TRY(hipMemcpy(*d, h.data(), sizeof(float2) * count, hipMemcpyHostToDevice));
    return CRSDR_OK;
}

constexpr int kMinLog2 = 4, kMaxLog2 = 14, kMaxLog2Plan = 22; // LDS-resident sizes; long-block plans up to 2^22 // LDS-resident transform sizes: 16 .. 16384 points

template <int LOG2N>
static constexpr size_t fft_lds_bytes() { return sizeof(float2) * ((size_t)1 << LOG2N) + 256; }

// ---- kernel launchers, dispatched on log2(B) ------------------------------------------------------
template <int LOG2N>
static hipError_t launch_ref_spectrum(hipStream_t s, int nblocks, const int8_t *rows, size_t block_stride, const float2 *tw,
                                      float2 *refspec, uint32_t xor80)
{
    auto kern = k_ref_spectrum<LOG2N>;
    constexpr size_t lds = fft_lds_bytes<LOG2N>();
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(FftGeom<LOG2N>::THREADS), lds, s, rows, block_stride, tw, refspec, xor80);
    return hipGetLastError();
}

template <int LOG2N>
static hipError_t launch_xcorr_lag(hipStream_t s, const XcorrArgs &a, int row_count, const float2 *tw)
{
    auto kern = k_xcorr_lag<LOG2N>;
    constexpr size_t lds = fft_lds_bytes<LOG2N>();
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(row_count, a.nblocks), dim3(FftGeom<LOG2N>::THREADS), lds, s, a, tw);
    return hipGetLastError();
}

// CRSDR_K1_VARIANT (README, "switches"): "auto" (default) = the two-row kernel (q, xcorr14q.hpp) for launches with enough rows per
// CU, the packed one-row kernel (p, xcorr14p.hpp) otherwise; "packed" / "q" force one of them.  Same arithmetic, identical bits.
// (Measured-slower experiments live in tools/: xcorr14h.hpp, half-row LDS images -- dead end (8) of DESIGN.md --, xcorr14w.hpp,
// 1024 threads per row -- dead end (11) --, and xcorr14_scalar.hpp, the scalar-fp32 twin of r01.)
static char k1_variant()
{
    static const char v = [] { const char *e = getenv("CRSDR_K1_VARIANT"); const char c = e ? e[0] : 'a'; return (c == 'p' || c == 'q') ? c : 'a'; }();
    return v;
}

static hipError_t launch_ref_spectrum14(hipStream_t s, int nblocks, const int8_t *rows, size_t block_stride, const float2 *twA,
                                        const float2 *twB, float2 *refspec, uint32_t xor80)
{
    auto kern = x14p::k_ref_spectrum14p;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, x14::LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(x14::THREADS), x14::LDS_BYTES, s, rows, block_stride, reinterpret_cast<const c2 *>(twA),
                       reinterpret_cast<const c2 *>(twB), (float4 *)refspec, xor80);
    return hipGetLastError();
}

static int device_cus()
{
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    return cus;
}
// which B = 16384 cross-correlation kernel a launch of rows_all rows takes: 'q' (two rows per CU) or 'p' (one)
static char k1_pick(int rows_all, bool allow_q)
{
    char variant = k1_variant();
    // measured (r01, T = 64): q is 3-7 % faster at 1024 / 256 / 128 / 96 / 48 rows per block (256 ... 12 rows per CU and launch),
    // 1.4 % at 64, level with p at 32 and 21 rows (8 and 5 per CU; r03: level at 10 as well)
    if (variant == 'a') variant = rows_all >= 12 * device_cus() ? 'q' : 'p';
    if (variant == 'q' && !allow_q) variant = 'p';   // a plan whose two-row kernel once ran out of a bounded wait stays on the packed kernel
    return variant;
}
static hipError_t launch_xcorr_lag14(hipStream_t s, const XcorrArgs &a, int row_count, const float2 *twA, const float2 *twB, int *waitflag, bool *used_q,
                                     unsigned int *work, unsigned int *work_base, bool allow_q, const std::function<hipError_t()> &before_bounded)
{
    const int cus = device_cus();
    const int rows_all = row_count * a.nblocks, items = rows_all + (a.fold ? a.nblocks : 0);
    const char variant = a.fold ? 'p' : k1_pick(rows_all, allow_q);      // (the caller folds the reference spectra in only where the packed kernel runs)
    if (used_q) *used_q = variant == 'q';
    // a launch with bounded waits (the two-row kernel's, a folded launch's wait for its reference spectra) snapshots the carried state first
    if ((variant == 'q' || a.fold) && before_bounded) { hipError_t eb = before_bounded(); if (eb != hipSuccess) return eb; }
    if (variant != 'q') {
        auto kp = x14p::k_xcorr_lag14p;
        hipError_t ep = hipFuncSetAttribute((const void *)kp, hipFuncAttributeMaxDynamicSharedMemorySize, x14::LDS_BYTES);
        if (ep != hipSuccess) return ep;
        hipLaunchKernelGGL(kp, dim3((unsigned)items), dim3(x14::THREADS), x14::LDS_BYTES, s, a, twA, twB, row_count);
        return hipGetLastError();
    }
    {   // two rows per CU in opposite phases (xcorr14q.hpp), one persistent workgroup per CU
        auto kq = x14p::k_xcorr_lag14q;
        hipError_t eq = hipFuncSetAttribute((const void *)kq, hipFuncAttributeMaxDynamicSharedMemorySize, x14p::LDSQ_BYTES);
        if (eq != hipSuccess) return eq;
        // polls per wait; tests force 0 ("0@3": from the process's fourth two-row launch on, so that clean launches come first)
        static const int qspin_env = [] { const char *e = getenv("CRSDR_K1_QSPIN"); return e ? atoi(e) : x14p::kQSpinLimit; }();
        static const long qspin_from = [] { const char *e = getenv("CRSDR_K1_QSPIN"); const char *at = e ? strchr(e, '@') : nullptr; return at ? atol(at + 1) : 0L; }();
        static long qlaunches = 0;
        const int qspin = qlaunches++ >= qspin_from ? qspin_env : x14p::kQSpinLimit;
        hipLaunchKernelGGL(kq, dim3((unsigned)std::max(1, std::min(cus, (items + 1) / 2))), dim3(2 * x14p::QG), x14p::LDSQ_BYTES, s, a, twA, twB, row_count,
                           waitflag, work, *work_base, qspin);
        const hipError_t el = hipGetLastError();
        if (el == hipSuccess) *work_base += (unsigned)items;      // a launch that ran advances the device counter by exactly its item count (xcorr14q.hpp)
        return el;
    }
}

template <int LOG2N, int DIR>
static hipError_t launch_op_fft(hipStream_t s, int howmany, float2 *out, const float2 *in, const float2 *tw)
{
    auto kern = k_op_fft<LOG2N, DIR>;
    constexpr size_t lds = fft_lds_bytes<LOG2N>();
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(howmany), dim3(FftGeom<LOG2N>::THREADS), lds, s, out, in, tw);
    return hipGetLastError();
}

// ---- long-block path (B = N1 x 16384) ----------------------------------------------------------------
#define CRSDR_DISPATCH_N1(l2, CALL)                                                     \
    [&]() -> hipError_t {                                                               \
        switch (l2) {                                                                   \
        case 1: { constexpr int LG = 1; return CALL; }                                  \
        case 2: { constexpr int LG = 2; return CALL; }                                  \
        case 3: { constexpr int LG = 3; return CALL; }                                  \
        case 4: { constexpr int LG = 4; return CALL; }                                  \
        case 5: { constexpr int LG = 5; return CALL; }                                  \
        case 6: { constexpr int LG = 6; return CALL; }                                  \
        case 7: { constexpr int LG = 7; return CALL; }                                  \
        case 8: { constexpr int LG = 8; return CALL; }                                  \
        default: return hipErrorInvalidValue;                                           \
        }                                                                               \
    }()

// column passes of the long-block path: persistent workgroups, two per CU (64 KiB tiles), each walking its share of the
// (row, tile) work items with the next item's loads in flight (longblock.hpp); CRSDR_LONG_PERSIST=0: one workgroup per item
static int lb_grid(int nwork, bool stage_c = false)
{
    // stage A: persistent, two workgroups per CU (what the 64 KiB tiles allow); stage C (reads only) is better off with one
    // workgroup per item: measured r02, cfg5, per launch of 10.5 rows: 55 us against 61 us persistent (its reads alone run at
    // 4.1 TB/s either way: 43 us); stage A (mostly stores) gains: 61 -> 52 us
    const int per_cu = stage_c ? 0 : 2;
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
#ifdef CRSDR_LB_EXPERIMENT
    static const int dbg_set = [] { const char *e = getenv("CRSDR_LB_DBG"); int v = e ? atoi(e) : 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(lb::lb_dbg), &v, sizeof(v)); return v; }();
    (void)dbg_set;
#endif
    return per_cu > 0 ? std::max(1, std::min(nwork, per_cu * cus)) : nwork;
}
template <int LOG2N1, bool IS_REF>
static hipError_t launch_long_fwd_cols(hipStream_t s, int nrows_launch, const int8_t *rows, int row_begin, uint32_t xor80,
                                       const lb::LongTw &tw, float2 *Y)
{
    auto kern = lb::k_long_fwd_cols<LOG2N1, IS_REF>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb::LDS_BYTES);
    if (e != hipSuccess) return e;
    const int nwork = lb::ntiles(LOG2N1) * nrows_launch;
    hipLaunchKernelGGL(kern, dim3((unsigned)lb_grid(nwork)), dim3(lb::THREADS), lb::LDS_BYTES, s, rows, row_begin, xor80, tw, Y, nwork);
    return hipGetLastError();
}
// stage B with two lines per CU in opposite phases (k_rows14_cf32q): a line's HBM phases run beside the other line's transforms
static hipError_t launch_long_rows_q(hipStream_t s, int n1, int nrows_launch, float2 *Y, const float2 *twA, const float2 *twB, float2 *refspec,
                                     int *waitflag, unsigned int *work, unsigned int *work_base, const x14p::RampArgs *ramp = nullptr, float2 *Yout = nullptr)
{
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    auto kq = ramp ? x14p::k_rows14_cf32q<true> : x14p::k_rows14_cf32q<false>;
    hipError_t e = hipFuncSetAttribute((const void *)kq, hipFuncAttributeMaxDynamicSharedMemorySize, x14p::LDSQ_BYTES);
    if (e != hipSuccess) return e;
    const int items = n1 * nrows_launch;
    static const int qspin = [] { const char *ev = getenv("CRSDR_K1_QSPIN"); return ev ? atoi(ev) : x14p::kQSpinLimit; }();
    // (r03: dealing the lines to just as many workgroups as make every round full -- cfg5: 2688 lines = 6 rounds x 448 groups on 224
    // CUs instead of 5 full rounds and a sixth with 128 lines -- measured level: 2 473 / 2 544 against 2 500 / 2 549 blocks/s)
    const int grid = std::max(1, std::min(cus, (items + 1) / 2));
    // one queue per XCD (k_rows14_cf32q); static order: the work counter is not used.  (The apply pass's per-row spectra are
    // 2.7 MB for cfg5 and stay in every L2: lines in memory order there.)
    const int nq = (!ramp && n1 % 8 == 0 && grid % 8 == 0) ? 8 : 1;
    (void)work; (void)work_base;
    hipLaunchKernelGGL(kq, dim3((unsigned)grid), dim3(2 * x14p::QG), x14p::LDSQ_BYTES, s, reinterpret_cast<c2 *>(Y),
                       reinterpret_cast<const c2 *>(twA), reinterpret_cast<const c2 *>(twB), (const float4 *)refspec, n1, nrows_launch, nq, waitflag, qspin,
                       ramp ? *ramp : x14p::RampArgs{}, reinterpret_cast<c2 *>(Yout));
    return hipGetLastError();
}
// stage B: the 16384-point row transforms run on the 32x32x16 structure of xcorr14.hpp
template <bool IS_REF>
static hipError_t launch_long_rows(hipStream_t s, int n1, int nrows_launch, float2 *Y, const float2 *twA, const float2 *twB, float2 *refspec)
{
    auto kern = x14p::k_rows14_cf32p<IS_REF, false>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, x14::LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(n1, nrows_launch), dim3(x14::THREADS), x14::LDS_BYTES, s, reinterpret_cast<c2 *>(Y), reinterpret_cast<const c2 *>(twA),
                       reinterpret_cast<const c2 *>(twB), (float4 *)refspec, x14p::RampArgs{});
    return hipGetLastError();
}
// apply pass of crsdr_plan_set_frac_apply: row transforms with the row's fractional-delay response, then column transforms to int8
static hipError_t launch_long_rows_ramp(hipStream_t s, int n1, int nrows_launch, float2 *Y, const float2 *twA, const float2 *twB, const x14p::RampArgs &ra)
{
    auto kern = x14p::k_rows14_cf32p<false, true>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, x14::LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(n1, nrows_launch), dim3(x14::THREADS), x14::LDS_BYTES, s, reinterpret_cast<c2 *>(Y), reinterpret_cast<const c2 *>(twA),
                       reinterpret_cast<const c2 *>(twB), (float4 *)nullptr, ra);
    return hipGetLastError();
}
static hipError_t launch_long_rows_ref1(hipStream_t s, float2 *Y, const float2 *twA, const float2 *twB, float2 *refspec)
{
    return launch_long_rows<true>(s, 1, 1, Y, twA, twB, refspec);
}
template <int LOG2N1>
static hipError_t launch_long_out_cols(hipStream_t s, int nrows_launch, const float2 *Z, const lb::LongTw &tw, int8_t *out)
{
    auto kern = lb::k_long_inv_cols<LOG2N1, true>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb::LDS_BYTES);
    if (e != hipSuccess) return e;
    const int nwork = lb::ntiles(LOG2N1) * nrows_launch;
    hipLaunchKernelGGL(kern, dim3((unsigned)lb_grid(nwork)), dim3(lb::THREADS), lb::LDS_BYTES, s, Z, tw, (lb::LongPartial *)nullptr, out, nwork);
    return hipGetLastError();
}
template <int LOG2N1>
static hipError_t launch_long_inv_cols(hipStream_t s, int nrows_launch, const float2 *Z, const lb::LongTw &tw, lb::LongPartial *part)
{
    auto kern = lb::k_long_inv_cols<LOG2N1>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb::LDS_BYTES);
    if (e != hipSuccess) return e;
    const int nwork = lb::ntiles(LOG2N1) * nrows_launch;
    hipLaunchKernelGGL(kern, dim3((unsigned)lb_grid(nwork, true)), dim3(lb::THREADS), lb::LDS_BYTES, s, Z, tw, part, (int8_t *)nullptr, nwork);
    return hipGetLastError();
}

template <int LOG2N>
static hipError_t launch_frac_apply(hipStream_t s, int row_count, int nblocks, const FracArgs &fa, const float2 *tw, const float2 *twA, const float2 *twB,
                                    const uint32_t *k2tab)
{
    if constexpr (LOG2N == 14) {
        if (k2tab && twA && twB) {             // the pass on K1's 32 x 32 x 16 network (xcorr14p.hpp)
            x14p::FracRowArgs ra{fa.rows, fa.block_stride, fa.packet, fa.packet_stride, fa.slab, fa.slab_stride, fa.nrows, fa.row_begin, fa.xor80, fa.lag, fa.frac,
                                 fa.frac_override, fa.gain, fa.phasor, k2tab, tw};
            constexpr int lds14 = x14::LDS_BYTES + 2048;          // the image + the row's two response tables
            hipError_t e = hipFuncSetAttribute((const void *)x14p::k_frac_apply14, hipFuncAttributeMaxDynamicSharedMemorySize, lds14);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(x14p::k_frac_apply14, dim3(row_count, nblocks), dim3(x14::THREADS), lds14, s, ra, twA, twB);
            return hipGetLastError();
        }
    }
    auto kern = k_frac_apply<LOG2N>;
    constexpr size_t lds = sizeof(float2) * ((size_t)1 << LOG2N) + 256;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(row_count, nblocks), dim3(FftGeom<LOG2N>::THREADS), lds, s, fa, tw);
    return hipGetLastError();
}

#define CRSDR_DISPATCH_LOG2(l2, CALL)                                                   \
    [&]() -> hipError_t {                                                               \
        switch (l2) {                                                                   \
        case 4: { constexpr int LG = 4; return CALL; }                                  \
        case 5: { constexpr int LG = 5; return CALL; }                                  \
        case 6: { constexpr int LG = 6; return CALL; }                                  \
        case 7: { constexpr int LG = 7; return CALL; }                                  \
        case 8: { constexpr int LG = 8; return CALL; }                                  \
        case 9: { constexpr int LG = 9; return CALL; }                                  \
        case 10: { constexpr int LG = 10; return CALL; }                                \
        case 11: { constexpr int LG = 11; return CALL; }                                \
        case 12: { constexpr int LG = 12; return CALL; }                                \
        case 13: { constexpr int LG = 13; return CALL; }                                \
        case 14: { constexpr int LG = 14; return CALL; }                                \
        default: return hipErrorInvalidValue;                                           \
        }                                                                               \
    }()

// ================================================================================================
// (i) per-op entry points (class cdsp).  Host pointers; staged through grow-only device buffers.
// ================================================================================================
namespace {
struct OpCtx {
    std::mutex mu;
    void *buf[4] = {nullptr, nullptr, nullptr, nullptr};      // [3]: the tiled covariance's partial sums
    size_t cap[4] = {0, 0, 0, 0};
    float2 *tw[kMaxLog2 + 1] = {};
    int tw_dev[kMaxLog2 + 1] = {};
    int dev = -1;
};
OpCtx g_op;

int op_reserve(int slot, size_t bytes)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (g_op.dev != dev) {
        for (int i = 0; i < 4; ++i) { if (g_op.buf[i]) (void)hipFree(g_op.buf[i]); g_op.buf[i] = nullptr; g_op.cap[i] = 0; }
        for (int i = 0; i <= kMaxLog2; ++i) { if (g_op.tw[i]) (void)hipFree(g_op.tw[i]); g_op.tw[i] = nullptr; }
        g_op.dev = dev;
    }
    if (g_op.cap[slot] >= bytes) return CRSDR_OK;
    if (g_op.buf[slot]) (void)hipFree(g_op.buf[slot]);
    g_op.buf[slot] = nullptr; g_op.cap[slot] = 0;
    size_t want = bytes < 4096 ? 4096 : bytes;
    HIP_TRY(hipMalloc(&g_op.buf[slot], want));
    g_op.cap[slot] = want;
    return CRSDR_OK;
}

inline dim3 grid1d(size_t n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }
} // namespace

#define OP_PROLOGUE(cond, msg)                                \
    if (!(cond)) return fail(CRSDR_EINVAL, msg);              \
    { int rc_ = require_device(); if (rc_) return rc_; }      \
    std::lock_guard<std::mutex> lock_(g_op.mu)

#define OP_RESERVE(slot, bytes) { int rc_ = op_reserve(slot, bytes); if (rc_) return rc_; }

extern "C" int crsdr_convtosigned(const uint8_t *in, uint8_t *out, int n)
{
    OP_PROLOGUE(in && out && n > 0 && (n % 8) == 0, "convtosigned: need in, out, n > 0, n % 8 == 0");
    OP_RESERVE(0, n); OP_RESERVE(1, n);
    HIP_TRY(hipMemcpy(g_op.buf[0], in, n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_op_convtosigned, grid1d(n / 4), dim3(256), 0, 0, (const uint32_t *)g_op.buf[0], (uint32_t *)g_op.buf[1], n / 4);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, g_op.buf[1], n, hipMemcpyDeviceToHost));
    return CRSDR_OK;
}

extern "C" int crsdr_convtofloat(float *out, const int8_t *s8bit, int n)
{
    OP_PROLOGUE(out && s8bit && n > 0, "convtofloat: need out, s8bit, n > 0");
    OP_RESERVE(0, n); OP_RESERVE(1, sizeof(float) * (size_t)n);
    HIP_TRY(hipMemcpy(g_op.buf[0], s8bit, n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_op_convtofloat, grid1d(n), dim3(256), 0, 0, (float *)g_op.buf[1], (const int8_t *)g_op.buf[0], n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, g_op.buf[1], sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
    return CRSDR_OK;
}

extern "C" int crsdr_scalarmul(float *out, const float *in, float sre, float sim, int n)
{
    OP_PROLOGUE(out && in && n > 0, "scalarmul: need out, in, n > 0");
    const size_t bytes = sizeof(float2) * (size_t)n;
    OP_RESERVE(0, bytes); OP_RESERVE(1, bytes);
    HIP_TRY(hipMemcpy(g_op.buf[0], in, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_op_scalarmul, grid1d(n), dim3(256), 0, 0, (float2 *)g_op.buf[1], (const float2 *)g_op.buf[0], make_float2(sre, sim), n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, g_op.buf[1], bytes, hipMemcpyDeviceToHost));
    return CRSDR_OK;
}

extern "C" int crsdr_convto8bit(int8_t *out, const float *in, int n)
{
    OP_PROLOGUE(out && in && n > 0, "convto8bit: need out, in, n > 0");
    const size_t nf = 2 * (size_t)n; // the reference doubles n internally (src/cdsp.cc:52)
    OP_RESERVE(0, sizeof(float) * nf); OP_RESERVE(1, nf);
    HIP_TRY(hipMemcpy(g_op.buf[0], in, sizeof(float) * nf, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_op_convto8bit, grid1d(nf), dim3(256), 0, 0, (int8_t *)g_op.buf[1], (const float *)g_op.buf[0], (int)nf);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, g_op.buf[1], nf, hipMemcpyDeviceToHost));
    return CRSDR_OK;
}

extern "C" int crsdr_conj_dotproduct(float *res, const float *a, const float *b, int n)
{
    OP_PROLOGUE(res && a && b && n > 0, "conj_dotproduct: need res, a, b, n > 0");
    const size_t bytes = sizeof(float2) * (size_t)n;
    OP_RESERVE(0, bytes); OP_RESERVE(1, bytes); OP_RESERVE(2, 16);
    HIP_TRY(hipMemcpy(g_op.buf[0], a, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(g_op.buf[1], b, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_op_conj_dot, dim3(1), dim3(1024), 0, 0, (float *)g_op.buf[2], (const float2 *)g_op.buf[0], (const float2 *)g_op.buf[1], n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(res, g_op.buf[2], 2 * sizeof(float), hipMemcpyDeviceToHost));
    return CRSDR_OK;
}

extern "C" int crsdr_magsquared(float *out, const float *in, int n)
{
    OP_PROLOGUE(out && in && n > 0, "magsquared: need out, in, n > 0");
    OP_RESERVE(0, sizeof(float2) * (size_t)n); OP_RESERVE(1, sizeof(float) * (size_t)n);
    HIP_TRY(hipMemcpy(g_op.buf[0], in, sizeof(float2) * (size_t)n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_op_magsquared, grid1d(n), dim3(256), 0, 0, (float *)g_op.buf[1], (const float2 *)g_op.buf[0], n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, g_op.buf[1], sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
    return CRSDR_OK;
}

extern "C" int crsdr_conjugatemul(float *out, const float *in1, const float *in2, int n)
{
    OP_PROLOGUE(out && in1 && in2 && n > 0, "conjugatemul: need out, in1, in2, n > 0");
    const size_t bytes = sizeof(float2) * (size_t)n;
    OP_RESERVE(0, bytes); OP_RESERVE(1, bytes); OP_RESERVE(2, bytes);
    HIP_TRY(hipMemcpy(g_op.buf[0], in1, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(g_op.buf[1], in2, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_op_conjugatemul, grid1d(n), dim3(256), 0, 0, (float2 *)g_op.buf[2], (const float2 *)g_op.buf[0], (const float2 *)g_op.buf[1], n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, g_op.buf[2], bytes, hipMemcpyDeviceToHost));
    return CRSDR_OK;
}

extern "C" int crsdr_indexofmax(uint32_t *index, const float *in, int n)
{
    OP_PROLOGUE(index && in && n > 0, "indexofmax: need index, in, n > 0");
    OP_RESERVE(0, sizeof(float) * (size_t)n); OP_RESERVE(2, 16);
    HIP_TRY(hipMemcpy(g_op.buf[0], in, sizeof(float) * (size_t)n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_op_indexofmax, dim3(1), dim3(1024), 0, 0, (uint32_t *)g_op.buf[2], (const float *)g_op.buf[0], n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(index, g_op.buf[2], sizeof(uint32_t), hipMemcpyDeviceToHost));
    return CRSDR_OK;
}

extern "C" int crsdr_fft(float *out, const float *in, int n, int sign, int howmany)
{
    OP_PROLOGUE(out && in && howmany > 0 && (sign == 1 || sign == -1), "fft: need out, in, howmany > 0, sign = +-1");
    const int l2 = ilog2_exact(n);
    if (l2 < kMinLog2 || l2 > kMaxLog2)
        return fail(CRSDR_EINVAL, "fft: n = %d unsupported (power of two in [16, 16384])", n);
    const size_t bytes = sizeof(float2) * (size_t)n * (size_t)howmany;
    OP_RESERVE(0, bytes); OP_RESERVE(1, bytes);
    if (!g_op.tw[l2]) { int rc = make_twiddles(n, &g_op.tw[l2]); if (rc) return rc; }
    HIP_TRY(hipMemcpy(g_op.buf[0], in, bytes, hipMemcpyHostToDevice));
    const float2 *tw = g_op.tw[l2];
    hipError_t e = (sign < 0)
        ? CRSDR_DISPATCH_LOG2(l2, (launch_op_fft<LG, -1>(0, howmany, (float2 *)g_op.buf[1], (const float2 *)g_op.buf[0], tw)))
        : CRSDR_DISPATCH_LOG2(l2, (launch_op_fft<LG, +1>(0, howmany, (float2 *)g_op.buf[1], (const float2 *)g_op.buf[0], tw)));
    HIP_TRY(e);
    HIP_TRY(hipMemcpy(out, g_op.buf[1], bytes, hipMemcpyDeviceToHost));
    return CRSDR_OK;
}

// ---- (iii) downstream helper: sample covariance of the aligned matrix on the matrix cores ----
// K split of the tiled kernel: the split that minimises rounds-of-workgroups x a tile's time / S + the partial sums' write and read
// time x S.  A tile's time is what its CU takes to pull the 2 x 128 rows in (measured ~50 GB/s per CU from L2 under the MFMAs; the
// matrix cores' share, 2 MFMAs per operand pair at 9.8e12 int8 MACs/s per CU, is shorter); the partials move at ~5 TB/s.
static int cov_pick_split(int ntri, int npairs, int blocksize, int cus)
{
    const double mfma_us = 2.0 * cov::CT * cov::CT * (double)blocksize / 9.8e6, ingest_us = 2.0 * cov::CT * (double)blocksize / 5e4;
    const double tile_us = std::max(mfma_us, ingest_us), part_us = (double)ntri * 2 * cov::CT * cov::CT * 4 * 2 / 5e6;
    // a workgroup's int32 partial sums are exact while its K range is <= 65536 bytes (|sum| <= 32768 samples x 2 x 128^2 = 2^30)
    const int s_min = std::max(1, (blocksize + 65535) / 65536);
    int best = s_min;
    double best_cost = 1e30;
    for (int s = s_min; s <= std::max(s_min, std::min(npairs, 32)); ++s) {
        const double cost = (double)((ntri * s + cus - 1) / cus) / s * tile_us + s * part_us;
        if (cost < best_cost) { best_cost = cost; best = s; }
    }
    return best;
}

extern "C" int crsdr_covariance(float *rxx, const int8_t *matrix, int nrows, int blocksize, int mem_kind)
{
    if (!rxx || !matrix || nrows < 2 || blocksize < 32 || (blocksize % 32)) return fail(CRSDR_EINVAL, "covariance: need rxx, matrix, nrows >= 2, blocksize % 32 == 0");
    if (mem_kind != CRSDR_MEM_HOST && mem_kind != CRSDR_MEM_DEVICE) return fail(CRSDR_EINVAL, "covariance: mem_kind = %d", mem_kind);
    { int rc_ = require_device(); if (rc_) return rc_; }
    std::lock_guard<std::mutex> lock_(g_op.mu);
    const size_t mb = (size_t)nrows * (size_t)blocksize, nsig = (size_t)nrows - 1, rb = sizeof(float2) * nsig * nsig;
    const int8_t *d_m = matrix;
    float2 *d_r = (float2 *)rxx;
    if (mem_kind == CRSDR_MEM_HOST) {
        OP_RESERVE(0, mb); OP_RESERVE(1, rb);
        HIP_TRY(hipMemcpy(g_op.buf[0], matrix, mb, hipMemcpyHostToDevice));
        d_m = (const int8_t *)g_op.buf[0];
        d_r = (float2 *)g_op.buf[1];
    }
    // (the one-tile-per-wave kernel keeps a whole row's sums in int32: exact up to 65536 bytes per row; longer rows take the tiled path,
    // whose K split keeps every partial inside that bound, whatever the channel count)
    if (blocksize > 65536 && (blocksize % (2 * cov::KC) || (uintptr_t)d_m % 4))
        return fail(CRSDR_EINVAL, "covariance: blocksize above 65536 must be a multiple of %d (and the matrix 4-byte aligned)", 2 * cov::KC);
    if (blocksize % (2 * cov::KC) == 0 && (uintptr_t)d_m % 4 == 0 && (nsig >= 64 || blocksize > 65536)) {
        // LDS-tiled form: 128 x 128 tiles on / above the diagonal, the K range split (in pairs of chunks) over the grid
        const int nt = (int)((nsig + cov::CT - 1) / cov::CT), ntri = nt * (nt + 1) / 2, npairs = blocksize / (2 * cov::KC);
        const int S = cov_pick_split(ntri, npairs, blocksize, device_cus());
        const unsigned grid = 8u * (unsigned)((ntri * S + 7) / 8);
        OP_RESERVE(2, sizeof(int2) * (size_t)nt * cov::CT * (size_t)S);
        OP_RESERVE(3, sizeof(int) * 2 * cov::CT * cov::CT * (size_t)ntri * (size_t)S);
        HIP_TRY(hipFuncSetAttribute((const void *)cov::k_covariance_tiled, hipFuncAttributeMaxDynamicSharedMemorySize, cov::COV_LDS_BYTES));
        hipLaunchKernelGGL(cov::k_covariance_tiled, dim3(grid), dim3(cov::COV_THREADS), cov::COV_LDS_BYTES, 0, d_m, nrows, blocksize, nt, ntri, S, (int *)g_op.buf[3],
                           (int2 *)g_op.buf[2]);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(cov::k_cov_reduce, dim3((unsigned)ntri, cov::CT / cov::CR), dim3(256), 0, 0, (const int *)g_op.buf[3], (const int2 *)g_op.buf[2], S, ntri, nt,
                           nrows, blocksize, d_r);
        HIP_TRY(hipGetLastError());
    } else {
        const unsigned tiles = (unsigned)((nsig + 63) / 64);
        OP_RESERVE(2, sizeof(int2) * (size_t)nrows);
        hipLaunchKernelGGL(cov::k_row_sums, dim3(nrows), dim3(256), 0, 0, d_m, blocksize, (int2 *)g_op.buf[2]);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(cov::k_covariance, dim3(tiles, tiles), dim3(256), 0, 0, d_m, nrows, blocksize, (const int2 *)g_op.buf[2], d_r);
        HIP_TRY(hipGetLastError());
    }
    if (mem_kind == CRSDR_MEM_HOST) HIP_TRY(hipMemcpy(rxx, d_r, rb, hipMemcpyDeviceToHost));
    else HIP_TRY(hipDeviceSynchronize());
    return CRSDR_OK;
}

extern "C" int crsdr_noisesubspace(float *vec, float *sv, const float *rxx, int m, int mem_kind)
{
    if (!vec || !rxx || m < 2 || m > music::MAX_M) return fail(CRSDR_EINVAL, "noisesubspace: need vec, rxx, 2 <= m <= %d", music::MAX_M);
    if (mem_kind != CRSDR_MEM_HOST && mem_kind != CRSDR_MEM_DEVICE) return fail(CRSDR_EINVAL, "noisesubspace: mem_kind = %d", mem_kind);
    { int rc_ = require_device(); if (rc_) return rc_; }
    std::lock_guard<std::mutex> lock_(g_op.mu);
    const size_t mm = sizeof(float2) * (size_t)m * m;
    OP_RESERVE(2, music::MAX_M * sizeof(float) + 2 * sizeof(int));
    float *d_sv = (float *)g_op.buf[2];
    int *d_info = (int *)((char *)g_op.buf[2] + music::MAX_M * sizeof(float));
    const float2 *d_r = (const float2 *)rxx;
    float2 *d_v = (float2 *)vec;
    if (mem_kind == CRSDR_MEM_HOST) {
        OP_RESERVE(0, mm); OP_RESERVE(1, mm);
        HIP_TRY(hipMemcpy(g_op.buf[0], rxx, mm, hipMemcpyHostToDevice));
        d_r = (const float2 *)g_op.buf[0];
        d_v = (float2 *)g_op.buf[1];
    }
    const size_t lds = 2 * sizeof(double2) * (size_t)m * m;
    HIP_TRY(hipFuncSetAttribute((const void *)music::k_herm_subspace, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(music::k_herm_subspace, dim3(1), dim3(music::JT), lds, 0, d_r, m, d_sv, d_v, d_info);
    HIP_TRY(hipGetLastError());
    int info[2] = {0, 0};
    HIP_TRY(hipMemcpy(info, d_info, sizeof(info), hipMemcpyDeviceToHost));
    if (mem_kind == CRSDR_MEM_HOST) {
        HIP_TRY(hipMemcpy(vec, d_v, mm, hipMemcpyDeviceToHost));
        if (sv) HIP_TRY(hipMemcpy(sv, d_sv, sizeof(float) * (size_t)m, hipMemcpyDeviceToHost));
    } else if (sv) {
        HIP_TRY(hipMemcpy(sv, d_sv, sizeof(float) * (size_t)m, hipMemcpyDeviceToDevice));
    }
    if (!info[1]) return fail(CRSDR_ESTATE, "noisesubspace: Jacobi iteration not converged after %d sweeps", info[0]);
    return CRSDR_OK;
}

extern "C" int crsdr_pmusic2d(float *pm, const float *vec, int m, int k, float d, int mx, int my, int ncx, int ncy, int mem_kind)
{
    if (!pm || !vec || m < 2 || m > music::MAX_M || mx < 1 || my < 1 || mx * my != m || k < 1 || k >= m || ncx < 1 || ncy < 1 ||
        (long long)ncx * ncy > (1 << 24))
        return fail(CRSDR_EINVAL, "pmusic2d: need pm, vec, m = mx*my in [2, %d], 1 <= k < m, grid <= 2^24 points", music::MAX_M);
    if (mem_kind != CRSDR_MEM_HOST && mem_kind != CRSDR_MEM_DEVICE) return fail(CRSDR_EINVAL, "pmusic2d: mem_kind = %d", mem_kind);
    { int rc_ = require_device(); if (rc_) return rc_; }
    std::lock_guard<std::mutex> lock_(g_op.mu);
    const size_t mm = sizeof(float2) * (size_t)m * m, pb = sizeof(float) * (size_t)ncx * ncy;
    const float2 *d_v = (const float2 *)vec;
    float *d_p = pm;
    if (mem_kind == CRSDR_MEM_HOST) {
        OP_RESERVE(0, mm); OP_RESERVE(1, pb);
        HIP_TRY(hipMemcpy(g_op.buf[0], vec, mm, hipMemcpyHostToDevice));
        d_v = (const float2 *)g_op.buf[0];
        d_p = (float *)g_op.buf[1];
    }
    const int nn = m - k;
    const size_t lds = sizeof(float2) * ((size_t)m * nn + (size_t)m * music::PT);
    HIP_TRY(hipFuncSetAttribute((const void *)music::k_pmusic2d, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned blocks = (unsigned)(((size_t)ncx * ncy + music::PT - 1) / music::PT);
    hipLaunchKernelGGL(music::k_pmusic2d, dim3(blocks), dim3(music::PT), lds, 0, d_v, m, m, k, nn, d, mx, my, ncx, ncy, d_p);
    HIP_TRY(hipGetLastError());
    if (mem_kind == CRSDR_MEM_HOST) HIP_TRY(hipMemcpy(pm, d_p, pb, hipMemcpyDeviceToHost));
    else HIP_TRY(hipDeviceSynchronize());
    return CRSDR_OK;
}

// ================================================================================================
// (ii) batched plan
// ================================================================================================
constexpr int kStageSlots = 4;
constexpr int kMaxBatch = 64;

struct crsdr_plan {
    int nrows = 0, B = 0, L = 0, log2n = 0, mode = 0, device = 0, row_begin = 1, row_count = 0, max_batch = 1;
    size_t packet_bytes = 0, matrix_off = 0, packet_stride = 0, own_packet_stride = 0;
    hipStream_t own_stream = nullptr, stream = nullptr, aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_ref[2] = {nullptr, nullptr}, ev_k1done[2] = {nullptr, nullptr};
    // per-batch {lag, mag, frac} are double-buffered (obuf: a pipelined fetch of batch b reads its buffer while batch b + 1 writes the
    // other), ev_k2done[i] = the last kernel of the batch that wrote buffer i has finished
    hipStream_t cs = nullptr;                       // copy stream of crsdr_plan_fetch_batch_async
    hipEvent_t ev_copydone[4] = {nullptr, nullptr, nullptr, nullptr};   // one per outstanding asynchronous fetch (ring)
    unsigned long copy_head = 0, copy_tail = 0;      // fetches waited for / issued
    int *h_k1flag = nullptr;                         // page-locked [4]: the two-row K1's error word as copied by each asynchronous fetch
    bool copy_pending = false;                       // the next submit's kernels wait for the newest of them
    hipEvent_t ev_k2done[2] = {nullptr, nullptr};
    bool k2done_valid[2] = {false, false};
    int obuf = 0;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool k1done_valid[2] = {false, false};
    int slot = 0;
    float2 *d_tw = nullptr, *d_twA = nullptr, *d_twB = nullptr, *d_refspec[2] = {nullptr, nullptr}; // [T][B] each
    int8_t *d_rows = nullptr;          // staging for host input [T][nrows][B]
    uint8_t *d_packet_alloc = nullptr; // own packet allocation (front padding for alignment)
    int8_t *d_packet_own = nullptr, *d_packet = nullptr;
    int8_t *d_slab = nullptr;          // slab output mode (crsdr_plan_bind_slab), caller-owned
    size_t slab_stride = 0, tail_offset = 0;   // tail_offset: crsdr_plan_bind_slab_ex -- per-row scalars after the rows of every slot
    int hdr_first = 0, hdr_count = 0;
    uint32_t *d_readcnt = nullptr;     // [T][nrows]
    uint8_t *d_mask = nullptr;
    int32_t *d_lag = nullptr, *d_lag_state = nullptr;         // [T][nrows], [nrows]
    float *d_mag = nullptr, *d_frac = nullptr, *d_mag_state = nullptr, *d_frac_state = nullptr;
    float2 *d_phasor = nullptr, *d_phase_state[2] = {nullptr, nullptr}; // [T][nrows], 2 x [nrows]
    long long *d_corr = nullptr;       // [T][nrows][2]: integer sums (three-kernel path) or the fused kernel's two hand-over slots
    int chain_slot = 0;                // fused path: slot [chain_slot] of every entry carries this batch's unit phasors,
    int chain_armed[2] = {0, 0};       // the kernel re-arms the other one; leading blocks of each slot known to be all-ones
    unsigned int *d_sync = nullptr;    // fused K2: [0] ticket counter, [1] status; [2] two-row K1: waits that ran out
    unsigned int q_work_base = 0;      // value of the two-row K1's work counter (d_sync[3]) when the next launch starts
    bool k1_used = false;              // a two-row K1 launch is (or was) in flight: check [2] at the next sync
    // carried state {lag, mag, frac}_state + phase_state[0..1] lives in ONE allocation (d_state) so that the first two-row
    // launch after a clean status check can snapshot it with one stream-ordered copy; a wait that ran out rolls the
    // plan back to that snapshot (check_fused_status) and keeps later launches on the packed kernel
    uint8_t *d_state = nullptr, *d_state_snap = nullptr;   // d_state_snap: kSnaps copies (the ring below)
    size_t state_bytes = 0;
    bool q_disabled = false;
    // Ring of stream-ordered snapshots of the carried state, one per batch that used a two-row / two-line kernel (entry = the
    // state BEFORE that batch).  A status word read clean behind batch k retires every snapshot older than the state before
    // batch k + 1; a wait that ran out rolls back to the OLDEST unretired one.  When the ring is full no new snapshot is
    // taken (the rollback then simply reaches further back).
    static constexpr int kSnaps = 8;
    struct Snap { unsigned long idx; int phase_cur; } snaps[kSnaps] = {};
    int snap_head = 0, snap_cnt = 0;
    unsigned long submit_idx = 0;          // batches submitted since create / reset / rollback
    unsigned long copy_idx[4] = {0, 0, 0, 0};   // submit index of the batch each outstanding asynchronous fetch belongs to
    bool fused_k2 = true, fused_used = false;   // CRSDR_K2_FUSED=0: the three-kernel phase path
    // B = 16384: the blocks' reference spectra as the first work items of the cross-correlation launch itself (CRSDR_K1_FOLD=0: from
    // k_ref_spectrum14p on the aux stream, two cross-stream dependencies and an event record per batch more)
    bool fold = true;
    unsigned int *d_refflag = nullptr, refgen = 0;     // [max_batch] words: the launch generation whose spectrum of block t is in place
    int phase_cur = 0;
    int last_nblocks = 0;
    // long-block path (B > 16384): B = N1 x 16384
    bool longblock = false;
    int log2n1 = 0;
    float2 *d_wc = nullptr, *d_wf = nullptr, *d_tw1 = nullptr, *d_Y = nullptr, *d_Yref = nullptr;
    lb::LongPartial *d_part = nullptr;
    int long_chunk = 1 << 30; // rows per pass of the long-block stages
    // fractional-delay correction (crsdr_plan_set_frac_apply; long blocks, digital mode)
    bool frac_apply = false, frac_override_on = false;
    float frac_gain = 1.0f;
    float *d_frac_override = nullptr;   // [nrows]
    uint32_t *d_k2tab = nullptr;        // [8192] frequency index k2 of every junction register pair of the row transforms
    bool frac_generic = false;          // CRSDR_FRAC_GENERIC=1: B = 16384 takes the generic fractional-delay kernel too (A/B, tests)
    float2 *d_Z = nullptr;              // second work area (set_frac_apply): the correlation pass's stage B writes here, so that the apply pass finds stage A's output still in d_Y
    float4 *d_rowspec = nullptr;        // [row_count][8192] the rows' responses G(k2) in the junction's order (k_ramp_rowspec); allocated by set_frac_apply
    // pinned staging ring for the small per-batch host arrays
    uint32_t *h_readcnt = nullptr;
    uint8_t *h_mask = nullptr;
    hipEvent_t ev_stage[kStageSlots] = {};
    bool stage_valid[kStageSlots] = {};
    int stage_slot = 0;
    bool submitted = false;
    // optional per-kernel event pairs (crsdr_plan_enable_profiling)
    int prof_slots = 0;
    uint32_t prof_mask = 0;              // bit k: kernel k gets an event pair; bit 31: whole-submit start/stop
    long prof_count = 0;                 // submits recorded since enable
    std::vector<hipEvent_t> prof_ev;     // [slot][kernel][begin,end]
    std::vector<unsigned char> prof_has; // [slot][kernel]
};

constexpr int kProfKernels = 4;

static hipEvent_t *prof_pair(crsdr_plan *p, int which)
{
    if (!p->prof_slots || !(p->prof_mask & (1u << which))) return nullptr;
    const int slot = (int)(p->prof_count % p->prof_slots);
    p->prof_has[(size_t)slot * kProfKernels + which] = 1;
    return &p->prof_ev[((size_t)slot * kProfKernels + which) * 2];
}

static int plan_init_state(crsdr_plan *p)
{
    const size_t n = (size_t)p->nrows, T = (size_t)p->max_batch;
    std::vector<float2> ph(n, make_float2(1.0f, 0.0f)); // src/csdrdevice.cc:39-40
    ph[0] = make_float2(0.0f, 0.0f);                     // pcorrection[0] is never written
    for (int i = 0; i < 2; ++i) HIP_TRY(hipMemcpy(p->d_phase_state[i], ph.data(), sizeof(float2) * n, hipMemcpyHostToDevice));
    for (size_t t = 0; t < T; ++t) HIP_TRY(hipMemcpy(p->d_phasor + t * n, ph.data(), sizeof(float2) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(p->d_lag, 0, sizeof(int32_t) * n * T * 2));
    HIP_TRY(hipMemset(p->d_mag, 0, sizeof(float) * n * T * 2));
    HIP_TRY(hipMemset(p->d_frac, 0, sizeof(float) * n * T * 2));
    HIP_TRY(hipMemset(p->d_lag_state, 0, sizeof(int32_t) * n));
    HIP_TRY(hipMemset(p->d_mag_state, 0, sizeof(float) * n));
    HIP_TRY(hipMemset(p->d_frac_state, 0, sizeof(float) * n));
    HIP_TRY(hipMemset(p->d_corr, 0xFF, sizeof(long long) * 4 * n * T));
    p->chain_armed[0] = p->chain_armed[1] = (int)T;
    p->phase_cur = 0;
    p->last_nblocks = 0;
    p->snap_cnt = 0; p->snap_head = 0;
    return CRSDR_OK;
}

static hipError_t launch_long_rows_ref1(hipStream_t s, float2 *Y, const float2 *twA, const float2 *twB, float2 *refspec);
// Which frequency k2 of the 16384-point row transform sits in which junction register?  Transform a one-hot line
// x[n] = delta[n - 1] with the reference-row form of the kernel (forward, conj, refspec layout): slot q then holds
// exp(+2 pi i k2 / 16384) twice over, and k2 is read off the angle.  Done once per long-block plan; the table must be a
// permutation of 0 .. 16383 or the plan is refused.
static int make_k2tab(crsdr_plan *p)
{
    const size_t N2 = lb::N2;
    float2 *line = p->d_Yref;              // long blocks have a line of their own; a 16384-point plan borrows one for this call
    if (!line) HIP_TRY(hipMalloc((void **)&line, sizeof(float2) * N2));
    const auto give_back = [&]() { if (line != p->d_Yref) (void)hipFree(line); };
    hipError_t e0 = hipMemset(line, 0, sizeof(float2) * N2);
    const float2 one = make_float2(1.0f, 0.0f);
    if (e0 == hipSuccess) e0 = hipMemcpy(line + 1, &one, sizeof(one), hipMemcpyHostToDevice);
    if (e0 == hipSuccess) e0 = hipDeviceSynchronize();       // the plan's streams are non-blocking: no implicit order with the null stream
    if (e0 == hipSuccess) e0 = launch_long_rows_ref1(p->own_stream, line, p->d_twA, p->d_twB, p->d_refspec[0]);
    if (e0 == hipSuccess) e0 = hipStreamSynchronize(p->own_stream);
    give_back();
    HIP_TRY(e0);
    std::vector<float2> h(N2);
    HIP_TRY(hipMemcpy(h.data(), p->d_refspec[0], sizeof(float2) * N2, hipMemcpyDeviceToHost));
    std::vector<uint32_t> tab(N2 / 2);
    std::vector<unsigned char> seen(N2, 0);
    for (size_t i = 0; i < N2; ++i) {
        const double rev = std::atan2((double)h[i].y, (double)h[i].x) / (2.0 * M_PI);      // conj(W^k2) = exp(+2 pi i k2 / N2)
        const long k2 = ((long)std::llround(rev * (double)N2) % (long)N2 + (long)N2) % (long)N2;
        const double mag = std::hypot((double)h[i].x, (double)h[i].y);
        if (std::fabs(mag - 1.0) > 1e-3 || seen[(size_t)k2]) return fail(CRSDR_EHIP, "plan_create: frequency map of the row transform is not a permutation (entry %zu)", i);
        seen[(size_t)k2] = 1;
        if (i & 1) tab[i >> 1] |= (uint32_t)k2 << 16; else tab[i >> 1] = (uint32_t)k2;
    }
    HIP_TRY(hipMalloc((void **)&p->d_k2tab, sizeof(uint32_t) * tab.size()));
    HIP_TRY(hipMemcpy(p->d_k2tab, tab.data(), sizeof(uint32_t) * tab.size(), hipMemcpyHostToDevice));
    return CRSDR_OK;
}

static int plan_alloc(crsdr_plan *p)
{
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamCreateWithFlags(&p->own_stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&p->aux, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&p->cs, hipStreamNonBlocking));
    for (auto &e : p->ev_copydone) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_TRY(hipHostMalloc((void **)&p->h_k1flag, 4 * sizeof(int), hipHostMallocDefault));
    memset(p->h_k1flag, 0, 4 * sizeof(int));
    p->stream = p->own_stream;
    hipEvent_t *evs[] = {&p->ev_fork, &p->ev_ref[0], &p->ev_ref[1], &p->ev_k1done[0], &p->ev_k1done[1], &p->ev_k2done[0], &p->ev_k2done[1]};
    for (auto e : evs) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    for (int i = 0; i < kStageSlots; ++i) HIP_TRY(hipEventCreateWithFlags(&p->ev_stage[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreate(&p->ev_start));
    HIP_TRY(hipEventCreate(&p->ev_stop));
    { int rc = make_twiddles(p->longblock ? lb::N2 : p->B, &p->d_tw); if (rc) return rc; }
    if (p->log2n == 14 || p->longblock) { int rc = make_twiddles14(&p->d_twA, &p->d_twB); if (rc) return rc; }
    if (p->longblock) {
        const size_t Bz = (size_t)p->B, n1 = Bz >> lb::LOG2N2;
        int rc;
        if ((rc = make_twiddle_table((double)Bz, Bz >> lb::FBITS, (size_t)1 << lb::FBITS, &p->d_wc))) return rc;
        if ((rc = make_twiddle_table((double)Bz, (size_t)1 << lb::FBITS, 1, &p->d_wf))) return rc;
        if ((rc = make_twiddle_table((double)n1, n1, 1, &p->d_tw1))) return rc;
        HIP_TRY(hipMalloc((void **)&p->d_Y, sizeof(float2) * Bz * (size_t)p->row_count));
        HIP_TRY(hipMalloc((void **)&p->d_Yref, sizeof(float2) * Bz));
        {
            const char *e = getenv("CRSDR_LONG_CHUNK_MB");
            // r01, cfg5 (21 x 2^21, 352 MB): 64 MB -4 %, 128 MB +1.5 %, 200 MB +4.7 % against unchunked.  With the r02 column stages
            // (register-first passes, persistent stage A) the launch boundaries weigh more than the cache residency of the work
            // area: 64 MB 1 952, 100 MB 2 061, 136 MB 2 197, 170 MB 2 257, 200 MB 2 370, 240 MB 2 326, one chunk 2 460 blocks/s.
            const long mb = e ? atol(e) : 400;
            if (mb > 0) p->long_chunk = (int)std::max<size_t>(1, ((size_t)mb << 20) / (sizeof(float2) * Bz));
        }
        HIP_TRY(hipMalloc((void **)&p->d_part, sizeof(lb::LongPartial) * (size_t)lb::ntiles(p->log2n1) * (size_t)p->row_count));
    }
    const size_t n = (size_t)p->nrows, T = (size_t)p->max_batch;
    const size_t rowbytes = n * (size_t)p->B;
    for (int i = 0; i < 2; ++i) HIP_TRY(hipMalloc((void **)&p->d_refspec[i], sizeof(float2) * (size_t)p->B * T));
    HIP_TRY(hipMalloc((void **)&p->d_rows, rowbytes * T));
    // own packets: pad the front so that the matrix (at +16+4N) starts 256-byte aligned; the
    // stride between the packets of a batch keeps that alignment
    const size_t pad = (256 - (p->matrix_off % 256)) % 256;
    p->own_packet_stride = (p->packet_bytes + 255) / 256 * 256;
    p->packet_stride = p->own_packet_stride;
    HIP_TRY(hipMalloc((void **)&p->d_packet_alloc, pad + p->own_packet_stride * T + 256));
    HIP_TRY(hipMemset(p->d_packet_alloc, 0, pad + p->own_packet_stride * T + 256));
    p->d_packet_own = reinterpret_cast<int8_t *>(p->d_packet_alloc + pad);
    p->d_packet = p->d_packet_own;
    HIP_TRY(hipMalloc((void **)&p->d_readcnt, sizeof(uint32_t) * n * T));
    HIP_TRY(hipMalloc((void **)&p->d_mask, n));
    HIP_TRY(hipMalloc((void **)&p->d_lag, sizeof(int32_t) * n * T * 2));   // two batches: K1 of batch b+1 runs beside the phase kernel of batch b
    HIP_TRY(hipMalloc((void **)&p->d_mag, sizeof(float) * n * T * 2));
    HIP_TRY(hipMalloc((void **)&p->d_frac, sizeof(float) * n * T * 2));
    HIP_TRY(hipMalloc((void **)&p->d_phasor, sizeof(float2) * n * T));
    HIP_TRY(hipMalloc((void **)&p->d_corr, sizeof(long long) * 4 * n * T));   // unit words, then chain values (k_align_fused)
    HIP_TRY(hipMalloc((void **)&p->d_sync, 64));
    HIP_TRY(hipMemset(p->d_sync, 0, 64));
    HIP_TRY(hipMalloc((void **)&p->d_refflag, sizeof(unsigned int) * T));
    HIP_TRY(hipMemset(p->d_refflag, 0, sizeof(unsigned int) * T));
    { const char *e = getenv("CRSDR_K1_FOLD"); if (e) p->fold = atoi(e) != 0; }
    { const char *e = getenv("CRSDR_K2_FUSED"); if (e) p->fused_k2 = atoi(e) != 0; }
    { const char *e = getenv("CRSDR_FRAC_GENERIC"); if (e) p->frac_generic = atoi(e) != 0; }
    {
        const size_t n8 = (n + 1) / 2 * 2;                   // keeps the float2 arrays 8-byte aligned
        p->state_bytes = n8 * (3 * 4 + 2 * 8);
        HIP_TRY(hipMalloc((void **)&p->d_state, p->state_bytes));
        HIP_TRY(hipMalloc((void **)&p->d_state_snap, p->state_bytes * crsdr_plan::kSnaps));
        p->d_phase_state[0] = reinterpret_cast<float2 *>(p->d_state);
        p->d_phase_state[1] = p->d_phase_state[0] + n8;
        p->d_lag_state = reinterpret_cast<int32_t *>(p->d_phase_state[1] + n8);
        p->d_mag_state = reinterpret_cast<float *>(p->d_lag_state + n8);
        p->d_frac_state = p->d_mag_state + n8;
    }
    HIP_TRY(hipHostMalloc((void **)&p->h_readcnt, sizeof(uint32_t) * n * T * kStageSlots, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&p->h_mask, n * kStageSlots, hipHostMallocDefault));
    if (p->longblock || p->log2n == 14) { int rc = make_k2tab(p); if (rc) return rc; }      // (short blocks: the fractional-delay pass on K1's network)
    HIP_TRY(hipMalloc((void **)&p->d_frac_override, sizeof(float) * (size_t)p->nrows));
    HIP_TRY(hipMemset(p->d_frac_override, 0, sizeof(float) * (size_t)p->nrows));
    return plan_init_state(p);
}

static void plan_free(crsdr_plan *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->own_stream) (void)hipStreamSynchronize(p->own_stream);
    if (p->aux) (void)hipStreamSynchronize(p->aux);
    if (p->cs) { (void)hipStreamSynchronize(p->cs); (void)hipStreamDestroy(p->cs); }
    for (auto e : p->ev_copydone) if (e) (void)hipEventDestroy(e);
    void *bufs[] = {p->d_Z, p->d_rowspec, p->d_frac_override, p->d_k2tab, p->d_wc, p->d_wf, p->d_tw1, p->d_Y, p->d_Yref, p->d_part, p->d_tw, p->d_twA, p->d_twB, p->d_refspec[0], p->d_refspec[1], p->d_rows, p->d_packet_alloc, p->d_readcnt,
                    p->d_mask, p->d_lag, p->d_mag, p->d_frac, p->d_phasor, p->d_corr, p->d_sync, p->d_state, p->d_state_snap, p->d_refflag};
    for (void *b : bufs) if (b) (void)hipFree(b);
    if (p->h_k1flag) (void)hipHostFree(p->h_k1flag);
    if (p->h_readcnt) (void)hipHostFree(p->h_readcnt);
    if (p->h_mask) (void)hipHostFree(p->h_mask);
    hipEvent_t evs[] = {p->ev_fork, p->ev_ref[0], p->ev_ref[1], p->ev_k1done[0], p->ev_k1done[1], p->ev_k2done[0], p->ev_k2done[1], p->ev_start, p->ev_stop};
    for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < kStageSlots; ++i) if (p->ev_stage[i]) (void)hipEventDestroy(p->ev_stage[i]);
    for (hipEvent_t e : p->prof_ev) if (e) (void)hipEventDestroy(e);
    if (p->own_stream) (void)hipStreamDestroy(p->own_stream);
    if (p->aux) (void)hipStreamDestroy(p->aux);
    delete p;
}

extern "C" int crsdr_plan_create(crsdr_plan **plan, const crsdr_plan_desc *desc)
{
    if (!plan || !desc) return fail(CRSDR_EINVAL, "plan_create: NULL argument");
    *plan = nullptr;
    const int l2 = ilog2_exact(desc->blocksize);
    if (desc->nrows < 2) return fail(CRSDR_EINVAL, "plan_create: nrows = %d (need the ref row and >= 1 signal row)", desc->nrows);
    if (l2 < kMinLog2 || l2 > kMaxLog2Plan)
        return fail(CRSDR_EINVAL, "plan_create: blocksize = %d unsupported (power of two in [16, 4194304])", desc->blocksize);
    if (desc->mode != CRSDR_MODE_FAITHFUL && desc->mode != CRSDR_MODE_DIGITAL)
        return fail(CRSDR_EINVAL, "plan_create: mode = %d", desc->mode);
    const int rb = desc->row_begin ? desc->row_begin : 1;
    const int rc_rows = desc->row_count ? desc->row_count : desc->nrows - rb;
    if (rb < 1 || rc_rows < 1 || rb + rc_rows > desc->nrows)
        return fail(CRSDR_EINVAL, "plan_create: slab [%d,%d) outside signal rows [1,%d)", rb, rb + rc_rows, desc->nrows);
    const int mb = desc->max_batch ? desc->max_batch : 1;
    if (mb < 1 || mb > kMaxBatch) return fail(CRSDR_EINVAL, "plan_create: max_batch = %d (1..%d)", mb, kMaxBatch);
    if (l2 > kMaxLog2 && mb != 1) return fail(CRSDR_EINVAL, "plan_create: long blocks (blocksize > 16384) take max_batch = 1");
    { int rc = require_device(); if (rc) return rc; }
    int ndev = 0;
    (void)crsdr_device_count(&ndev);
    if (desc->device < 0 || desc->device >= ndev) return fail(CRSDR_ENODEV, "plan_create: device %d of %d", desc->device, ndev);

    crsdr_plan *p = new (std::nothrow) crsdr_plan();
    if (!p) return fail(CRSDR_ENOMEM, "plan_create: out of host memory");
    p->nrows = desc->nrows; p->B = desc->blocksize; p->L = p->B / 2; p->log2n = l2; p->mode = desc->mode;
    p->device = desc->device; p->row_begin = rb; p->row_count = rc_rows; p->max_batch = mb;
    p->longblock = l2 > kMaxLog2; p->log2n1 = p->longblock ? l2 - lb::LOG2N2 : 0;
    p->matrix_off = 16 + 4 * (size_t)p->nrows;
    p->packet_bytes = p->matrix_off + (size_t)p->nrows * (size_t)p->B;
    int rc = plan_alloc(p);
    if (rc) { plan_free(p); return rc; }
    *plan = p;
    return CRSDR_OK;
}

extern "C" int crsdr_plan_destroy(crsdr_plan *plan)
{
    if (!plan) return fail(CRSDR_EINVAL, "plan_destroy: NULL plan");
    plan_free(plan);
    return CRSDR_OK;
}

static int check_fused_status(crsdr_plan *p);
extern "C" int crsdr_plan_sync(crsdr_plan *p)
{
    if (!p) return fail(CRSDR_EINVAL, "plan_sync: NULL plan");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipStreamSynchronize(p->aux));
    HIP_TRY(hipStreamSynchronize(p->cs));
    p->copy_pending = false;
    p->copy_head = p->copy_tail;
    return check_fused_status(p);
}

extern "C" int crsdr_plan_reset(crsdr_plan *p)
{
    if (!p) return fail(CRSDR_EINVAL, "plan_reset: NULL plan");
    int rc = crsdr_plan_sync(p);
    if (rc) return rc;
    p->submitted = false;
    return plan_init_state(p);
}

extern "C" int crsdr_plan_set_stream(crsdr_plan *p, void *hip_stream)
{
    if (!p) return fail(CRSDR_EINVAL, "plan_set_stream: NULL plan");
    int rc = crsdr_plan_sync(p);
    if (rc) return rc;
    p->stream = hip_stream ? (hipStream_t)hip_stream : p->own_stream;
    return CRSDR_OK;
}

extern "C" size_t crsdr_plan_packet_bytes(const crsdr_plan *p) { return p ? p->packet_bytes : 0; }
extern "C" size_t crsdr_plan_matrix_offset(const crsdr_plan *p) { return p ? p->matrix_off : 0; }
extern "C" int crsdr_plan_bind_slab(crsdr_plan *p, void *device_slab, size_t slab_stride, int hdr_first, int hdr_count)
{
    if (!p) return fail(CRSDR_EINVAL, "plan_bind_slab: NULL plan");
    p->tail_offset = 0;
    if (!device_slab) { p->d_slab = nullptr; p->slab_stride = 0; p->hdr_first = p->hdr_count = 0; return CRSDR_OK; }
    if ((uintptr_t)device_slab % 4 || slab_stride % 4 || slab_stride < (size_t)p->row_count * (size_t)p->B || hdr_first < 0 || hdr_count < 0 ||
        hdr_first + hdr_count > p->max_batch)
        return fail(CRSDR_EINVAL, "plan_bind_slab: slab and stride 4-byte aligned, stride >= row_count*blocksize, header range inside the batch");
    p->d_slab = (int8_t *)device_slab; p->slab_stride = slab_stride; p->hdr_first = hdr_first; p->hdr_count = hdr_count;
    return CRSDR_OK;
}

extern "C" int crsdr_assemble_slabs(void *device_packets, size_t packet_stride, int nrows, int blocksize, const void *device_recv, int nsrc,
                                    int nblocks, void *hip_stream)
{
    if (!device_packets || !device_recv || nrows < 2 || blocksize < 16 || nsrc < 1 || nblocks < 1 || (nrows - 1) % nsrc)
        return fail(CRSDR_EINVAL, "assemble_slabs: need packets, recv, nsrc dividing the signal rows, nblocks >= 1");
    const size_t body_off = 16 + 4 * (size_t)nrows + (size_t)blocksize;          // row 1 of the matrix
    const size_t slab_bytes = (size_t)((nrows - 1) / nsrc) * (size_t)blocksize;
    if (((uintptr_t)device_packets + body_off) % 4 || packet_stride % 4 || (uintptr_t)device_recv % 4 ||
        (nblocks > 1 && packet_stride < body_off + (size_t)(nrows - 1) * blocksize) || nblocks > 65535 || nsrc > 65535)
        return fail(CRSDR_EINVAL, "assemble_slabs: 4-byte alignment of the matrix rows, stride >= packet bytes");
    { int rc_ = require_device(); if (rc_) return rc_; }
    hipStream_t s = (hipStream_t)hip_stream;
    const bool v16 = (((uintptr_t)device_packets + body_off) % 16 == 0) && (packet_stride % 16 == 0) && ((uintptr_t)device_recv % 16 == 0) && (slab_bytes % 16 == 0);
    const size_t words = slab_bytes / (v16 ? 16 : 4);
    const unsigned chunks = (unsigned)std::min<size_t>(std::max<size_t>(1, words / 1024), 64);
    if (v16) hipLaunchKernelGGL(k_assemble_slabs<uint4>, dim3(chunks, nblocks, nsrc), dim3(256), 0, s, (int8_t *)device_packets, packet_stride, body_off, (const uint4 *)device_recv, nblocks, words);
    else hipLaunchKernelGGL(k_assemble_slabs<uint32_t>, dim3(chunks, nblocks, nsrc), dim3(256), 0, s, (int8_t *)device_packets, packet_stride, body_off, (const uint32_t *)device_recv, nblocks, words);
    HIP_TRY(hipGetLastError());
    return CRSDR_OK;
}

extern "C" size_t crsdr_plan_packet_stride(const crsdr_plan *p) { return p ? p->packet_stride : 0; }

extern "C" int crsdr_plan_device_buffers(crsdr_plan *p, void **packet, void **lag, void **mag, void **frac, void **phasor)
{
    if (!p) return fail(CRSDR_EINVAL, "plan_device_buffers: NULL plan");
    if (packet) *packet = p->d_packet;
    const size_t ob = (size_t)p->obuf * (size_t)p->nrows * (size_t)p->max_batch;   // the buffers of the last submit
    if (lag) *lag = p->d_lag + ob;
    if (mag) *mag = p->d_mag + ob;
    if (frac) *frac = p->d_frac + ob;
    if (phasor) *phasor = p->d_phasor;
    return CRSDR_OK;
}

extern "C" int crsdr_plan_bind_packet(crsdr_plan *p, void *device_packet, size_t packet_stride)
{
    if (!p) return fail(CRSDR_EINVAL, "plan_bind_packet: NULL plan");
    if (device_packet && (((uintptr_t)device_packet + p->matrix_off) % 4 || packet_stride % 4 ||
                          (packet_stride < p->packet_bytes && p->max_batch > 1)))
        return fail(CRSDR_EINVAL, "plan_bind_packet: matrix start and stride must be 4-byte aligned, stride >= packet bytes");
    p->d_packet = device_packet ? (int8_t *)device_packet : p->d_packet_own;
    p->packet_stride = device_packet ? packet_stride : p->own_packet_stride;
    return CRSDR_OK;
}

extern "C" int crsdr_plan_set_frac_apply(crsdr_plan *p, int enable, float gain, const float *frac_override)
{
    if (!p) return fail(CRSDR_EINVAL, "plan_set_frac_apply: NULL plan");
    if (enable < 0 || enable > 2) return fail(CRSDR_EINVAL, "plan_set_frac_apply: enable = %d (0 off, 1 on, 2 on without the second work area)", enable);
    if (!enable) {
        // off: the two buffers this call allocated go back (a second cf32 work area is 8 * blocksize bytes per owned row)
        if (p->d_Z || p->d_rowspec) {
            int rc = crsdr_plan_sync(p);
            if (rc) return rc;
            if (p->d_Z) (void)hipFree(p->d_Z);
            if (p->d_rowspec) (void)hipFree(p->d_rowspec);
            p->d_Z = nullptr; p->d_rowspec = nullptr;
        }
        p->frac_apply = false;
        return CRSDR_OK;
    }
    if (p->mode != CRSDR_MODE_DIGITAL)
        return fail(CRSDR_EINVAL, "plan_set_frac_apply: needs a plan in CRSDR_MODE_DIGITAL (the faithful mode never shifts samples)");
    // |D| is bounded: the response's phase k_s D / B is formed in fp32 (cis2pi), whose argument loses 6e-8 of its size per ulp --
    // at |D| <= 64 samples the phase stays within 1e-5 rad.  A proper peak's parabolic estimate is |frac| <= 1/2, so D = gain * frac
    // needs |gain| <= 128; caller-supplied delays beyond the bound belong in the integer lag (the resampler servo's business).
    if (!(gain == gain) || std::fabs(gain) > 128.0f) return fail(CRSDR_EINVAL, "plan_set_frac_apply: gain must be finite, |gain| <= 128");
    if (frac_override)
        for (int r = 1; r < p->nrows; ++r)
            if (!(std::fabs(frac_override[r]) <= 64.0f))
                return fail(CRSDR_EINVAL, "plan_set_frac_apply: frac_override[%d] = %g (need |D| <= 64 samples; larger delays belong in the integer lag)", r, (double)frac_override[r]);
    HIP_TRY(hipSetDevice(p->device));
    if (frac_override) {
        int rc = crsdr_plan_sync(p);
        if (rc) return rc;
        HIP_TRY(hipMemcpy(p->d_frac_override, frac_override, sizeof(float) * (size_t)p->nrows, hipMemcpyHostToDevice));
    }
    if (!p->longblock) {                   // LDS-resident blocks: one kernel behind the phase kernels, nothing to allocate
        p->frac_apply = true; p->frac_gain = gain; p->frac_override_on = frac_override != nullptr;
        return CRSDR_OK;
    }
    if (!p->d_rowspec) HIP_TRY(hipMalloc((void **)&p->d_rowspec, sizeof(float4) * 8192 * (size_t)p->row_count));
    if (enable == 2 && p->d_Z) {           // memory-lean form asked for: give the second work area back
        int rc = crsdr_plan_sync(p);
        if (rc) return rc;
        (void)hipFree(p->d_Z);
        p->d_Z = nullptr;
    }
    if (enable == 1 && !p->d_Z && hipMalloc((void **)&p->d_Z, sizeof(float2) * (size_t)p->B * (size_t)p->row_count) != hipSuccess) {
        p->d_Z = nullptr;                  // no room for a second work area: the apply pass repeats stage A instead (what enable = 2 asks for)
        (void)hipGetLastError();
    }
    p->frac_apply = true; p->frac_gain = gain; p->frac_override_on = frac_override != nullptr;
    return CRSDR_OK;
}

// slab mode with tails (crsdr_plan_bind_slab_ex): {lag, mag, frac, phasor} of the owned rows behind the rows of every slot
static int pack_tails(crsdr_plan *p, hipStream_t S, int nblocks, const int32_t *o_lag, const float *o_mag, const float *o_frac, const uint32_t *d_readcnt,
                      uint32_t seq)
{
    if (!p->d_slab || !p->tail_offset) return CRSDR_OK;
    hipLaunchKernelGGL(k_pack_tails, dim3((unsigned)((p->row_count + 255) / 256), (unsigned)nblocks), dim3(256), 0, S, p->d_slab, p->slab_stride, p->tail_offset,
                       p->row_begin, p->row_count, p->nrows, o_lag, o_mag, o_frac, p->d_phasor, d_readcnt, seq);
    HIP_TRY(hipGetLastError());
    return CRSDR_OK;
}

// Snapshot of the carried state before the batch now being submitted (at most one per batch; stream-ordered behind the previous
// batch's kernels).  The entry only counts once the copy has been accepted by the runtime.
static hipError_t take_snapshot(crsdr_plan *p, hipStream_t s)
{
    if (p->snap_cnt == crsdr_plan::kSnaps) return hipSuccess;                                   // full: the oldest stays the rollback point
    if (p->snap_cnt && p->snaps[(p->snap_head + p->snap_cnt - 1) % crsdr_plan::kSnaps].idx == p->submit_idx) return hipSuccess;
    const int pos = (p->snap_head + p->snap_cnt) % crsdr_plan::kSnaps;
    const hipError_t e = hipMemcpyAsync(p->d_state_snap + (size_t)pos * p->state_bytes, p->d_state, p->state_bytes, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return e;
    p->snaps[pos].idx = p->submit_idx;
    p->snaps[pos].phase_cur = p->phase_cur;
    p->snap_cnt++;
    return hipSuccess;
}
// The status word was read clean behind batch k: every batch <= k is good, so a snapshot is obsolete as soon as a newer one
// that is still a state before some batch <= k + 1 exists; the last one goes too if nothing was submitted after batch k.
static void retire_snapshots(crsdr_plan *p, unsigned long k)
{
    while (p->snap_cnt >= 2 && p->snaps[(p->snap_head + 1) % crsdr_plan::kSnaps].idx <= k + 1) { p->snap_head = (p->snap_head + 1) % crsdr_plan::kSnaps; p->snap_cnt--; }
    if (p->snap_cnt == 1 && p->snaps[p->snap_head].idx <= k && p->submit_idx == k + 1) p->snap_cnt = 0;
}

extern "C" int crsdr_plan_submit_batch(crsdr_plan *p, const void *rows, int mem_kind, int nblocks, size_t block_stride,
                                       const uint32_t *readcnt, const uint8_t *lag_mask, uint32_t seq, uint32_t flags)
{
    if (!p || !rows) return fail(CRSDR_EINVAL, "plan_submit: NULL plan or rows");
    if (mem_kind != CRSDR_MEM_HOST && mem_kind != CRSDR_MEM_DEVICE) return fail(CRSDR_EINVAL, "plan_submit: mem_kind = %d", mem_kind);
    if (nblocks < 1 || nblocks > p->max_batch) return fail(CRSDR_EINVAL, "plan_submit: nblocks = %d (plan max_batch = %d)", nblocks, p->max_batch);
    const size_t B = (size_t)p->B, n = (size_t)p->nrows, T = (size_t)nblocks;
    if (block_stride == 0) block_stride = n * B;
    if (block_stride < n * B) return fail(CRSDR_EINVAL, "plan_submit: block_stride smaller than one block");
    if (mem_kind == CRSDR_MEM_DEVICE && ((uintptr_t)rows % 4 || block_stride % 4))
        return fail(CRSDR_EINVAL, "plan_submit: device rows and block_stride must be 4-byte aligned (16-byte for the vector kernels)");
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t S = p->stream, A = p->aux;
    const uint32_t xor80 = (flags & CRSDR_OFFSET_BINARY) ? 0x80808080u : 0u;

    const int8_t *d_in = (const int8_t *)rows;
    size_t d_stride = block_stride;
    bool input_ready = (flags & CRSDR_INPUT_READY) && mem_kind == CRSDR_MEM_DEVICE;
    if (mem_kind == CRSDR_MEM_HOST) {
        // reference row + this plan's slab only (cbuffer hand-off, src/crtlsdr.cc:173-203)
        for (size_t t = 0; t < T; ++t) {
            const int8_t *src = (const int8_t *)rows + t * block_stride;
            int8_t *dst = p->d_rows + t * n * B;
            HIP_TRY(hipMemcpyAsync(dst, src, B, hipMemcpyHostToDevice, S));
            HIP_TRY(hipMemcpyAsync(dst + B * (size_t)p->row_begin, src + B * (size_t)p->row_begin, B * (size_t)p->row_count,
                                   hipMemcpyHostToDevice, S));
        }
        d_in = p->d_rows;
        d_stride = n * B;
    }

    // small per-batch host arrays through the pinned ring
    const uint32_t *d_readcnt = nullptr;
    const uint8_t *d_mask = nullptr;
    bool any_lag = !(flags & CRSDR_NO_LAG);
    if (readcnt || lag_mask) {
        const int ss = p->stage_slot;
        p->stage_slot = (ss + 1) % kStageSlots;
        if (p->stage_valid[ss]) HIP_TRY(hipEventSynchronize(p->ev_stage[ss]));
        if (readcnt) {
            uint32_t *h = p->h_readcnt + (size_t)ss * n * (size_t)p->max_batch;
            memcpy(h, readcnt, sizeof(uint32_t) * n * T);
            HIP_TRY(hipMemcpyAsync(p->d_readcnt, h, sizeof(uint32_t) * n * T, hipMemcpyHostToDevice, S));
            d_readcnt = p->d_readcnt;
        }
        if (lag_mask) {
            uint8_t *h = p->h_mask + (size_t)ss * n;
            memcpy(h, lag_mask, n);
            bool any = false;
            for (int r = p->row_begin; r < p->row_begin + p->row_count; ++r) any |= (h[r] != 0);
            any_lag = any_lag && any;
            HIP_TRY(hipMemcpyAsync(p->d_mask, h, n, hipMemcpyHostToDevice, S));
            d_mask = p->d_mask;
        }
        HIP_TRY(hipEventRecord(p->ev_stage[ss], S));
        p->stage_valid[ss] = true;
    }

    if (p->copy_pending) {
        // a pipelined fetch of the previous batch is (or was) in flight: this batch's kernels overwrite what it reads.  The
        // input copies above stay ahead of this wait, so they share the link with the device-to-host copies
        hipEvent_t newest = p->ev_copydone[(p->copy_tail - 1) % 4];
        HIP_TRY(hipStreamWaitEvent(S, newest, 0));
        p->copy_pending = false;
    }
    if (p->prof_slots) {
        const int ps = (int)(p->prof_count % p->prof_slots);
        for (int k = 0; k < kProfKernels; ++k) p->prof_has[(size_t)ps * kProfKernels + k] = 0;
        if (p->prof_mask & (1u << 31)) HIP_TRY(hipEventRecord(p->ev_start, S));
    }

    const int pin = p->phase_cur, pout = pin ^ 1;
    const int obuf = (p->obuf ^= 1);                      // this batch's {lag, mag, frac}
    const size_t ooff = (size_t)obuf * n * (size_t)p->max_batch;
    int32_t *o_lag = p->d_lag + ooff;
    float *o_mag = p->d_mag + ooff, *o_frac = p->d_frac + ooff;
    AlignArgs aa;
    aa.rows = d_in; aa.block_stride = d_stride; aa.packet = p->d_packet; aa.packet_stride = p->packet_stride;
    aa.readcnt = d_readcnt; aa.lag_mask = d_mask; aa.lag = o_lag; aa.lag_state = p->d_lag_state; aa.corr = p->d_corr;
    aa.phase_in = p->d_phase_state[pin]; aa.phase_out = p->d_phase_state[pout]; aa.phasor = p->d_phasor;
    aa.nrows = p->nrows; aa.B = p->B; aa.row_begin = p->row_begin; aa.nblocks = nblocks;
    aa.digital = (p->mode == CRSDR_MODE_DIGITAL); aa.refnoise = (flags & CRSDR_REFNOISE_ENABLED) ? 1 : 0;
    aa.xcorr_ran = any_lag ? 1 : 0;
    aa.inline_chain = 0;
    // rows in and packet rows out are touched once: non-temporal (measured in the locked cadence, A/B in one call: stores +2.3 %, loads neutral)
    aa.nt = 3;
    aa.seq = seq; aa.xor80 = xor80;
    aa.slab = p->d_slab; aa.slab_stride = p->slab_stride; aa.hdr_first = p->hdr_first; aa.hdr_count = p->hdr_count;
    aa.lag_out = o_lag; aa.mag_out = o_mag; aa.frac_out = o_frac; aa.mag_state = p->d_mag_state; aa.frac_state = p->d_frac_state;

    // 16-byte accesses need 16-byte aligned rows and matrix rows (always true for the plan's own
    // buffers; a caller-bound packet or device input may only be 4-byte aligned -> word kernels)
    const bool vec16 = ((uintptr_t)d_in % 16 == 0) && (d_stride % 16 == 0) && (((uintptr_t)p->d_packet + p->matrix_off) % 16 == 0) &&
                       (p->packet_stride % 16 == 0 || nblocks == 1) &&
                       (!p->d_slab || ((uintptr_t)p->d_slab % 16 == 0 && p->slab_stride % 16 == 0));
    const int chunks = p->B > 16384 ? p->B / 16384 : 1; // long rows: 16 KiB of the row per workgroup
    const bool fused = p->fused_k2 && vec16 && chunks == 1;
    // (K1 of batch b + 1 on its own stream beside the phase kernel of batch b was built and measured in r01 / r02 -- dead ends 5 and 10
    // of DESIGN.md: the kernels do overlap, and K1 stretches by exactly what the phase kernel costs alone -- and removed in r03.)
    bool corr_zeroed = false;
    bool kept_fwd = false;                        // long blocks: stage A's output survived the correlation pass (in d_Y; stage B wrote d_Z)
    hipStream_t KS = S;

    if (any_lag && p->longblock) {
        // B = N1 x 16384: column FFTs -> row FFTs (x conj ref, inverse) -> inverse column FFTs + argmax -> finalize
        lb::LongTw tw{p->d_wc, p->d_wf, p->d_tw1, p->d_tw, (uint32_t)(p->B - 1)};
        const int n1 = 1 << p->log2n1, l1 = p->log2n1;
        // The reference row's two stages (one row: 256 tiles, 128 lines -- they cannot fill the device) run on the aux stream
        // into the other spectrum slot, like K0 of the short blocks: with resident input they sit under the previous block's
        // stage C and phase kernels instead of in front of this block's stage A (21 us of 521 per cfg5 block).  The work line
        // d_Yref is only touched on the aux stream; a slot is rewritten after the last stage B that read it (ev_k1done).
        const int slot = (p->slot ^= 1);
        const bool forked = !input_ready || d_readcnt || d_mask;
        if (forked) {
            HIP_TRY(hipEventRecord(p->ev_fork, S));
            if (!input_ready) HIP_TRY(hipStreamWaitEvent(A, p->ev_fork, 0));
        }
        if (p->k1done_valid[slot]) HIP_TRY(hipStreamWaitEvent(A, p->ev_k1done[slot], 0));
        hipEvent_t *pe0 = prof_pair(p, CRSDR_KERNEL_REF_SPECTRUM);
        if (pe0) HIP_TRY(hipEventRecord(pe0[0], A));
        HIP_TRY(CRSDR_DISPATCH_N1(l1, (launch_long_fwd_cols<LG, true>(A, 1, d_in, 0, xor80, tw, p->d_Yref))));
        HIP_TRY(launch_long_rows<true>(A, n1, 1, p->d_Yref, p->d_twA, p->d_twB, p->d_refspec[slot]));
        if (pe0) HIP_TRY(hipEventRecord(pe0[1], A));
        HIP_TRY(hipEventRecord(p->ev_ref[slot], A));
        HIP_TRY(hipStreamWaitEvent(S, p->ev_ref[slot], 0));
        XcorrArgs xa;
        xa.rows = d_in; xa.block_stride = d_stride; xa.refspec = p->d_refspec[slot]; xa.lag_mask = d_mask;
        xa.row_begin = p->row_begin; xa.nrows = p->nrows; xa.nblocks = 1; xa.xor80 = xor80; xa.stagger = 0;
        xa.lag = o_lag; xa.mag = o_mag; xa.frac = o_frac;
        xa.lag_state = p->d_lag_state; xa.mag_state = p->d_mag_state; xa.frac_state = p->d_frac_state;
        hipEvent_t *pe1 = prof_pair(p, CRSDR_KERNEL_XCORR_LAG);
        if (pe1) HIP_TRY(hipEventRecord(pe1[0], S));
        // with the fractional-delay pass to follow, stage B writes into the second work area: stage A's output stays in d_Y and
        // the apply pass does not repeat it (73 us of a 770 us cfg5 block) -- for the whole block or not at all (two-line kernel only)
        {
            static const int longq0 = [] { const char *e = getenv("CRSDR_LONG_Q"); return e ? atoi(e) : 1; }();
            const char kv0 = k1_variant();
            kept_fwd = p->d_Z && p->frac_apply && aa.digital && longq0 && (kv0 == 'a' || kv0 == 'q') && !p->q_disabled;
            for (int r0 = 0; r0 < p->row_count; r0 += p->long_chunk)
                kept_fwd = kept_fwd && (long)n1 * std::min(p->long_chunk, p->row_count - r0) >= 1024;
        }
        // rows go through the three stages in chunks whose cf32 work area fits the 256 MB memory-side cache
        for (int r0 = 0; r0 < p->row_count; r0 += p->long_chunk) {
            const int cnt = std::min(p->long_chunk, p->row_count - r0);
            float2 *Yc = p->d_Y + (size_t)r0 * (size_t)p->B;
            HIP_TRY(CRSDR_DISPATCH_N1(l1, (launch_long_fwd_cols<LG, false>(S, cnt, d_in, p->row_begin + r0, xor80, tw, Yc))));
            // stage B: two lines per CU (k_rows14_cf32q) once a launch has enough lines per CU; the packed one-line kernel otherwise
            // (and always under CRSDR_K1_VARIANT=packed|scalar, or after a bounded wait of the two-line kernels ran out)
            static const int longq = [] { const char *e = getenv("CRSDR_LONG_Q"); return e ? atoi(e) : 1; }();
            const char kv = k1_variant();
            const bool useq = longq && (kv == 'a' || kv == 'q') && !p->q_disabled && (long)n1 * cnt >= 1024;
            float2 *Zc = kept_fwd ? p->d_Z + (size_t)r0 * (size_t)p->B : Yc;
            if (useq) {
                HIP_TRY(take_snapshot(p, S));              // same rollback protocol as the two-row K1 (check_fused_status)
                HIP_TRY(launch_long_rows_q(S, n1, cnt, Yc, p->d_twA, p->d_twB, p->d_refspec[slot], reinterpret_cast<int *>(p->d_sync + 2), p->d_sync + 3, &p->q_work_base,
                                           nullptr, Zc != Yc ? Zc : nullptr));
                p->k1_used = true;
            } else
                HIP_TRY(launch_long_rows<false>(S, n1, cnt, Yc, p->d_twA, p->d_twB, p->d_refspec[slot]));
            HIP_TRY(CRSDR_DISPATCH_N1(l1, (launch_long_inv_cols<LG>(S, cnt, Zc, tw, p->d_part + (size_t)r0 * lb::ntiles(l1)))));
        }
        HIP_TRY(hipEventRecord(p->ev_k1done[slot], S));       // the last reader of this block's reference spectrum has been enqueued
        p->k1done_valid[slot] = true;
        corr_zeroed = aa.refnoise && !fused && chunks > 1 && nblocks == 1;      // (the three-kernel path's integer sums: see below)
        hipLaunchKernelGGL(lb::k_long_finalize, dim3(p->row_count), dim3(256), 0, S, kept_fwd ? p->d_Z : p->d_Y, p->d_part, tw, n1, lb::ntiles(l1), xa, corr_zeroed ? p->d_corr : nullptr);
        HIP_TRY(hipGetLastError());
        if (pe1) HIP_TRY(hipEventRecord(pe1[1], S));
    } else if (any_lag) {
        const int slot = (p->slot ^= 1);
        // B = 16384: the reference spectra are the first work items of the cross-correlation launch itself (fold) -- one launch, no
        // second stream, no event between the launches of a batch.  r03, per-rank shape of the 8-GPU run (128 rows x 20 blocks,
        // 0.16 ms per batch): the kernel on the aux stream could only start when the previous batch's K1 let go of its CUs, and the
        // cross-stream wait + the event record between K1 and the phase kernel left 11 + 11 us of gaps around 138 us of kernels.
        // Folded where the launch is small (the packed kernel's launches: a GPU's share of an 8-GPU run, 0.16 ms per batch): K1 itself
        // grows by the reference items and its first rows' wait for them (0.118 -> 0.129 ms per 2560-row launch), the gaps around it
        // shrink by more (fenced 110 k -> 117 k blocks/s, back to back 122.9 k -> 124.6 k) and the host issues a batch in 10 us
        // instead of 26 - 40.  The two-row kernel's large launches keep the aux-stream kernel: there it hides completely, and the
        // folded form measured 0.910 -> 0.932 ms per 20-block launch of 1024 rows (every group's first row waits for block 0's item).
        const bool fold = p->fold && p->log2n == 14 && !p->q_disabled && k1_pick(p->row_count * nblocks, !p->q_disabled) == 'p';
        if (!fold) {
            // K0 on the aux stream: with resident input it overlaps the previous batch's K1 / K2.  (Putting it on the main stream
            // when both streams are idle -- a cold batch -- was measured in r02: exposed K0 0.0526 -> 0.0516 ms per cold batch and
            // no difference beyond the spread in a 20-block timed region; not kept.)
            const bool forked = !input_ready || d_readcnt || d_mask;
            if (forked) {
                HIP_TRY(hipEventRecord(p->ev_fork, S)); // input copies, mask copy / the caller's producer work
                if (!input_ready) HIP_TRY(hipStreamWaitEvent(A, p->ev_fork, 0));
            }
            if (p->k1done_valid[slot]) HIP_TRY(hipStreamWaitEvent(A, p->ev_k1done[slot], 0)); // refspec[slot] free again
            hipEvent_t *pe0 = prof_pair(p, CRSDR_KERNEL_REF_SPECTRUM);
            if (pe0) HIP_TRY(hipEventRecord(pe0[0], A));
            if (p->log2n == 14) HIP_TRY(launch_ref_spectrum14(A, nblocks, d_in, d_stride, p->d_twA, p->d_twB, p->d_refspec[slot], xor80));
            else HIP_TRY(CRSDR_DISPATCH_LOG2(p->log2n, (launch_ref_spectrum<LG>(A, nblocks, d_in, d_stride, p->d_tw, p->d_refspec[slot], xor80))));
            if (pe0) HIP_TRY(hipEventRecord(pe0[1], A));
            HIP_TRY(hipEventRecord(p->ev_ref[slot], A));
        }

        XcorrArgs xa;
        xa.rows = d_in; xa.block_stride = d_stride; xa.refspec = p->d_refspec[slot]; xa.lag_mask = d_mask;
        xa.row_begin = p->row_begin; xa.nrows = p->nrows; xa.nblocks = nblocks; xa.xor80 = xor80;
        xa.stagger = 0;       // (x 512 cycles of head start for half the waves: measured r01, 0 is best for the packed kernels)
        xa.lag = o_lag; xa.mag = o_mag; xa.frac = o_frac;
        xa.lag_state = p->d_lag_state; xa.mag_state = p->d_mag_state; xa.frac_state = p->d_frac_state;
        if (fold) {
            xa.fold = 1; xa.refspec_w = p->d_refspec[slot]; xa.refflag = p->d_refflag; xa.refgen = ++p->refgen;
            xa.errflag = reinterpret_cast<int *>(p->d_sync + 2);
        } else
            HIP_TRY(hipStreamWaitEvent(KS, p->ev_ref[slot], 0));
        hipEvent_t *pe1 = prof_pair(p, CRSDR_KERNEL_XCORR_LAG);
        if (pe1) HIP_TRY(hipEventRecord(pe1[0], KS));
        if (p->log2n == 14) {
            bool q = false;
            // every launch with bounded waits snapshots the carried state first (one 28 B/row copy, ordered after the previous
            // batch's kernels on this stream) while the ring has room: what a wait that ran out is rolled back to
            auto snapshot = [p, KS]() -> hipError_t { return take_snapshot(p, KS); };
            HIP_TRY(launch_xcorr_lag14(KS, xa, p->row_count, p->d_twA, p->d_twB, reinterpret_cast<int *>(p->d_sync + 2), &q, p->d_sync + 3, &p->q_work_base,
                                       !p->q_disabled, snapshot));
            p->k1_used |= q || fold;
        }
        else HIP_TRY(CRSDR_DISPATCH_LOG2(p->log2n, (launch_xcorr_lag<LG>(KS, xa, p->row_count, p->d_tw))));
        if (pe1) HIP_TRY(hipEventRecord(pe1[1], KS));
        if (!fold) {                                     // (a folded launch's spectra are written and read on this stream only)
            HIP_TRY(hipEventRecord(p->ev_k1done[slot], KS));
            p->k1done_valid[slot] = true;
        }
    }
    if (fused) {
        // fused phase path: every row is read once (k_align_fused); timed under CRSDR_KERNEL_ALIGN_QUANT
        static const int spin_limit = [] { const char *e = getenv("CRSDR_K2_SPIN"); return e ? atoi(e) : kFusedSpinLimit; }();
        // hand-over words: two slots per (row, block).  A batch publishes into one and its workgroups re-arm the other
        // (all-ones) for the next batch, so the steady state needs no memset between launches; the host only tracks how
        // many leading blocks of each slot are armed and falls back to a memset when a batch needs more than that
        const int cs = p->chain_slot;
        unsigned long long *chain = reinterpret_cast<unsigned long long *>(p->d_corr);
        unsigned long long *chainv = chain + 2 * n * (size_t)p->max_batch;      // second half of d_corr: the chain values, same two slots
        FusedSync fs{p->d_sync, p->d_sync + 1, chain + cs, aa.refnoise ? chain + (cs ^ 1) : nullptr, chainv + cs, aa.refnoise ? chainv + (cs ^ 1) : nullptr,
                     p->row_count, spin_limit};
        hipEvent_t *pe = prof_pair(p, CRSDR_KERNEL_ALIGN_QUANT);
        if (pe) HIP_TRY(hipEventRecord(pe[0], S));
        if (aa.refnoise) {
            if (nblocks > 1 && p->chain_armed[cs] < nblocks) {
                HIP_TRY(hipMemsetAsync(p->d_corr, 0xFF, sizeof(long long) * 4 * n * T, S));
                p->chain_armed[0] = p->chain_armed[1] = (int)T;
            }
            p->chain_armed[cs] = 0;                                             // published into
            p->chain_armed[cs ^ 1] = std::max(p->chain_armed[cs ^ 1], nblocks); // re-armed by this launch
            p->chain_slot = cs ^ 1;
        }
        {
            // (two rows per workgroup sharing the reference row's registers: built and measured in r03, 4.05 against 4.85 TB/s -- tools/k2_pair.hpp;
            //  persistent workgroups with the next item's row loads in flight: 3.1 - 3.6 against 4.76 TB/s -- tools/k2_persist.hpp)
            const dim3 grid((unsigned)((1 + p->row_count) * nblocks));
            if (p->B == 16384 && !xor80) hipLaunchKernelGGL((k_align_fused<true, false>), grid, dim3(kAlignThreads), 0, S, aa, fs);
            else if (p->B == 16384) hipLaunchKernelGGL((k_align_fused<true, true>), grid, dim3(kAlignThreads), 0, S, aa, fs);
            else hipLaunchKernelGGL((k_align_fused<false, true>), grid, dim3(kAlignThreads), 0, S, aa, fs);
        }
        HIP_TRY(hipGetLastError());
        if (p->frac_apply && !p->longblock && aa.digital) {
            // fractional-delay correction of LDS-resident blocks: the owned rows once more, through the frequency domain
            FracArgs fa{d_in, d_stride, p->d_packet, p->packet_stride, p->d_slab, p->slab_stride, p->nrows, p->row_begin, xor80, o_lag, o_frac,
                        p->frac_override_on ? p->d_frac_override : nullptr, p->frac_gain, p->d_phasor};
            HIP_TRY(CRSDR_DISPATCH_LOG2(p->log2n, (launch_frac_apply<LG>(S, p->row_count, nblocks, fa, p->d_tw, p->d_twA, p->d_twB, p->frac_generic ? nullptr : p->d_k2tab))));
        }
        if (pe) HIP_TRY(hipEventRecord(pe[1], S));
        { int rc_ = pack_tails(p, S, nblocks, o_lag, o_mag, o_frac, d_readcnt, seq); if (rc_) return rc_; }
        HIP_TRY(hipEventRecord(p->ev_k2done[obuf], S));
        p->k2done_valid[obuf] = true;
        p->fused_used = true;
        p->phase_cur = pout;
        p->last_nblocks = nblocks;
        if (p->prof_slots) { if (p->prof_mask & (1u << 31)) HIP_TRY(hipEventRecord(p->ev_stop, S)); p->prof_count++; }
        p->submitted = true;
        p->submit_idx++;
        return CRSDR_OK;
    }
    if (aa.refnoise) {
        hipEvent_t *pe = prof_pair(p, CRSDR_KERNEL_PHASE_DOT);
        if (pe) HIP_TRY(hipEventRecord(pe[0], S));
        p->chain_armed[0] = p->chain_armed[1] = 0;   // this path keeps its integer sums in d_corr
        if (chunks > 1 && !corr_zeroed) HIP_TRY(hipMemsetAsync(p->d_corr, 0, sizeof(long long) * 2 * n * T, S)); // atomically accumulated (k_long_finalize has zeroed a tracked block's)
        // long rows: 32 KiB per workgroup with all 16 loads of a thread in flight at once (cfg5: 34 -> 30 us per block by the plan's events)
        if (vec16 && chunks >= 2) hipLaunchKernelGGL((k_phase_dot<true, 8>), dim3(p->row_count, nblocks, chunks / 2), dim3(kAlignThreads), 0, S, aa);
        else if (vec16) hipLaunchKernelGGL((k_phase_dot<true, 4>), dim3(p->row_count, nblocks, chunks), dim3(kAlignThreads), 0, S, aa);
        else hipLaunchKernelGGL((k_phase_dot<false, 4>), dim3(p->row_count, nblocks, chunks), dim3(kAlignThreads), 0, S, aa);
        HIP_TRY(hipGetLastError());
        if (pe) HIP_TRY(hipEventRecord(pe[1], S));
    }
    const bool apply = p->frac_apply && p->longblock && aa.digital;
    // long rows, one block per submit, rows rotated by k_align_quant itself: the single chain step is folded there
    aa.inline_chain = (chunks > 1 && nblocks == 1 && !apply) ? 1 : 0;
    if (!aa.inline_chain) {
        hipLaunchKernelGGL(k_phase_chain, dim3((p->row_count + 63) / 64), dim3(64), 0, S, aa, p->row_count);
        HIP_TRY(hipGetLastError());
    }
    {
        hipEvent_t *pe = prof_pair(p, CRSDR_KERNEL_ALIGN_QUANT);
        if (pe) HIP_TRY(hipEventRecord(pe[0], S));
        // with the fractional-delay correction on, the rows come from the apply pass below: here only header + readcnt + row 0
        const unsigned gx = apply ? 1u : 1u + (unsigned)p->row_count;
        if (vec16) hipLaunchKernelGGL(k_align_quant<true>, dim3(gx, nblocks, chunks), dim3(kAlignThreads), 0, S, aa);
        else hipLaunchKernelGGL(k_align_quant<false>, dim3(gx, nblocks, chunks), dim3(kAlignThreads), 0, S, aa);
        HIP_TRY(hipGetLastError());
        if (apply) {
            // apply pass (crsdr_plan_set_frac_apply): int8 -> column FFTs -> row FFTs x H_row -> inverse -> inverse column FFTs -> int8
            lb::LongTw tw{p->d_wc, p->d_wf, p->d_tw1, p->d_tw, (uint32_t)(p->B - 1)};
            const int n1 = 1 << p->log2n1, l1 = p->log2n1;
            int8_t *obase = p->d_slab ? p->d_slab : p->d_packet + p->matrix_off + (size_t)p->row_begin * (size_t)p->B;
            for (int r0 = 0; r0 < p->row_count; r0 += p->long_chunk) {
                const int cnt = std::min(p->long_chunk, p->row_count - r0);
                float2 *Yc = p->d_Y + (size_t)r0 * (size_t)p->B;
                x14p::RampArgs ra{o_lag, o_frac, p->frac_override_on ? p->d_frac_override : nullptr, p->d_phasor, p->d_k2tab, p->d_wc, p->d_wf, lb::FBITS, p->frac_gain, p->row_begin + r0, l1};
                if (!kept_fwd) HIP_TRY(CRSDR_DISPATCH_N1(l1, (launch_long_fwd_cols<LG, false>(S, cnt, d_in, p->row_begin + r0, xor80, tw, Yc))));
                // stage B'': two lines per CU with the rows' responses as per-row spectra (k_ramp_rowspec + k_rows14_cf32q<true>) once a
                // launch has enough lines; the one-line kernel that forms the response per bin otherwise
                static const int longq2 = [] { const char *e = getenv("CRSDR_LONG_Q"); return e ? atoi(e) : 1; }();
                const char kv2 = k1_variant();
                if (longq2 && (kv2 == 'a' || kv2 == 'q') && !p->q_disabled && (long)n1 * cnt >= 1024) {
                    hipLaunchKernelGGL(x14p::k_ramp_rowspec, dim3(8192 / 256, cnt), dim3(256), 0, S, p->d_rowspec + (size_t)r0 * 8192, ra);
                    HIP_TRY(hipGetLastError());
                    HIP_TRY(take_snapshot(p, S));          // (the same rollback protocol as the correlation pass)
                    HIP_TRY(launch_long_rows_q(S, n1, cnt, Yc, p->d_twA, p->d_twB, reinterpret_cast<float2 *>(p->d_rowspec + (size_t)r0 * 8192),
                                               reinterpret_cast<int *>(p->d_sync + 2), p->d_sync + 3, &p->q_work_base, &ra));
                    p->k1_used = true;
                } else
                    HIP_TRY(launch_long_rows_ramp(S, n1, cnt, Yc, p->d_twA, p->d_twB, ra));
                HIP_TRY(CRSDR_DISPATCH_N1(l1, (launch_long_out_cols<LG>(S, cnt, Yc, tw, obase + (size_t)r0 * (size_t)p->B))));
            }
        } else if (p->frac_apply && !p->longblock && aa.digital) {       // LDS-resident blocks on the three-kernel path: as behind the fused kernel
            FracArgs fa{d_in, d_stride, p->d_packet, p->packet_stride, p->d_slab, p->slab_stride, p->nrows, p->row_begin, xor80, o_lag, o_frac,
                        p->frac_override_on ? p->d_frac_override : nullptr, p->frac_gain, p->d_phasor};
            HIP_TRY(CRSDR_DISPATCH_LOG2(p->log2n, (launch_frac_apply<LG>(S, p->row_count, nblocks, fa, p->d_tw, p->d_twA, p->d_twB, p->frac_generic ? nullptr : p->d_k2tab))));
        }
        if (pe) HIP_TRY(hipEventRecord(pe[1], S));
    }
    { int rc_ = pack_tails(p, S, nblocks, o_lag, o_mag, o_frac, d_readcnt, seq); if (rc_) return rc_; }
    HIP_TRY(hipEventRecord(p->ev_k2done[obuf], S));
    p->k2done_valid[obuf] = true;
    p->phase_cur = pout;
    p->last_nblocks = nblocks;
    if (p->prof_slots) { if (p->prof_mask & (1u << 31)) HIP_TRY(hipEventRecord(p->ev_stop, S)); p->prof_count++; }
    p->submitted = true;
    p->submit_idx++;
    return CRSDR_OK;
}

extern "C" int crsdr_plan_submit(crsdr_plan *p, const void *rows, int mem_kind, const uint32_t *readcnt,
                                 const uint8_t *lag_mask, uint32_t seq, uint32_t flags)
{
    return crsdr_plan_submit_batch(p, rows, mem_kind, 1, 0, readcnt, lag_mask, seq, flags);
}

// (the fused phase kernel counts look-back waits it gave up on and computed locally -- d_sync[1] -- : nothing to report, results
// are exact either way)
static int check_fused_status(crsdr_plan *p)
{
    // the two-row K1 (xcorr14q.hpp) bounds its waits for the LDS image and its group barriers; one that runs out means a
    // scheduling assumption failed and the rows of that launch are not to be trusted: an error, never silent
    if (p->k1_used) {
        int w = 0;
        HIP_TRY(hipMemcpy(&w, p->d_sync + 2, sizeof(w), hipMemcpyDeviceToHost));
        if (w) {
            // Roll back: every batch submitted since the last clean check may have folded garbage lags into the carried
            // lag / mag / frac and EMA phase state.  Restore the snapshot taken before the first of them, re-arm the
            // phase kernel's hand-over words, and keep this plan on the packed kernel from here on -- the caller
            // resubmits those batches and gets what an undisturbed run would have given.
            (void)hipStreamSynchronize(p->stream); (void)hipStreamSynchronize(p->aux); (void)hipStreamSynchronize(p->cs);
            (void)hipMemset(p->d_sync, 0, 64);                    // flags and the work counter (its count is off after a failed launch)
            p->q_work_base = 0;
            unsigned long lost = 0;
            if (p->snap_cnt) {
                const crsdr_plan::Snap &sn = p->snaps[p->snap_head];      // the OLDEST unretired snapshot: the state before batch sn.idx
                (void)hipMemcpy(p->d_state, p->d_state_snap + (size_t)p->snap_head * p->state_bytes, p->state_bytes, hipMemcpyDeviceToDevice);
                p->phase_cur = sn.phase_cur;
                lost = p->submit_idx - sn.idx;
                p->submit_idx = sn.idx;
            }
            (void)hipMemset(p->d_corr, 0xFF, sizeof(long long) * 4 * (size_t)p->nrows * (size_t)p->max_batch);
            p->chain_armed[0] = p->chain_armed[1] = p->max_batch;
            p->snap_cnt = 0; p->snap_head = 0; p->k1_used = false; p->q_disabled = true; p->submitted = false;
            p->copy_pending = false; p->copy_head = p->copy_tail;       // outstanding asynchronous fetches delivered untrusted data: dropped
            return fail(CRSDR_EHIP, "xcorr: %d workgroup(s) of the two-row kernel ran out of a bounded wait; the last %lu submitted batch(es) -- everything "
                                    "since the last batch whose status was read clean -- were rolled back (carried lag and phase state restored): resubmit "
                                    "them; this plan now uses the packed kernel (CRSDR_K1_VARIANT=packed selects it from the start)", w, lost);
        }
        p->snap_cnt = 0; p->snap_head = 0; p->k1_used = false;      // clean and the streams are drained: the next two-row launch takes a fresh snapshot
    }
    return CRSDR_OK;
}

extern "C" int crsdr_plan_fetch_block(crsdr_plan *p, int block, int32_t *lag, float *mag, float *frac, float *phasor,
                                      int8_t *packet)
{
    if (!p) return fail(CRSDR_EINVAL, "plan_fetch: NULL plan");
    if (!p->submitted) return fail(CRSDR_ESTATE, "plan_fetch: nothing submitted");
    if (block < 0) block = p->last_nblocks - 1;
    if (block >= p->last_nblocks) return fail(CRSDR_EINVAL, "plan_fetch: block %d of a batch of %d", block, p->last_nblocks);
    if (packet && p->d_slab) return fail(CRSDR_ESTATE, "plan_fetch: slab output is bound, packets are assembled by the caller (crsdr_assemble_slabs)");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    { int rc_ = check_fused_status(p); if (rc_) return rc_; }
    const size_t n = (size_t)p->nrows, o = (size_t)block * n;
    const size_t ob = (size_t)p->obuf * n * (size_t)p->max_batch + o;
    if (lag) HIP_TRY(hipMemcpy(lag, p->d_lag + ob, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    if (mag) HIP_TRY(hipMemcpy(mag, p->d_mag + ob, sizeof(float) * n, hipMemcpyDeviceToHost));
    if (frac) HIP_TRY(hipMemcpy(frac, p->d_frac + ob, sizeof(float) * n, hipMemcpyDeviceToHost));
    if (phasor) HIP_TRY(hipMemcpy(phasor, p->d_phasor + o, sizeof(float2) * n, hipMemcpyDeviceToHost));
    if (packet) HIP_TRY(hipMemcpy(packet, p->d_packet + (size_t)block * p->packet_stride, p->packet_bytes, hipMemcpyDeviceToHost));
    return CRSDR_OK;
}

extern "C" int crsdr_plan_fetch_batch_async(crsdr_plan *p, int32_t *lag, float *mag, float *frac, float *phasor, int8_t *packets,
                                            size_t host_packet_stride)
{
    if (!p) return fail(CRSDR_EINVAL, "plan_fetch_batch_async: NULL plan");
    if (!p->submitted) return fail(CRSDR_ESTATE, "plan_fetch_batch_async: nothing submitted");
    if (packets && p->d_slab) return fail(CRSDR_ESTATE, "plan_fetch_batch_async: slab output is bound, packets are assembled by the caller");
    if (packets && host_packet_stride < p->packet_bytes) return fail(CRSDR_EINVAL, "plan_fetch_batch_async: host_packet_stride smaller than a packet");
    if (p->copy_tail - p->copy_head >= 4) return fail(CRSDR_ESTATE, "plan_fetch_batch_async: four fetches outstanding, call crsdr_plan_fetch_wait first");
    HIP_TRY(hipSetDevice(p->device));
    const size_t n = (size_t)p->nrows, T = (size_t)p->last_nblocks;
    const size_t ob = (size_t)p->obuf * n * (size_t)p->max_batch;
    HIP_TRY(hipStreamWaitEvent(p->cs, p->ev_k2done[p->obuf], 0));          // the batch's last kernel
    if (lag) HIP_TRY(hipMemcpyAsync(lag, p->d_lag + ob, sizeof(int32_t) * n * T, hipMemcpyDeviceToHost, p->cs));
    if (mag) HIP_TRY(hipMemcpyAsync(mag, p->d_mag + ob, sizeof(float) * n * T, hipMemcpyDeviceToHost, p->cs));
    if (frac) HIP_TRY(hipMemcpyAsync(frac, p->d_frac + ob, sizeof(float) * n * T, hipMemcpyDeviceToHost, p->cs));
    if (phasor) HIP_TRY(hipMemcpyAsync(phasor, p->d_phasor, sizeof(float2) * n * T, hipMemcpyDeviceToHost, p->cs));
    if (packets) {
        if (host_packet_stride == p->packet_stride)
            HIP_TRY(hipMemcpyAsync(packets, p->d_packet, p->packet_stride * (T - 1) + p->packet_bytes, hipMemcpyDeviceToHost, p->cs));
        else
            for (size_t t = 0; t < T; ++t)
                HIP_TRY(hipMemcpyAsync(packets + t * host_packet_stride, p->d_packet + t * p->packet_stride, p->packet_bytes, hipMemcpyDeviceToHost, p->cs));
    }
    // the two-row K1's error word rides along (a synchronous read in crsdr_plan_fetch_wait would wait for the whole device --
    // the next batch's upload included -- and serialise the two directions of the link: measured 25 + 25 instead of 45 GB/s)
    HIP_TRY(hipMemcpyAsync(&p->h_k1flag[p->copy_tail % 4], p->d_sync + 2, sizeof(int), hipMemcpyDeviceToHost, p->cs));
    HIP_TRY(hipEventRecord(p->ev_copydone[p->copy_tail % 4], p->cs));
    p->copy_idx[p->copy_tail % 4] = p->submit_idx - 1;                     // the batch these copies (and this status word) sit behind
    p->copy_tail++;
    p->copy_pending = true;
    return CRSDR_OK;
}

extern "C" int crsdr_plan_fetch_wait(crsdr_plan *p)
{
    if (!p) return fail(CRSDR_EINVAL, "plan_fetch_wait: NULL plan");
    HIP_TRY(hipSetDevice(p->device));
    if (p->copy_head == p->copy_tail) return fail(CRSDR_ESTATE, "plan_fetch_wait: no asynchronous fetch outstanding");
    HIP_TRY(hipEventSynchronize(p->ev_copydone[p->copy_head % 4]));      // the OLDEST outstanding fetch: later ones keep flying
    const int w = p->h_k1flag[p->copy_head % 4];                          // copied behind that batch's kernels: final for it
    const unsigned long k = p->copy_idx[p->copy_head % 4];
    p->copy_head++;
    if (w) return crsdr_plan_sync(p);                                      // drains everything, rolls back to the oldest unretired snapshot, reports
    retire_snapshots(p, k);                                                // batches <= k are good: the rollback point moves up behind them
    return CRSDR_OK;
}

extern "C" int crsdr_plan_fetch(crsdr_plan *p, int32_t *lag, float *mag, float *frac, float *phasor, int8_t *packet)
{
    return crsdr_plan_fetch_block(p, -1, lag, mag, frac, phasor, packet);
}

extern "C" int crsdr_plan_last_elapsed_ms(crsdr_plan *p, float *ms)
{
    if (!p || !ms) return fail(CRSDR_EINVAL, "plan_last_elapsed_ms: NULL argument");
    if (!p->submitted || !p->prof_slots || !(p->prof_mask & (1u << 31)))
        return fail(CRSDR_ESTATE, "plan_last_elapsed_ms: enable profiling with CRSDR_PROFILE_SUBMIT and submit first");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipEventSynchronize(p->ev_stop));
    HIP_TRY(hipEventElapsedTime(ms, p->ev_start, p->ev_stop));
    return CRSDR_OK;
}

extern "C" int crsdr_plan_enable_profiling(crsdr_plan *p, int slots, uint32_t kernel_mask)
{
    if (!p || slots < 0 || slots > 4096) return fail(CRSDR_EINVAL, "plan_enable_profiling: bad argument");
    int rc = crsdr_plan_sync(p);
    if (rc) return rc;
    for (hipEvent_t e : p->prof_ev) if (e) (void)hipEventDestroy(e);
    p->prof_ev.clear(); p->prof_has.clear();
    p->prof_slots = 0; p->prof_count = 0; p->prof_mask = kernel_mask;
    if (slots == 0) return CRSDR_OK;
    p->prof_ev.assign((size_t)slots * kProfKernels * 2, nullptr);
    p->prof_has.assign((size_t)slots * kProfKernels, 0);
    for (auto &e : p->prof_ev) HIP_TRY(hipEventCreate(&e));
    p->prof_slots = slots;
    return CRSDR_OK;
}

extern "C" int crsdr_plan_kernel_times(crsdr_plan *p, int which, float *ms, int capacity, int *count)
{
    if (!p || !ms || !count || which < 0 || which >= kProfKernels || capacity < 0) return fail(CRSDR_EINVAL, "plan_kernel_times: bad argument");
    *count = 0;
    if (!p->prof_slots) return fail(CRSDR_ESTATE, "plan_kernel_times: profiling not enabled");
    int rc = crsdr_plan_sync(p);
    if (rc) return rc;
    const long nrec = p->prof_count < p->prof_slots ? p->prof_count : p->prof_slots;
    const long first = p->prof_count - nrec;
    for (long i = first; i < p->prof_count && *count < capacity; ++i) {
        const int slot = (int)(i % p->prof_slots);
        if (!p->prof_has[(size_t)slot * kProfKernels + which]) continue;
        hipEvent_t *pe = &p->prof_ev[((size_t)slot * kProfKernels + which) * 2];
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, pe[0], pe[1]));
        ms[(*count)++] = t;
    }
    return CRSDR_OK;
}

#include "exchange_impl.hpp"
