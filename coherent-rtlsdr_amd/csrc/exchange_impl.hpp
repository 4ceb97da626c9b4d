// exchange_impl.hpp -- the one exchange step of the path under the C ABI (include/crsdr.h, "exchange slots" and
// crsdr_exchange_*): slot geometry, the point-to-point schedule (pure host arithmetic, testable without GPUs), the
// assembly kernel launcher and the RCCL transport (librccl resolved with dlopen).  Included at the end of crsdr.hip.
#pragma once
#include <dlfcn.h>

namespace {

struct XGeo {
    int per = 0;               // signal rows per rank
    size_t rows_bytes = 0;     // per * B
    size_t tail_bytes = 0;     // 24 * per: lag i32 | mag f32 | frac f32 | phasor 2 x f32 | readcnt u32
    size_t tail_slot = 0;      // tail_bytes rounded up to 16 (stride of the in-place mode's tail staging)
    size_t slot = 0;           // rows_bytes + tail_bytes rounded up to 16
    size_t scalars = 0;        // 20 * nrows rounded up to 16
    size_t matrix_off = 0;     // 16 + 4 * nrows
};

inline size_t up16(size_t v) { return (v + 15) / 16 * 16; }

int xgeo(int nrows, int B, int nranks, XGeo *g)
{
    if (nrows < 2 || B < 16 || nranks < 1 || (nrows - 1) % nranks) return fail(CRSDR_EINVAL, "exchange: %d signal rows do not split evenly over %d ranks", nrows - 1, nranks);
    g->per = (nrows - 1) / nranks;
    g->rows_bytes = (size_t)g->per * (size_t)B;
    g->tail_bytes = 24 * (size_t)g->per;
    g->tail_slot = up16(g->tail_bytes);
    g->slot = up16(g->rows_bytes + g->tail_bytes);
    g->scalars = up16(20 * (size_t)nrows);
    g->matrix_off = 16 + 4 * (size_t)nrows;
    return CRSDR_OK;
}

inline void xrooted(int nblocks, int nranks, int q, int *first, int *count)
{
    const int bpr = (nblocks + nranks - 1) / nranks;
    const int f = std::min(q * bpr, nblocks);                 // a rank beyond the last run: [nblocks, nblocks)
    *first = f;
    *count = std::min(bpr, nblocks - f);
}

constexpr int kXSelfLoop = 0x100;     // diagnostic: the own chunk goes through the transport too (world-1 tests of the RCCL plumbing)

// the operations of one batch for `rank`, in issue order
void xschedule(const XGeo &g, int nranks, int rank, int nblocks, int mode, int B, size_t packet_stride, std::vector<crsdr_xop> &ops)
{
    const bool inplace = (mode & 0xff) == CRSDR_XCHG_INPLACE, selfloop = (mode & kXSelfLoop) != 0;
    int myf = 0, myc = 0;
    xrooted(nblocks, nranks, rank, &myf, &myc);
    auto peer_ok = [&](int r) { return r != rank || selfloop; };
    if (!inplace) {
        for (int r = 0; r < nranks; ++r)                       // chunk r of the staging: rank r's slots of my blocks
            if (peer_ok(r) && myc > 0) ops.push_back({r, 1, 1, myf, (uint64_t)r * (uint64_t)myc * g.slot, (uint64_t)myc * g.slot});
        for (int q = 0; q < nranks; ++q) {
            int f, c;
            xrooted(nblocks, nranks, q, &f, &c);
            if (peer_ok(q) && c > 0) ops.push_back({q, 0, 0, f, (uint64_t)f * g.slot, (uint64_t)c * g.slot});
        }
        return;
    }
    // in place: rows of block (myf + j) from rank r land in matrix rows 1 + r*per of packet j; tails go to the tail staging
    for (int j = 0; j < myc; ++j)
        for (int r = 0; r < nranks; ++r)
            if (peer_ok(r)) ops.push_back({r, 1, 2, myf + j, (uint64_t)j * packet_stride + g.matrix_off + ((uint64_t)1 + (uint64_t)r * g.per) * (uint64_t)B, g.rows_bytes});
    for (int r = 0; r < nranks; ++r)
        for (int j = 0; j < myc; ++j)
            if (peer_ok(r)) ops.push_back({r, 1, 3, myf + j, ((uint64_t)r * myc + j) * g.tail_slot, g.tail_bytes});
    for (int q = 0; q < nranks; ++q) {
        int f, c;
        xrooted(nblocks, nranks, q, &f, &c);
        if (!peer_ok(q)) continue;
        for (int t = f; t < f + c; ++t) ops.push_back({q, 0, 0, t, (uint64_t)t * g.slot, g.rows_bytes});
    }
    for (int q = 0; q < nranks; ++q) {
        int f, c;
        xrooted(nblocks, nranks, q, &f, &c);
        if (!peer_ok(q)) continue;
        for (int t = f; t < f + c; ++t) ops.push_back({q, 0, 0, t, (uint64_t)t * g.slot + g.rows_bytes, g.tail_bytes});
    }
}

// rows / tails of `nsrc` chunks (real source rank = src_base + z) into packets / scalars blocks
int launch_assemble(hipStream_t s, int8_t *packets, size_t packet_stride, int8_t *scalars, size_t scalars_stride, int nrows, int per, int B,
                    const int8_t *recv, int nsrc, int nblocks, size_t slot_stride, size_t tail_offset, int has_tail, int src_base, int skip_rank,
                    int self_rank, const int8_t *self_src, int rows_in_place)
{
    if (nblocks < 1 || nsrc < 1) return CRSDR_OK;
    const size_t body_off = 16 + 4 * (size_t)nrows + (size_t)B;       // row 1 of the matrix
    const size_t rows_bytes = (size_t)per * (size_t)B;
    const bool v16 = (((uintptr_t)packets + body_off) % 16 == 0) && (packet_stride % 16 == 0) && ((uintptr_t)recv % 16 == 0) && (slot_stride % 16 == 0) &&
                     (rows_bytes % 16 == 0) && (!self_src || (uintptr_t)self_src % 16 == 0);
    const unsigned chunks = rows_in_place && !self_src ? 1u : (unsigned)std::min<size_t>(std::max<size_t>(1, rows_bytes / 16 / 1024), 64);
    const dim3 grid(chunks, (unsigned)nblocks, (unsigned)nsrc);
    if (v16)
        hipLaunchKernelGGL(k_assemble_slots<uint4>, grid, dim3(256), 0, s, packets, packet_stride, body_off, scalars, scalars_stride, nrows, per, B, recv, nblocks,
                           slot_stride, tail_offset, has_tail, src_base, skip_rank, self_rank, self_src, rows_in_place);
    else
        hipLaunchKernelGGL(k_assemble_slots<uint32_t>, grid, dim3(256), 0, s, packets, packet_stride, body_off, scalars, scalars_stride, nrows, per, B, recv, nblocks,
                           slot_stride, tail_offset, has_tail, src_base, skip_rank, self_rank, self_src, rows_in_place);
    HIP_TRY(hipGetLastError());
    return CRSDR_OK;
}

// ---- librccl, resolved at run time --------------------------------------------------------------------------------
// (rccl.h: ncclUniqueId is 128 opaque bytes passed by value; ncclInt8 = 0; every call returns ncclSuccess = 0)
struct xid { char internal[CRSDR_EXCHANGE_ID_BYTES]; };
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(xid *) = nullptr;
    int (*CommInitRank)(void **, int, xid, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::mutex mu;
    bool load()
    {
        std::lock_guard<std::mutex> lock(mu);
        if (lib) return true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void *h = nullptr;
        for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
        if (!h) return false;
        GetUniqueId = (int (*)(xid *))dlsym(h, "ncclGetUniqueId");
        CommInitRank = (int (*)(void **, int, xid, int))dlsym(h, "ncclCommInitRank");
        CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
        GroupStart = (int (*)())dlsym(h, "ncclGroupStart");
        GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
        Send = (int (*)(const void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclSend");
        Recv = (int (*)(void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclRecv");
        GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
        if (!(GetUniqueId && CommInitRank && CommDestroy && GroupStart && GroupEnd && Send && Recv && GetErrorString)) { dlclose(h); return false; }
        lib = h;
        return true;
    }
} g_rccl;

#define RCCL_TRY(call)                                                                                     \
    do {                                                                                                   \
        int r_ = (call);                                                                                   \
        if (r_ != 0) return fail(CRSDR_EHIP, "%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

} // namespace

struct crsdr_exchange {
    void *comm = nullptr;
    int nranks = 0, rank = 0, device = 0;
    bool selfloop = false;
    int8_t *tail_stage = nullptr;
    size_t tail_cap = 0;
    // a bound plan (crsdr_exchange_bind_plan): two ring-buffered sets of device buffers, the exchange on its own stream
    crsdr_plan *plan = nullptr;
    int xmode = 0, bpr = 0;
    XGeo geo;
    size_t pstride = 0, pk_off[2] = {0, 0};
    int8_t *d_send[2] = {nullptr, nullptr}, *d_recv[2] = {nullptr, nullptr}, *d_packets[2] = {nullptr, nullptr}, *d_scal[2] = {nullptr, nullptr};
    hipStream_t xs = nullptr, cs = nullptr;
    hipEvent_t ev_sub = nullptr, ev_done[2] = {nullptr, nullptr};
    bool done_valid[2] = {false, false};
    struct { int nblocks, first, count; unsigned long idx; } out[2] = {};
    unsigned long head = 0, tail = 0;          // batches fetched / submitted
};

extern "C" int crsdr_plan_bind_slab_ex(crsdr_plan *p, void *device_slab, size_t slab_stride, int hdr_first, int hdr_count, size_t tail_offset)
{
    if (!p) return fail(CRSDR_EINVAL, "plan_bind_slab_ex: NULL plan");
    int rc = crsdr_plan_bind_slab(p, device_slab, slab_stride, hdr_first, hdr_count);
    if (rc) return rc;
    p->tail_offset = 0;
    if (!device_slab || tail_offset == 0) return CRSDR_OK;
    if (tail_offset % 4 || tail_offset < (size_t)p->row_count * (size_t)p->B || tail_offset + 24 * (size_t)p->row_count > slab_stride) {
        (void)crsdr_plan_bind_slab(p, nullptr, 0, 0, 0);
        return fail(CRSDR_EINVAL, "plan_bind_slab_ex: tail_offset must be 4-byte aligned, >= row_count*blocksize, and leave 24 bytes per owned row inside the slot");
    }
    p->tail_offset = tail_offset;
    return CRSDR_OK;
}

extern "C" int crsdr_exchange_geometry(int nrows, int blocksize, int nranks, size_t *slot_stride, size_t *tail_offset, size_t *scalars_stride)
{
    XGeo g;
    int rc = xgeo(nrows, blocksize, nranks, &g);
    if (rc) return rc;
    if (slot_stride) *slot_stride = g.slot;
    if (tail_offset) *tail_offset = g.rows_bytes;
    if (scalars_stride) *scalars_stride = g.scalars;
    return CRSDR_OK;
}

extern "C" int crsdr_exchange_rooted_blocks(int nblocks, int nranks, int rank, int *first, int *count)
{
    if (nblocks < 0 || nranks < 1 || rank < 0 || rank >= nranks || !first || !count) return fail(CRSDR_EINVAL, "exchange_rooted_blocks: bad argument");
    xrooted(nblocks, nranks, rank, first, count);
    return CRSDR_OK;
}

extern "C" int crsdr_assemble_slots(void *device_packets, size_t packet_stride, void *device_scalars, size_t scalars_stride, int nrows, int blocksize,
                                    const void *device_recv, int nsrc, int nblocks, size_t slot_stride, size_t tail_offset, int self_rank,
                                    const void *device_self, void *hip_stream)
{
    XGeo g;
    { int rc = xgeo(nrows, blocksize, nsrc, &g); if (rc) return rc; }
    if (!device_packets || (!device_recv && !(nsrc == 1 && device_self)) || nblocks < 1 || nblocks > 65535 || nsrc > 65535)
        return fail(CRSDR_EINVAL, "assemble_slots: need packets, recv, 1 <= nblocks, nsrc <= 65535");
    const size_t body_off = g.matrix_off + (size_t)blocksize;
    if (((uintptr_t)device_packets + body_off) % 4 || packet_stride % 4 || (uintptr_t)device_recv % 4 || slot_stride % 4 || tail_offset % 4 ||
        slot_stride < g.rows_bytes || (tail_offset && (tail_offset < g.rows_bytes || tail_offset + g.tail_bytes > slot_stride)) ||
        (nblocks > 1 && packet_stride < body_off + (size_t)(nrows - 1) * (size_t)blocksize) ||
        (device_scalars && ((uintptr_t)device_scalars % 4 || scalars_stride % 4 || (nblocks > 1 && scalars_stride < 20 * (size_t)nrows))) ||
        (device_self && (uintptr_t)device_self % 4))
        return fail(CRSDR_EINVAL, "assemble_slots: 4-byte alignment of every pointer / stride, slots >= rows (+ tail), strides >= one packet / scalars block");
    { int rc_ = require_device(); if (rc_) return rc_; }
    return launch_assemble((hipStream_t)hip_stream, (int8_t *)device_packets, packet_stride, (int8_t *)device_scalars, scalars_stride, nrows, g.per, blocksize,
                           (const int8_t *)device_recv, nsrc, nblocks, slot_stride, tail_offset, tail_offset != 0 /* read counters reach the headers with or without a scalars block */, 0, -1,
                           device_self ? self_rank : -1, (const int8_t *)device_self, 0);
}

extern "C" int crsdr_exchange_schedule(int nranks, int rank, int nblocks, int mode, int nrows, int blocksize, size_t packet_stride,
                                       crsdr_xop *ops, int capacity, int *count)
{
    if (!count || rank < 0 || rank >= nranks || nblocks < 1 || capacity < 0 || (capacity > 0 && !ops)) return fail(CRSDR_EINVAL, "exchange_schedule: bad argument");
    if ((mode & 0xff) != CRSDR_XCHG_STAGED && (mode & 0xff) != CRSDR_XCHG_INPLACE) return fail(CRSDR_EINVAL, "exchange_schedule: mode = %d", mode);
    XGeo g;
    { int rc = xgeo(nrows, blocksize, nranks, &g); if (rc) return rc; }
    std::vector<crsdr_xop> v;
    xschedule(g, nranks, rank, nblocks, mode, blocksize, packet_stride, v);
    *count = (int)v.size();
    for (int i = 0; i < (int)v.size() && i < capacity; ++i) ops[i] = v[i];
    return CRSDR_OK;
}

extern "C" int crsdr_exchange_unique_id(void *id)
{
    if (!id) return fail(CRSDR_EINVAL, "exchange_unique_id: NULL");
    { int rc_ = require_device(); if (rc_) return rc_; }
    if (!g_rccl.load()) {
        const char *why = dlerror();           // read once: a second call returns NULL
        return fail(CRSDR_ENODEV, "exchange: librccl.so.1 not found (%s)", why ? why : "symbols missing");
    }
    xid u;
    RCCL_TRY(g_rccl.GetUniqueId(&u));
    std::memcpy(id, u.internal, CRSDR_EXCHANGE_ID_BYTES);
    return CRSDR_OK;
}

extern "C" int crsdr_exchange_create(crsdr_exchange **x, const void *id, int nranks, int rank, int device)
{
    if (!x || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(CRSDR_EINVAL, "exchange_create: bad argument");
    *x = nullptr;
    { int rc_ = require_device(); if (rc_) return rc_; }
    if (!g_rccl.load()) return fail(CRSDR_ENODEV, "exchange: librccl.so.1 not found");
    HIP_TRY(hipSetDevice(device));
    crsdr_exchange *e = new (std::nothrow) crsdr_exchange();
    if (!e) return fail(CRSDR_ENOMEM, "exchange_create: out of host memory");
    e->nranks = nranks; e->rank = rank; e->device = device;
    { const char *s = getenv("CRSDR_XCHG_SELF"); e->selfloop = s && atoi(s) != 0; }
    xid u;
    std::memcpy(u.internal, id, CRSDR_EXCHANGE_ID_BYTES);
    int r = g_rccl.CommInitRank(&e->comm, nranks, u, rank);
    if (r != 0) { delete e; return fail(CRSDR_EHIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r)); }
    *x = e;
    return CRSDR_OK;
}

// what crsdr_exchange_bind_plan allocated (all of it, or what a bind that failed half way got as far as)
static void xengine_release(crsdr_exchange *x)
{
    for (int k = 0; k < 2; ++k) {
        for (int8_t **b : {&x->d_send[k], &x->d_recv[k], &x->d_packets[k], &x->d_scal[k]}) { if (*b) (void)hipFree(*b); *b = nullptr; }
        if (x->ev_done[k]) (void)hipEventDestroy(x->ev_done[k]);
        x->ev_done[k] = nullptr; x->done_valid[k] = false;
    }
    if (x->ev_sub) (void)hipEventDestroy(x->ev_sub);
    if (x->xs) (void)hipStreamDestroy(x->xs);
    if (x->cs) (void)hipStreamDestroy(x->cs);
    x->ev_sub = nullptr; x->xs = nullptr; x->cs = nullptr;
}

extern "C" int crsdr_exchange_destroy(crsdr_exchange *x)
{
    if (!x) return fail(CRSDR_EINVAL, "exchange_destroy: NULL");
    (void)hipSetDevice(x->device);
    if (x->plan) {
        (void)hipDeviceSynchronize();
        (void)crsdr_plan_bind_slab(x->plan, nullptr, 0, 0, 0);
        (void)crsdr_plan_bind_packet(x->plan, nullptr, 0);
    }
    if (x->comm) (void)g_rccl.CommDestroy(x->comm);
    if (x->tail_stage) (void)hipFree(x->tail_stage);
    xengine_release(x);
    delete x;
    return CRSDR_OK;
}

extern "C" int crsdr_exchange_batch(crsdr_exchange *x, int mode, const void *device_send, void *device_recv, int nblocks, void *device_packets,
                                    size_t packet_stride, void *device_scalars, size_t scalars_stride, int nrows, int blocksize, void *hip_stream)
{
    if (!x || !device_send || !device_packets || nblocks < 1) return fail(CRSDR_EINVAL, "exchange_batch: NULL exchange / send / packets or no blocks");
    if (mode != CRSDR_XCHG_STAGED && mode != CRSDR_XCHG_INPLACE) return fail(CRSDR_EINVAL, "exchange_batch: mode = %d", mode);
    if (mode == CRSDR_XCHG_STAGED && !device_recv) return fail(CRSDR_EINVAL, "exchange_batch: the staged mode needs device_recv");
    XGeo g;
    { int rc = xgeo(nrows, blocksize, x->nranks, &g); if (rc) return rc; }
    if (((uintptr_t)device_packets + g.matrix_off) % 4 || packet_stride % 4 || (uintptr_t)device_send % 16 || (device_recv && (uintptr_t)device_recv % 16))
        return fail(CRSDR_EINVAL, "exchange_batch: send / recv 16-byte aligned, matrix start and packet stride 4-byte aligned");
    {
        // the same stride checks as crsdr_assemble_slots: a short stride would let the assembly kernel -- or, in place, ncclRecv itself --
        // write one block's rows / scalars over its neighbour's
        int f0 = 0, c0 = 0;
        xrooted(nblocks, x->nranks, x->rank, &f0, &c0);
        const size_t body_off = g.matrix_off + (size_t)blocksize;
        if (c0 > 1 && packet_stride < body_off + (size_t)(nrows - 1) * (size_t)blocksize)
            return fail(CRSDR_EINVAL, "exchange_batch: packet_stride %zu is smaller than one packet (%zu bytes) and this rank assembles %d blocks", packet_stride,
                        body_off + (size_t)(nrows - 1) * (size_t)blocksize, c0);
        if (device_scalars && ((uintptr_t)device_scalars % 4 || scalars_stride % 4 || (c0 > 1 && scalars_stride < 20 * (size_t)nrows)))
            return fail(CRSDR_EINVAL, "exchange_batch: device_scalars / scalars_stride 4-byte aligned, scalars_stride >= 20 * nrows");
    }
    HIP_TRY(hipSetDevice(x->device));
    hipStream_t s = (hipStream_t)hip_stream;
    const int bpr = (nblocks + x->nranks - 1) / x->nranks;
    int myf = 0, myc = 0;
    xrooted(nblocks, x->nranks, x->rank, &myf, &myc);
    if (mode == CRSDR_XCHG_INPLACE) {
        const size_t need = (size_t)x->nranks * (size_t)bpr * g.tail_slot;
        if (x->tail_cap < need) {                              // grows on the first batch (and when the geometry changes), never per batch after that
            HIP_TRY(hipDeviceSynchronize());                    // earlier batches may have used other streams: nothing may still read the old staging
            if (x->tail_stage) (void)hipFree(x->tail_stage);
            x->tail_stage = nullptr; x->tail_cap = 0;
            HIP_TRY(hipMalloc((void **)&x->tail_stage, need));
            x->tail_cap = need;
        }
    }
    std::vector<crsdr_xop> ops;
    xschedule(g, x->nranks, x->rank, nblocks, mode | (x->selfloop ? kXSelfLoop : 0), blocksize, packet_stride, ops);
    int8_t *base[4] = {(int8_t *)const_cast<void *>(device_send), (int8_t *)device_recv, (int8_t *)device_packets, x->tail_stage};
    if (!ops.empty()) {
        RCCL_TRY(g_rccl.GroupStart());
        for (const crsdr_xop &o : ops) {
            int r = o.is_recv ? g_rccl.Recv(base[o.buffer] + o.offset, (size_t)o.bytes, 0 /* ncclInt8 */, o.peer, x->comm, s)
                              : g_rccl.Send(base[o.buffer] + o.offset, (size_t)o.bytes, 0, o.peer, x->comm, s);
            if (r != 0) { (void)g_rccl.GroupEnd(); return fail(CRSDR_EHIP, "ncclSend / ncclRecv failed: %s", g_rccl.GetErrorString(r)); }
        }
        RCCL_TRY(g_rccl.GroupEnd());
    }
    if (myc < 1) return CRSDR_OK;
    const int8_t *self_slots = (const int8_t *)device_send + (size_t)myf * g.slot;
    const bool tails = true;      // exchange slots always carry tails: their read counters go into the packet headers even where no scalars block is wanted
    if (mode == CRSDR_XCHG_STAGED)
        return launch_assemble(s, (int8_t *)device_packets, packet_stride, (int8_t *)device_scalars, scalars_stride, nrows, g.per, blocksize, (const int8_t *)device_recv,
                               x->nranks, myc, g.slot, g.rows_bytes, tails, 0, -1, x->selfloop ? -1 : x->rank, x->selfloop ? nullptr : self_slots, 0);
    // in place: remote rows already sit in the packets; remote tails come from the tail staging ([r][bpr] slots, this rank's myc of them used),
    // the rank's own rows + tails from its send slots
    if (tails) {
        int rc = launch_assemble(s, (int8_t *)device_packets, packet_stride, (int8_t *)device_scalars, scalars_stride, nrows, g.per, blocksize, x->tail_stage,
                                 x->nranks, myc, g.tail_slot, 0, 1, 0, x->selfloop ? -1 : x->rank, -1, nullptr, 1);
        if (rc) return rc;
    }
    if (x->selfloop) return CRSDR_OK;
    return launch_assemble(s, (int8_t *)device_packets, packet_stride, (int8_t *)device_scalars, scalars_stride, nrows, g.per, blocksize, self_slots, 1, myc, g.slot,
                           g.rows_bytes, tails, x->rank, -1, -1, nullptr, 0);
}


// ---- a sharded plan and its exchange as one engine (include/crsdr.h) -------------------------------------------------------------
extern "C" int crsdr_exchange_bind_plan(crsdr_exchange *x, crsdr_plan *plan, int mode)
{
    if (!x || !plan) return fail(CRSDR_EINVAL, "exchange_bind_plan: NULL exchange or plan");
    if (mode != CRSDR_XCHG_STAGED && mode != CRSDR_XCHG_INPLACE) return fail(CRSDR_EINVAL, "exchange_bind_plan: mode = %d", mode);
    if (x->plan) return fail(CRSDR_ESTATE, "exchange_bind_plan: a plan is already bound");
    XGeo g;
    { int rc = xgeo(plan->nrows, plan->B, x->nranks, &g); if (rc) return rc; }
    if (plan->device != x->device || plan->row_count != g.per || plan->row_begin != 1 + x->rank * g.per)
        return fail(CRSDR_EINVAL, "exchange_bind_plan: the plan must live on device %d and own rank %d's slab: rows [%d, %d) of %d (it owns [%d, %d) on device %d)", x->device,
                    x->rank, 1 + x->rank * g.per, 1 + (x->rank + 1) * g.per, plan->nrows, plan->row_begin, plan->row_begin + plan->row_count, plan->device);
    HIP_TRY(hipSetDevice(x->device));
    x->geo = g; x->xmode = mode;
    x->bpr = (plan->max_batch + x->nranks - 1) / x->nranks;
    x->pstride = (plan->packet_bytes + 255) / 256 * 256;
    const int rc_alloc = [&]() -> int {
        HIP_TRY(hipStreamCreateWithFlags(&x->xs, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&x->cs, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&x->ev_sub, hipEventDisableTiming));
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(hipEventCreateWithFlags(&x->ev_done[k], hipEventDisableTiming));
            HIP_TRY(hipMalloc((void **)&x->d_send[k], (size_t)plan->max_batch * g.slot));
            HIP_TRY(hipMalloc((void **)&x->d_recv[k], (size_t)x->nranks * (size_t)x->bpr * g.slot));
            HIP_TRY(hipMalloc((void **)&x->d_packets[k], (size_t)x->bpr * x->pstride + 256));
            HIP_TRY(hipMalloc((void **)&x->d_scal[k], (size_t)x->bpr * g.scalars));
            HIP_TRY(hipMemset(x->d_packets[k], 0, (size_t)x->bpr * x->pstride + 256));
            HIP_TRY(hipMemset(x->d_scal[k], 0, (size_t)x->bpr * g.scalars));
            x->pk_off[k] = (256 - ((uintptr_t)x->d_packets[k] + g.matrix_off) % 256) % 256;      // the matrix of every packet 256-byte aligned
        }
        HIP_TRY(hipDeviceSynchronize());
        return CRSDR_OK;
    }();
    if (rc_alloc) { xengine_release(x); return rc_alloc; }    // (the error text is the failing call's)
    x->plan = plan;
    return CRSDR_OK;
}

extern "C" int crsdr_exchange_submit_batch(crsdr_exchange *x, const void *rows, int mem_kind, int nblocks, size_t block_stride, const uint32_t *readcnt,
                                           const uint8_t *lag_mask, uint32_t seq, uint32_t flags)
{
    if (!x || !x->plan) return fail(CRSDR_ESTATE, "exchange_submit_batch: no plan bound (crsdr_exchange_bind_plan)");
    if (x->tail - x->head >= 2) return fail(CRSDR_ESTATE, "exchange_submit_batch: two batches outstanding, call crsdr_exchange_fetch_rooted first");
    crsdr_plan *p = x->plan;
    if (nblocks < 1 || nblocks > p->max_batch) return fail(CRSDR_EINVAL, "exchange_submit_batch: nblocks = %d (plan max_batch = %d)", nblocks, p->max_batch);
    HIP_TRY(hipSetDevice(x->device));
    const int k = (int)(x->tail & 1);
    hipStream_t S = p->stream;
    if (x->done_valid[k]) HIP_TRY(hipStreamWaitEvent(S, x->ev_done[k], 0));      // set k is free again: its exchange and assembly are complete
    int f = 0, c = 0;
    xrooted(nblocks, x->nranks, x->rank, &f, &c);
    int8_t *pk = x->d_packets[k] + x->pk_off[k];
    int rc;
    if ((rc = crsdr_plan_bind_packet(p, pk, x->pstride))) return rc;
    if ((rc = crsdr_plan_bind_slab_ex(p, x->d_send[k], x->geo.slot, f, c, x->geo.rows_bytes))) return rc;
    if ((rc = crsdr_plan_submit_batch(p, rows, mem_kind, nblocks, block_stride, readcnt, lag_mask, seq, flags))) return rc;
    HIP_TRY(hipEventRecord(x->ev_sub, S));
    HIP_TRY(hipStreamWaitEvent(x->xs, x->ev_sub, 0));
    // the exchange and the assembly run on the side stream: the plan's stream is free for the next batch (into the other set)
    if ((rc = crsdr_exchange_batch(x, x->xmode, x->d_send[k], x->d_recv[k], nblocks, pk, x->pstride, x->d_scal[k], x->geo.scalars, p->nrows, p->B, x->xs))) return rc;
    HIP_TRY(hipEventRecord(x->ev_done[k], x->xs));
    x->done_valid[k] = true;
    x->out[k].nblocks = nblocks; x->out[k].first = f; x->out[k].count = c; x->out[k].idx = p->submit_idx - 1;
    x->tail++;
    return CRSDR_OK;
}

extern "C" int crsdr_exchange_fetch_rooted(crsdr_exchange *x, int8_t *packets, size_t host_packet_stride, void *scalars, size_t host_scalars_stride,
                                           void *own_tails, size_t host_tails_stride, int *first, int *count, int *nblocks)
{
    if (!x || !x->plan) return fail(CRSDR_ESTATE, "exchange_fetch_rooted: no plan bound");
    if (x->head == x->tail) return fail(CRSDR_ESTATE, "exchange_fetch_rooted: nothing submitted");
    crsdr_plan *p = x->plan;
    const int k = (int)(x->head & 1), c = x->out[k].count;
    if (packets && c > 1 && host_packet_stride < p->packet_bytes) return fail(CRSDR_EINVAL, "exchange_fetch_rooted: host_packet_stride smaller than a packet");
    if (scalars && c > 1 && host_scalars_stride < 20 * (size_t)p->nrows) return fail(CRSDR_EINVAL, "exchange_fetch_rooted: host_scalars_stride smaller than 20 * nrows");
    if (own_tails && x->out[k].nblocks > 1 && host_tails_stride < x->geo.tail_bytes) return fail(CRSDR_EINVAL, "exchange_fetch_rooted: host_tails_stride smaller than 24 * rows per rank");
    HIP_TRY(hipSetDevice(x->device));
    HIP_TRY(hipStreamWaitEvent(x->cs, x->ev_done[k], 0));
    const int8_t *pk = x->d_packets[k] + x->pk_off[k];
    if (own_tails)       // {lag, mag, frac, phasor} of this rank's own rows for EVERY block of the batch: the tails of its send slots
        HIP_TRY(hipMemcpy2DAsync(own_tails, host_tails_stride, x->d_send[k] + x->geo.rows_bytes, x->geo.slot, x->geo.tail_bytes, (size_t)x->out[k].nblocks,
                                 hipMemcpyDeviceToHost, x->cs));
    for (int j = 0; j < c; ++j) {
        if (packets) HIP_TRY(hipMemcpyAsync(packets + (size_t)j * host_packet_stride, pk + (size_t)j * x->pstride, p->packet_bytes, hipMemcpyDeviceToHost, x->cs));
        if (scalars) HIP_TRY(hipMemcpyAsync((int8_t *)scalars + (size_t)j * host_scalars_stride, x->d_scal[k] + (size_t)j * x->geo.scalars, 20 * (size_t)p->nrows,
                                            hipMemcpyDeviceToHost, x->cs));
    }
    // the two-row kernel's status word behind this batch's kernels (as crsdr_plan_fetch_wait reads it)
    HIP_TRY(hipMemcpyAsync(&p->h_k1flag[x->head % 4], p->d_sync + 2, sizeof(int), hipMemcpyDeviceToHost, x->cs));
    HIP_TRY(hipStreamSynchronize(x->cs));
    const int w = p->h_k1flag[x->head % 4];
    const unsigned long idx = x->out[k].idx;
    if (first) *first = x->out[k].first;
    if (count) *count = c;
    if (nblocks) *nblocks = x->out[k].nblocks;
    x->head++;
    if (w) {
        x->head = x->tail;                                   // whatever else was outstanding went through the same garbage lags
        x->done_valid[0] = x->done_valid[1] = false;
        (void)hipDeviceSynchronize();
        return crsdr_plan_sync(p);                            // rolls the plan back and reports
    }
    retire_snapshots(p, idx);
    return CRSDR_OK;
}
