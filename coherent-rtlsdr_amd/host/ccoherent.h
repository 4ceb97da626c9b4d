// ccoherent.h -- the coherence engine with the reference's class surface (include/ccoherent.h:33-73):
// ccoherent(crefsdr*, lvector<csdrdevice*>*, crefnoise*, int nfft), start / join / request_exit,
// queuelag, computelag, lagqueuesize, clearlagqueue.  One threadf iteration = one block
// (src/ccoherent.cc:245-294); instead of per-row VOLK/FFTW calls it fills one [N][B] int8 matrix
// and hands it to the batched plan of libcrsdr.so (crsdr_plan_submit / crsdr_plan_fetch).
#ifndef CCOHERENTH
#define CCOHERENTH
#include <atomic>
#include <functional>
#include <thread>
#include <vector>

#include "../../include/crsdr.h"
#include "common.h"
#include "cpacketizer.h"
#include "crefnoise.h"
#include "csdrdevice.h"

// Rows split over the GPUs of a node, one process per GPU: rank r of `ranks` owns signal rows [1 + r * per, 1 + (r + 1) * per),
// per = (N - 1) / ranks, on HIP device `device`; `id` = the CRSDR_EXCHANGE_ID_BYTES rank 0 got from crsdr_exchange_unique_id and handed to
// the others (a file, a socket, MPI).  id == nullptr with ranks == 1: an ordinary unsharded engine on `device`.
struct ccoherent_shard {
    int ranks = 1, rank = 0, device = 0;
    const void *id = nullptr;
    int xmode = CRSDR_XCHG_STAGED;
};

class ccoherent {
    std::thread thread;
    static void threadf(ccoherent *);
    static void threadf_batched(ccoherent *, int T, int delay_us);
    lvector<csdrdevice *> *devices;
    crefsdr *refdev;
    crefnoise *refnoise;
    int nfft;          // kept for signature parity; the plan has no nfft = 8 queue cap (src/ccoherent.cc:124)
    int blocksize, nrows, mode;
    crsdr_plan *plan;
    std::vector<csdrdevice *> lagqueue;
    int8_t *rows, *packet;          // page-locked (crsdr_host_alloc), the engine-owned buffers of src/ccoherent.cc:44-47
    size_t packet_bytes;
    std::vector<uint32_t> readcnt;
    std::vector<uint8_t> mask;
    std::vector<int32_t> lag;
    std::vector<float> mag, frac, phasor;
    uint32_t seq;
    uint32_t locked_steps = 0;
    std::atomic<uint32_t> published{0};
    // batched / pipelined mode (enable_batching): two page-locked slots of T blocks in, two of T packets + scalars out
    int batch = 0;
    int8_t *brows[2] = {nullptr, nullptr}, *bpackets[2] = {nullptr, nullptr};
    int32_t *blag[2] = {nullptr, nullptr};
    float *bmag[2] = {nullptr, nullptr}, *bfrac[2] = {nullptr, nullptr}, *bphasor[2] = {nullptr, nullptr};
    size_t bstride = 0;                 // bytes between the host packets of a slot: >= one reference-length message (zero tail in place)
    int bcount[2] = {0, 0};
    // what fill_batch gathered beside the rows: every device's read counter per block (cpacketize::write's readcnt argument,
    // src/ccoherent.cc:253,278), which devices asked for a lag (:266) and the block flags (:271); false = the caller filled
    // batch_rows() itself (benchmark) and the plan numbers the blocks
    std::vector<uint32_t> breadcnt[2];
    std::vector<uint8_t> bmask[2];
    uint32_t bflags[2] = {0, 0};
    bool bfilled[2] = {false, false};
    bool bpublish = true;
    // sharded engine (ccoherent_shard with an id): the plan owns this rank's slab, the exchange assembles on a rotating root
    crsdr_exchange *xchg = nullptr;
    int ranks = 1, rank = 0, per = 0, row_begin = 1;
    int8_t *bscal[2] = {nullptr, nullptr}, *btails[2] = {nullptr, nullptr};    // page-locked: scalars blocks of the rooted blocks, own-row tails of every block
    size_t sstride = 0, tstride = 0;
    int brooted_first[2] = {0, 0}, brooted_count[2] = {0, 0};
public:
    std::atomic<bool> do_exit;
    ccoherent(crefsdr *, lvector<csdrdevice *> *, crefnoise *, int nfft, int mode = CRSDR_MODE_FAITHFUL, int batch = 1, const ccoherent_shard *shard = nullptr);
    bool ok() const { return plan != nullptr; }
    bool sharded() const { return xchg != nullptr; }
    int batch_rooted_first(int slot) const { return brooted_first[slot]; }     // sharded: the blocks of the slot's batch this rank assembled
    int batch_rooted_count(int slot) const { return brooted_count[slot]; }
    ~ccoherent();
    void start();
    // the same thread, a batch at a time and pipelined: fill(b) ; submit(b) ; collect(b - 1) -- src/ccoherent.cc:245-294 run T blocks
    // per iteration; every block's packet goes out through cpacketize::publish from this thread.  delay_us: idle time per
    // batch (tests use it to provoke ring overruns).
    void start_batched(int T, int delay_us = 0);
    uint32_t get_published_blocks() const { return published; }
    void request_exit();
    void join();
    size_t lagqueuesize();
    void clearlagqueue();
    void queuelag(csdrdevice *d);
    void computelag();
    bool step();                      // one threadf iteration; false on a device-side error
    // ---- the same loop a batch at a time, pipelined over PCIe (what the per-block step() cannot do: its submit + fetch are
    // synchronous through host memory).  The caller (or fill_batch) puts T blocks into batch_rows(slot); submit_batch sends
    // them and starts the asynchronous fetch of their results; while they are on their way the next slot is filled and
    // submitted; collect_batch waits for a slot's results and hands the last block's lags / phasors to the devices.
    //     fill(0); submit_batch(0, T); for (b = 1 ..) { fill(b & 1); submit_batch(b & 1, T); collect_batch((b - 1) & 1); }
    bool enable_batching(int T);      // false if the plan was not created for batches of T (ctor argument `batch`)
    int8_t *batch_rows(int slot) { return brows[slot]; }                       // [T][nrows][blocksize], page-locked
    const int8_t *batch_packet(int slot, int t) const { return bpackets[slot] + (size_t)t * bstride; }
    const int32_t *batch_lag(int slot, int t) const { return blag[slot] + (size_t)t * nrows; }
    const float *batch_phasor(int slot, int t) const { return bphasor[slot] + 2 * (size_t)t * nrows; }
    // nblocks x the per-block gather of step() (src/ccoherent.cc:250-283): read every device, keep ITS read counter for the
    // packet header, note who asked for a lag, consume.  before_block(t), if given, runs ahead of block t's reads (the
    // synchronous synthetic source is advanced there; streaming devices need nothing).
    bool fill_batch(int slot, int nblocks, const std::function<void(int)> &before_block = nullptr);
    // flags: extra CRSDR_* submit flags; a slot filled by fill_batch adds the engine's own (refnoise gate, raw uint8 rings,
    // CRSDR_NO_LAG when nobody asked) and ships its read counters and lag mask
    bool submit_batch(int slot, int nblocks, uint32_t flags);
    // waits for the slot's results; per block, in order: set_lag for every device that had asked (src/ccoherent.cc:232-233),
    // the phase factors to the devices / the debug payload (:279), and the block's packet to cpacketize::publish (:288)
    bool collect_batch(int slot);
    void set_batch_publish(bool on) { bpublish = on; }                        // false: collect_batch leaves the packets in batch_packet()
    const uint32_t *batch_readcnt(int slot, int t) const { return breadcnt[slot].data() + (size_t)t * nrows; }
    size_t get_packet_bytes() const { return packet_bytes; }
    const std::vector<float> &get_frac() const { return frac; }
    uint32_t get_locked_steps() const { return locked_steps; }   // blocks that ran the phase path only
};
#endif
