// ccoherent.h -- the coherence engine with the reference's class surface (include/ccoherent.h:33-73):
// ccoherent(crefsdr*, lvector<csdrdevice*>*, crefnoise*, int nfft), start / join / request_exit,
// queuelag, computelag, lagqueuesize, clearlagqueue.  One threadf iteration = one block
// (src/ccoherent.cc:245-294); instead of per-row VOLK/FFTW calls it fills one [N][B] int8 matrix
// and hands it to the batched plan of libcrsdr.so (crsdr_plan_submit / crsdr_plan_fetch).
#ifndef CCOHERENTH
#define CCOHERENTH
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/crsdr.h"
#include "common.h"
#include "cpacketizer.h"
#include "crefnoise.h"
#include "csdrdevice.h"

class ccoherent {
    std::thread thread;
    static void threadf(ccoherent *);
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
    // batched / pipelined mode (enable_batching): two page-locked slots of T blocks in, two of T packets + scalars out
    int batch = 0;
    int8_t *brows[2] = {nullptr, nullptr}, *bpackets[2] = {nullptr, nullptr};
    int32_t *blag[2] = {nullptr, nullptr};
    float *bmag[2] = {nullptr, nullptr}, *bfrac[2] = {nullptr, nullptr}, *bphasor[2] = {nullptr, nullptr};
    size_t bstride = 0;
    int bcount[2] = {0, 0};
public:
    std::atomic<bool> do_exit;
    ccoherent(crefsdr *, lvector<csdrdevice *> *, crefnoise *, int nfft, int mode = CRSDR_MODE_FAITHFUL, int batch = 1);
    ~ccoherent();
    void start();
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
    bool fill_batch(int slot, int nblocks);                                    // nblocks x (read every device, consume): the per-block gather of step()
    bool submit_batch(int slot, int nblocks, uint32_t flags);
    bool collect_batch(int slot);
    size_t get_packet_bytes() const { return packet_bytes; }
    const std::vector<float> &get_frac() const { return frac; }
    uint32_t get_locked_steps() const { return locked_steps; }   // blocks that ran the phase path only
};
#endif
