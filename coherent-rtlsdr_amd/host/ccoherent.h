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
public:
    std::atomic<bool> do_exit;
    ccoherent(crefsdr *, lvector<csdrdevice *> *, crefnoise *, int nfft, int mode = CRSDR_MODE_FAITHFUL);
    ~ccoherent();
    void start();
    void request_exit();
    void join();
    size_t lagqueuesize();
    void clearlagqueue();
    void queuelag(csdrdevice *d);
    void computelag();
    bool step();                      // one threadf iteration; false on a device-side error
    const std::vector<float> &get_frac() const { return frac; }
    uint32_t get_locked_steps() const { return locked_steps; }   // blocks that ran the phase path only
};
#endif
