// csdrdevice.h -- the DSP-bearing part of the reference's device classes (include/csdrdevice.h:42-287,
// src/csdrdevice.cc, src/crtlsdr.cc:173-223): per-channel block hand-off (read / consume / readcnt),
// lag request handshake (requestfft / set_lag / get_lagp), and the per-row DSP members
// convtofloat / est_phasecorrect / phasecorrect.  Tuner control (librtlsdr) is out of scope; the
// concrete device here is a synthetic block source (csyntheticsdr) over host/csynth.c -- the
// hardware-free csdrdevice the reference only stubs (class czmqsdr, include/csdrdevice.h:270-272).
#ifndef CSDRDEVICEH
#define CSDRDEVICEH
#include <atomic>
#include <chrono>
#include <complex>
#include <cstdio>
#include <condition_variable>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "cdsp.h"
#include "cpacketizer.h"
extern "C" {
#include "csynth.h"
}

typedef std::complex<float> lv_32fc_t;   // layout-identical to VOLK's type (src/ccoherent.cc:65)

struct lagpoint {                        // include/csdrdevice.h:42-54
    uint64_t ts;
    float lag, mag, PAPR;
    lagpoint() : ts(0), lag(0), mag(0), PAPR(0) {}
};

class csdrdevice {
    uint32_t readcnt;
protected:
    std::mutex fftmtx;
    std::condition_variable fftcv;
    std::atomic<bool> lagrequested, lagready, synced, streaming;
    std::complex<float> *sfloat;          // B complex, zero-initialised (src/csdrdevice.cc:43-45)
    std::complex<float> phasecorr, phasecorrprev;
    uint32_t asyncbufn, blocksize, samplerate, fcenter;
    lagpoint lagp;
    std::string devname;
public:
    cpacketize packetize;

    virtual int8_t *read() = 0;                                   // include/csdrdevice.h:118
    virtual void consume() = 0;
    virtual uint32_t get_readcntbuf() = 0;
    virtual const std::complex<float> *convtofloat() = 0;        // src/crtlsdr.cc:205-207 / :215-218
    virtual const std::complex<float> *convtofloat(const std::complex<float> *) = 0;

    std::complex<float> est_phasecorrect(const lv_32fc_t *ref);  // src/csdrdevice.cc:58-69
    std::complex<float> get_phasecorrect() { return phasecorr; } // :76-78
    std::complex<float> *phasecorrect();                          // :80-84
    // the batched engine computes the EMA on the device; it mirrors the result back here
    void set_phasecorrect(std::complex<float> p) { phasecorr = p; phasecorrprev = p; }

    void requestfft() { lagready = false; lagrequested = true; } // include/csdrdevice.h:128
    void requestfftblocking();                                    // :129-136
    void set_lag(float lag, float mag);                           // :138-151
    bool is_lagrequested() { return lagrequested; }
    const std::complex<float> *get_sptr() { return sfloat; }
    const lagpoint *get_lagp() { return &lagp; }
    uint32_t get_blocksize() { return blocksize; }
    float get_samplerate() { return samplerate; }
    std::string get_devname() { return devname; }
    inline uint32_t inc_readcnt() { return readcnt++; }
    inline uint32_t get_readcnt() { return readcnt; }
    inline bool is_ready() { return (readcnt >= asyncbufn) && streaming; } // include/csdrdevice.h:187-189
    inline bool get_synchronized() { return synced; }
    inline void set_synchronized(bool s) { synced = s; }

    csdrdevice(uint32_t asyncbufn_, uint32_t blocksize_, uint32_t samplerate_, uint32_t fcenter_);
    virtual ~csdrdevice();
};

// One synthetic "antenna array": generates whole blocks (all rows) once and hands rows to devices.
class csynthsource {
    csynth_params *params;
    std::vector<int8_t> rows;
    int nsig, L, block;
public:
    csynthsource(int nsig_, int L_, uint64_t seed, int dmax, bool locked);
    ~csynthsource();
    void advance();                                   // generate the next block
    int8_t *row(int r) { return rows.data() + (size_t)r * 2 * L; }
    const csynth_params *get_params() const { return params; }
    int64_t get_delay(int k) const { return params->d[k]; }
    void set_delay(int k, int64_t d) { params->d[k] = d; }   // the resampler model of ccontrol moves this
    int blockindex() const { return block; }
    int get_L() const { return L; }
};

// ring of block buffers between a producer thread and the engine thread: the reference's cbuffer
// (include/common.h:41-149) with librtlsdr's buffers replaced by owned storage.  Raw offset-binary
// uint8 is kept as delivered; the XOR of cbuffer::setbufferptr -> cdsp::convtosigned (:114-122) is
// fused into the GPU loads instead (CRSDR_OFFSET_BINARY).
class cbuffer {
    std::vector<std::vector<uint8_t>> slot;
    std::vector<uint32_t> readcnt;
    uint32_t N, wp, rp;
public:
    cbuffer(uint32_t n, uint32_t blocksize) : slot(n, std::vector<uint8_t>(blocksize)), readcnt(n, 0), N(n), wp(0), rp(0) {}
    uint8_t *writeptr() { return slot[wp & (N - 1)].data(); }
    void commit(uint32_t rcnt) { readcnt[wp & (N - 1)] = rcnt; ++wp; }
    uint8_t *getbufferptr() { return slot[rp & (N - 1)].data(); }
    uint32_t get_rcnt() { return readcnt[rp & (N - 1)]; }
    void consume() { ++rp; }
    uint32_t backlog() const { return wp - rp; }
    uint32_t capacity() const { return N; }
};

// signal channel (crtlsdr equivalent): samples land in sfloat[0..L).
// Two ways to run it: synchronous (read() takes the row of the source's current block -- the demo's
// deterministic mode) or streaming (start(): a producer thread plays the librtlsdr callback thread,
// src/crtlsdr.cc:32-68,173-193: it fills the ring with raw offset-binary uint8 blocks and wakes read()).
class csyntheticsdr : public csdrdevice {
protected:
    csynthsource *src;
    int rowindex;
    int8_t *cur;
    // streaming mode
    std::unique_ptr<cbuffer> ring;
    std::thread producer;
    std::mutex mtx;
    std::condition_variable cv, cv_space;
    std::atomic<int> newdata{0};
    std::atomic<bool> do_exit{false};
    uint32_t cur_rcnt = 0, produced = 0, overruns = 0;
    bool held = false;          // guarded by mtx: read() handed the oldest ring slot to the engine, consume() has not returned it yet
    int pace_us = 0, max_blocks = 0;
    FILE *replay = nullptr;
    bool lossless = false;      // replay: a recording has no clock -- the producer waits for a free ring slot instead of dropping a block
    static void asynch_threadf(csyntheticsdr *d);
    // resampler model (ccontrol): accumulated slip in samples while a correction is set
    float correction = 0.0f;
    double slip = 0.0;
public:
    csyntheticsdr(csynthsource *s, int row, uint32_t blocksize_, uint32_t samplerate_ = 2048000, uint32_t fcenter_ = 0);
    ~csyntheticsdr() override;
    int8_t *read() override;
    void consume() override;
    uint32_t get_readcntbuf() override { return ring ? cur_rcnt : get_readcnt(); }
    void start(int pace_us_, int max_blocks_);          // crtlsdr::start src/crtlsdr.cc:24-30
    // Replay a recording instead of synthesising: `path` holds this channel's raw offset-binary uint8 IQ stream,
    // exactly what librtlsdr's callback delivers and `rtl_sdr -f ... file` writes; one ring block per blocksize bytes,
    // the producer stops at end of file.  (The reference's hardware-free device is the empty czmqsdr stub,
    // include/csdrdevice.h:270-272.)
    bool start_replay(const char *path, int pace_us_, int max_blocks_);
    void stop();                                        // :36-42
    bool is_streaming_raw() const { return (bool)ring; } // rows are raw uint8 (offset binary) in streaming mode
    uint32_t get_overruns() const { return overruns; }
    // streaming mode: the producer has made `total` blocks and none is left in the ring (read() would block for ever)
    bool produced_all(uint32_t total) { std::lock_guard<std::mutex> lock(mtx); return ring && get_readcnt() >= total; }
    uint32_t backlog() { std::lock_guard<std::mutex> lock(mtx); return ring ? ring->backlog() : 0; }
    bool drained(uint32_t total) { std::lock_guard<std::mutex> lock(mtx); return ring && get_readcnt() >= total && ring->backlog() == 0; }
    int set_correction_f(float f) { correction = f; return 0; } // crtlsdr::set_correction_f src/crtlsdr.cc:167-170
    float get_correction_f() const { return correction; }
    void advance_resampler();                           // one block at the current correction: slip += p * L
    int get_row() const { return rowindex; }
    const std::complex<float> *convtofloat() override;
    const std::complex<float> *convtofloat(const std::complex<float> *p) override;
};

// reference-noise channel (crefsdr, include/csdrdevice.h:274-287): samples land in sfloat[L..2L)
class crefsdr : public csyntheticsdr {
public:
    crefsdr(csynthsource *s, uint32_t blocksize_) : csyntheticsdr(s, 0, blocksize_) { devname = "M REF"; }
    const std::complex<float> *convtofloat() override;
    const std::complex<float> *convtofloat(const std::complex<float> *p) override;
};
#endif
