// cpacketizer.cc -- see cpacketizer.h.  Layout and hand-off follow src/cpacketizer.cc:91-185.
#include "cpacketizer.h"
#include "cdsp.h"
#include <cstdio>
#include <cstring>
#include <dlfcn.h>

// ---- libzmq C API, resolved at run time (zmq.h: ZMQ_PUB = 1, ZMQ_LINGER = 17) ---------------------
namespace {
struct zmq_api {
    void *lib = nullptr;
    void *(*ctx_new)() = nullptr;
    int (*ctx_term)(void *) = nullptr;
    void *(*socket)(void *, int) = nullptr;
    int (*close)(void *) = nullptr;
    int (*bind)(void *, const char *) = nullptr;
    int (*send)(void *, const void *, size_t, int) = nullptr;
    int (*setsockopt)(void *, int, const void *, size_t) = nullptr;
    bool load()
    {
        if (lib) return true;
        const char *names[] = {"libzmq.so.5", "libzmq.so", "/opt/conda/lib/libzmq.so.5", "/usr/lib/x86_64-linux-gnu/libzmq.so.5"};
        for (const char *n : names) { lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) return false;
        ctx_new = (void *(*)())dlsym(lib, "zmq_ctx_new");
        ctx_term = (int (*)(void *))dlsym(lib, "zmq_ctx_term");
        socket = (void *(*)(void *, int))dlsym(lib, "zmq_socket");
        close = (int (*)(void *))dlsym(lib, "zmq_close");
        bind = (int (*)(void *, const char *))dlsym(lib, "zmq_bind");
        send = (int (*)(void *, const void *, size_t, int))dlsym(lib, "zmq_send");
        setsockopt = (int (*)(void *, int, const void *, size_t))dlsym(lib, "zmq_setsockopt");
        return ctx_new && ctx_term && socket && close && bind && send && setsockopt;
    }
} zq;
void *g_ctx = nullptr, *g_sock = nullptr, *g_dbg = nullptr;
} // namespace
std::string cpacketize::debugaddress = "tcp://*:5557";
bool cpacketize::publishing() { return g_sock != nullptr; }

int cpacketize::objcount = 0;
uint32_t cpacketize::globalseqn = 0;
std::condition_variable cpacketize::cv;
std::mutex cpacketize::bmutex;
std::unique_ptr<int8_t[]> cpacketize::packetbuf0;
std::unique_ptr<int8_t[]> cpacketize::packetbuf1;
bool cpacketize::noheader = false;
size_t cpacketize::packetlen = 0;
bool cpacketize::bufferfilled = false;
uint32_t cpacketize::blocksize = 0;
bool cpacketize::do_exit = false;
std::vector<std::complex<float>> cpacketize::pcorrection;
cpacketize::sink_t cpacketize::sink;

cpacketize::cpacketize() { objcount++; }                     // src/cpacketizer.cc:45-51
cpacketize::~cpacketize() { objcount--; }

bool cpacketize::refpadding = true;    // the reference's message length is the default; false = the short form (no tail)
size_t cpacketize::packetlength(uint32_t N, uint32_t L)
{
    // L = blocksize in int8 values.  The reference sizes the message (16 + 4N) + 2*N*blocksize (src/cpacketizer.cc:91-96: it
    // counts the int8 values twice) and sends all of it (:125); that length is what goes on the wire by default, the tail
    // zero.  refpadding = false drops the tail (clients parse only the first N*blocksize data bytes, matlabclient/zmqsdr.c:121-143)
    const size_t data = (size_t)N * L * (refpadding ? 2 : 1);
    return noheader ? data : (16 + 4 * (size_t)N) + data;
}

void cpacketize::resize_buffers(uint32_t N, uint32_t L)
{
    std::lock_guard<std::mutex> lock(bmutex);
    packetlen = packetlength(N, L);
    packetbuf0.reset(new int8_t[packetlen]());
    packetbuf1.reset(new int8_t[packetlen]());
    pcorrection.assign(N, std::complex<float>(0.0f, 0.0f));
}

void cpacketize::init(std::string address, bool noheader_, uint32_t nchannels_, uint32_t blocksize_)
{
    noheader = noheader_;                                      // src/cpacketizer.cc:58-74
    blocksize = blocksize_;
    do_exit = false;
    bufferfilled = false;
    globalseqn = 0;
    resize_buffers(nchannels_, blocksize_);
    if (!address.empty()) {
        if (!zq.load()) {
            std::fprintf(stderr, "cpacketize: libzmq not found, packets go to the sink callback only\n");
        } else {
            const int linger = 0;
            g_ctx = zq.ctx_new();                              // context = new zmq::context_t(1)
            g_sock = zq.socket(g_ctx, 1 /* ZMQ_PUB */);        // socket->bind(address)
            zq.setsockopt(g_sock, 17 /* ZMQ_LINGER */, &linger, sizeof(linger));
            if (zq.bind(g_sock, address.c_str()) != 0) {
                std::fprintf(stderr, "cpacketize: cannot bind %s\n", address.c_str());
                zq.close(g_sock); g_sock = nullptr;
            }
            g_dbg = zq.socket(g_ctx, 1);                       // debugsocket->bind("tcp://*:5557")
            zq.setsockopt(g_dbg, 17, &linger, sizeof(linger));
            if (zq.bind(g_dbg, debugaddress.c_str()) != 0) { zq.close(g_dbg); g_dbg = nullptr; }
        }
    }
}

void cpacketize::cleanup()
{
    if (g_sock) { zq.close(g_sock); g_sock = nullptr; }        // src/cpacketizer.cc:76-83
    if (g_dbg) { zq.close(g_dbg); g_dbg = nullptr; }
    if (g_ctx) { zq.ctx_term(g_ctx); g_ctx = nullptr; }
    packetbuf0.reset(); packetbuf1.reset(); packetlen = 0;
}
void cpacketize::request_exit() { { std::lock_guard<std::mutex> l(bmutex); do_exit = true; } cv.notify_all(); }

int cpacketize::send()
{
    {   // src/cpacketizer.cc:119-123: wait for notifysend()
        std::unique_lock<std::mutex> lock(bmutex);
        cv.wait(lock, [] { return bufferfilled || do_exit; });
        if (!bufferfilled) return -1;
        bufferfilled = false;
    }
    if (!noheader) {
        // header of the buffer that was just swapped out.  (The reference fills packetbuf0's header
        // before the wait and then sends packetbuf1, src/cpacketizer.cc:110-125, so its sequence
        // number lags one packet behind the data; here the header belongs to the data it ships with.)
        hdr0 *hdr = reinterpret_cast<hdr0 *>(packetbuf1.get());
        hdr->globalseqn = globalseqn++;
        hdr->N = (uint32_t)objcount;
        hdr->L = blocksize >> 1;
        hdr->unused = 0;
    }
    if (g_sock) zq.send(g_sock, packetbuf1.get(), packetlen, 0);                                         // :125
    if (g_dbg) zq.send(g_dbg, pcorrection.data(), (size_t)objcount * sizeof(std::complex<float>), 0);      // :127
    if (sink) sink(packetbuf1.get(), packetlen, pcorrection.data(), (size_t)objcount);
    return 0;
}

int cpacketize::publish(const int8_t *message, size_t bytes, const std::complex<float> *phase, size_t n)
{
    // src/cpacketizer.cc:125,127 for a packet that was assembled elsewhere (header included: its globalseqn is the engine's)
    const int8_t *m = message;
    if (noheader) m += 16 + 4 * n;                             // -R: the matrix alone; `bytes` (= packetlength) already excludes the header
    if (g_sock) zq.send(g_sock, m, bytes, 0);
    if (g_dbg && phase) zq.send(g_dbg, phase, n * sizeof(std::complex<float>), 0);
    if (sink) sink(m, bytes, phase, n);
    globalseqn++;
    return 0;
}

int cpacketize::writedebug(uint32_t channeln, std::complex<float> p)
{
    pcorrection[channeln] = p;                                 // src/cpacketizer.cc:131-134
    return 0;
}

static inline uint32_t row_offset(bool noheader, int objcount, uint32_t channeln, uint32_t blocksize)
{
    return noheader ? channeln * blocksize : (uint32_t)(sizeof(hdr0) + objcount * sizeof(uint32_t)) + channeln * blocksize;
}

int cpacketize::write(uint32_t channeln, uint32_t readcnt, int8_t *rp)
{
    if (!noheader) *(reinterpret_cast<uint32_t *>(packetbuf0.get()) + sizeof(hdr0) / sizeof(uint32_t) + channeln) = readcnt;
    std::memcpy(packetbuf0.get() + row_offset(noheader, objcount, channeln, blocksize), rp, blocksize); // src/cpacketizer.cc:137-156
    return 0;
}

int cpacketize::write(uint32_t channeln, uint32_t readcnt, std::complex<float> *in)
{
    if (!noheader) *(reinterpret_cast<uint32_t *>(packetbuf0.get()) + sizeof(hdr0) / sizeof(uint32_t) + channeln) = readcnt;
    cdsp::convto8bit(reinterpret_cast<std::complex<int8_t> *>(packetbuf0.get() + row_offset(noheader, objcount, channeln, blocksize)),
                     in, (blocksize >> 1));                    // src/cpacketizer.cc:158-172
    return 0;
}

int cpacketize::notifysend()
{
    std::unique_lock<std::mutex> lock(bmutex);                 // src/cpacketizer.cc:174-185
    packetbuf0.swap(packetbuf1);
    bufferfilled = true;
    lock.unlock();
    cv.notify_one();
    return 0;
}
