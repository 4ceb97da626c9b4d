// cpacketizer.h -- the write side of the reference's cpacketize (include/cpacketizer.h:32-79,
// src/cpacketizer.cc:91-185): static double-buffered packet  hdr0 + u32 readcnt[N] + int8 IQ[N][L][2].
// init()/send() publish like the reference (ZMQ PUB on `address`, phase factors on the debug PUB,
// src/cpacketizer.cc:58-74,109-129) through libzmq's C API, loaded at run time (cppzmq's zmq.hpp is
// not available here); without libzmq, or with an empty address, packets only reach the sink callback.
#ifndef PACKETIZEH
#define PACKETIZEH
#include <complex>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

struct hdr0 { // include/cpacketizer.h:32-37
    uint32_t globalseqn;
    uint32_t N;
    uint32_t L;
    uint32_t unused;
};

class cpacketize {
public:
    typedef std::function<void(const int8_t *packet, size_t bytes, const std::complex<float> *phase, size_t n)> sink_t;

    cpacketize();
    ~cpacketize();

    // -- process-wide set-up (src/main.cc:261 calls init once; src/cpacketizer.cc:58-96)
    static void init(std::string address, bool noheader_, uint32_t nchannels_, uint32_t blocksize_);
    static void cleanup();
    static std::string debugaddress;   // reference: "tcp://*:5557" (src/cpacketizer.cc:66); set before init()
    static sink_t sink;                // every sent packet also goes here (embedding / tests)
    static bool publishing();          // true when the ZMQ PUB sockets are bound
    static bool refpadding;            // set before init(); default true: messages have the reference's length (below)
    // Message bytes for N channels of blocksize L int8 values: the reference's (16 + 4N) + 2*N*L (src/cpacketizer.cc:91-96,
    // sent whole at :125), of which clients read the first N*L data bytes (matlabclient/zmqsdr.c:121-143); the second
    // N*L bytes are a zero tail here (unspecified memory in the reference).  refpadding = false: (16 + 4N) + N*L.
    static size_t packetlength(uint32_t N, uint32_t L);

    // -- per block, from the engine thread (src/ccoherent.cc:253,277-279,288; src/cpacketizer.cc:137-185)
    int write(uint32_t channeln, uint32_t readcnt, int8_t *rp);                 // a row that is already int8
    int write(uint32_t channeln, uint32_t readcnt, std::complex<float> *in);    // convto8bit on the way in
    int writedebug(uint32_t channeln, std::complex<float> p);
    int notifysend();
    void request_exit();

    // -- publish loop, main thread (src/main.cc:277-279; src/cpacketizer.cc:109-129)
    static int send();
    // -- a message that is already assembled (the batched engine: the plan wrote hdr0 + readcnt + matrix, the zero tail is in
    // place): `bytes` = packetlength(N, blocksize) starting at the header -- with noheader the matrix alone is sent --
    // plus the N phase factors of the debug side channel.  Called from ONE thread; do not run the send() loop beside it.
    static int publish(const int8_t *message, size_t bytes, const std::complex<float> *phase, size_t n);

private:
    static void resize_buffers(uint32_t N, uint32_t L);
    static std::unique_ptr<int8_t[]> packetbuf0, packetbuf1;   // one being sent, one written to
    static std::vector<std::complex<float>> pcorrection;       // debug side channel: N phase factors
    static std::mutex bmutex;
    static std::condition_variable cv;
    static size_t packetlen;
    static uint32_t globalseqn, blocksize;
    static bool noheader, bufferfilled, do_exit;
    static int objcount;
};
#endif
